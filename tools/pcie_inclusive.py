#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: sequences start in pinned HOST memory, are copied to HBM
and deduplicated (never the bench `value`, which starts with the input resident in HBM)."""
import sys, time
sys.path.insert(0, ".")
import torch
from fastq_dupaway_amd import Engine, Reads

n, L = 40_000_000, 150
dev = torch.device("cuda", 0)
d_bases = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
with Engine(segments=1, capacity_reads=n, capacity_bases=n * L) as e:
    e.synth_reads(3, 0, n, L, 200, 0, d_bases, None); e.sync()
    h_bases = torch.empty(n * L + 16, dtype=torch.uint8).pin_memory()
    h_bases.copy_(d_bases); torch.cuda.synchronize()
    h_keep = torch.empty(n, dtype=torch.uint8).pin_memory()
    chunks = 8; m = n // chunks
    for rep in range(3):
        e.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(chunks):                      # copy chunk k, dedup it; copies and kernels are stream-ordered
            d_bases[k * m * L:(k + 1) * m * L].copy_(h_bases[k * m * L:(k + 1) * m * L], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            e.submit([Reads(d_bases[k * m * L:], uniform_len=L, uniform_stride=L)], m, keep=keep[k * m:])
        e.sync()
        h_keep.copy_(keep, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {n / dt / 1e6:.1f} Mreads/s  ({n * L / dt / 1e9:.1f} GB/s of sequence bytes over PCIe), {dt * 1e3:.1f} ms for {n} reads")
