#!/usr/bin/env python3
"""Skew probe: 50 M synthetic reads of which a given fraction is overwritten with poly-G (one key
repeated millions of times, as 2-colour sequencers produce), timed through the engine.
  python tools/heavy_key_probe.py 0.3          [FQD_BULK_MIN=-1 for the atomic path]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from fastq_dupaway_amd import Engine, Reads
n, L = 50_000_000, 150
dev = torch.device('cuda', 0)
bases = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
keep = torch.empty(n, dtype=torch.uint8, device=dev)
with Engine(segments=1, capacity_reads=n) as e:
    e.synth_reads(5, 0, n, L, 100, 0, bases, None); e.sync()
    g = torch.Generator(device=dev); g.manual_seed(1)
    mask = torch.rand(n, device=dev, generator=g) < float(sys.argv[1])
    bases[: n * L].view(n, L)[mask] = ord('G')
    torch.cuda.synchronize()
    for rep in range(3):
        e.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        e.submit([Reads(bases, uniform_len=L, uniform_stride=L)], n, keep=keep); e.sync()
        dt = time.perf_counter() - t0
        print(f"heavy fraction {sys.argv[1]}: {dt*1e3:.1f} ms, kept {int(keep.sum())}, polyG reads {int(mask.sum())}")
