#!/usr/bin/env python3
"""Times fqd_bgzf_deflate / fqd_bgzf_inflate / fqd_scan_records on FASTQ text built on the device.
  python tools/inflate_probe.py [--records 1900000] [--reps 3]"""
import argparse
import struct
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def walk(raw):
    rows, at, out = [], 0, 0
    while at < len(raw):
        total = struct.unpack_from("<H", raw, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", raw, at + total - 8)
        if isize:
            rows.append((at + 18, total - 26, out, isize, crc)); out += isize
        at += total
    a = np.array(rows, dtype=np.uint64).reshape(-1, 5)
    return [a[:, 0].copy(), a[:, 1].astype(np.uint32), a[:, 2].copy(), a[:, 3].astype(np.uint32), a[:, 4].astype(np.uint32)], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_900_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--host-packed", action="store_true", help="also inflate the same text packed by the host codec (libdeflate / zlib level 1: many more matches)")
    ap.add_argument("--quality", choices=["mixed", "flat", "binned"], default="mixed", help="flat: every quality 'I' (long runs, as the configs[4] generator writes); "
                    "binned: nine in ten 'F', the rest ':', ',' or '#' (what current sequencers write)")
    ap.add_argument("--gz-level", default="1", help="--host-packed: the level the host codec packs at (FQD_GZ_LEVEL; 6 is what bgzip uses)")
    a = ap.parse_args()
    import torch
    from fastq_dupaway_amd import Engine
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(6)
    n, L = a.records, 150
    rec = torch.empty((n, 18 + L + 3 + L + 1), dtype=torch.uint8, device=dev)
    x = torch.arange(n, device=dev, dtype=torch.int64)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    for p in range(9):
        rec[:, 10 - p] = (48 + x % 10).to(torch.uint8); x = x // 10
    rec[:, 11:18] = torch.tensor(list(b" 1:N:0\n"), dtype=torch.uint8, device=dev)
    rec[:, 18:18 + L] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, L), device=dev, generator=g)]
    rec[:, 18 + L] = 10; rec[:, 19 + L] = ord("+"); rec[:, 20 + L] = 10
    if a.quality == "flat":
        rec[:, 21 + L:21 + 2 * L] = ord("I")
    elif a.quality == "binned":
        q = torch.tensor(list(b":,#"), dtype=torch.uint8, device=dev)[torch.randint(0, 3, (n, L), device=dev, generator=g)]
        rec[:, 21 + L:21 + 2 * L] = torch.where(torch.rand((n, L), device=dev, generator=g) < 0.9, torch.full_like(q, ord("F")), q)
    else:
        rec[:, 21 + L:21 + 2 * L] = torch.tensor(list(b"FFFFFFFF:,#"), dtype=torch.uint8, device=dev)[torch.randint(0, 11, (n, L), device=dev, generator=g)]
    rec[:, 21 + 2 * L] = 10
    src = rec.reshape(-1); nbytes = src.numel()
    with Engine(segments=1, device=0) as e:
        dst = torch.empty(e.bgzf_bound(nbytes), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for _ in range(a.reps):
            t0 = time.perf_counter(); size = e.bgzf_deflate(src, nbytes, dst, 4); dt = time.perf_counter() - t0
            print(f"deflate {nbytes / 1e6:.0f} MB -> {size / 1e6:.0f} MB: {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s", flush=True)
        raw = dst[:size].cpu().numpy().tobytes()
        arrs, total = walk(raw)
        t = lambda v: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v.view(np.int32)).to(dev)
        args = [t(v) for v in arrs]
        text = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for _ in range(a.reps):
            t0 = time.perf_counter(); bad = e.bgzf_inflate(dst, *args, len(arrs[0]), text); dt = time.perf_counter() - t0
            print(f"inflate {len(arrs[0])} members -> {total / 1e6:.0f} MB: {dt * 1e3:.1f} ms = {total / dt / 1e9:.1f} GB/s, bad {bad}, equal {bool(torch.equal(text[:total], src))}", flush=True)
        if a.host_packed:
            import os, subprocess, tempfile
            d = tempfile.mkdtemp(prefix="fqd_probe_", dir="/tmp")
            packer = os.path.join(d, "pack")
            subprocess.run(["g++", "-O2", "-std=c++17", "-o", packer, str(ROOT / "tools" / "bgzf_pack.cpp"),
                            str(ROOT / "fastq-dupaway_amd" / "host" / "file_io.cpp"), "-lz", "-lpthread"], check=True)
            src.cpu().numpy().tofile(os.path.join(d, "t.fq"))
            subprocess.run([packer, os.path.join(d, "t.fq"), os.path.join(d, "t.fq.gz")], check=True, env=dict(os.environ, FQD_GZ_LEVEL=a.gz_level))
            raw = open(os.path.join(d, "t.fq.gz"), "rb").read()
            arrs, total = walk(raw)
            comp = torch.frombuffer(bytearray(raw + bytes(16)), dtype=torch.uint8).to(dev)
            args = [t(v) for v in arrs]
            text.zero_(); torch.cuda.synchronize()
            for _ in range(a.reps):
                t0 = time.perf_counter(); bad = e.bgzf_inflate(comp, *args, len(arrs[0]), text); dt = time.perf_counter() - t0
                print(f"inflate (host-packed, {len(raw) / 1e6:.0f} MB) {len(arrs[0])} members -> {total / 1e6:.0f} MB: {dt * 1e3:.1f} ms = {total / dt / 1e9:.1f} GB/s, "
                      f"bad {bad}, equal {bool(torch.equal(text[:total], src))}", flush=True)
            import shutil; shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
