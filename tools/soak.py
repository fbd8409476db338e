#!/usr/bin/env python3
"""Soak: many seeds / duplicate rates / sizes through the engine on one GPU, every result compared
with the generator's closed-form flags.  Looks for rare scheduling-dependent errors that a single
parity run could miss.   python tools/soak.py [--rounds 40] [--paired]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--reads", type=int, default=20_000_000)
    a = ap.parse_args()
    import torch
    from fastq_dupaway_amd import Engine, Reads
    S, L, n_max = (2 if a.paired else 1), 150, a.reads
    dev = torch.device("cuda", 0)
    bases = [torch.empty(n_max * L + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
    expect = torch.empty(n_max, dtype=torch.uint8, device=dev)
    keep = torch.empty(n_max, dtype=torch.uint8, device=dev)
    bad = 0
    t0 = time.time()
    with Engine(segments=S, capacity_reads=n_max) as e:
        for r in range(a.rounds):
            seed = 1000 + r
            dup = [0, 50, 200, 500, 900][r % 5]
            n = n_max - (r * 7919) % (n_max // 3)
            for m in range(S):
                e.synth_reads(seed, 0, n, L, dup, m, bases[m], expect if m == S - 1 else None)
            e.reset()
            keep.zero_()
            pieces = 1 if r % 3 else 3                       # some rounds arrive as three batches
            at = 0
            for p in range(pieces):
                k = n // pieces if p + 1 < pieces else n - at
                e.submit([Reads(bases[m][at * L:], uniform_len=L, uniform_stride=L) for m in range(S)], k, keep=keep[at:])
                at += k
            e.sync()
            ok = bool(torch.equal(keep[:n], expect[:n]))
            st = e.stats()
            ok = ok and st["duplicates"] == int((expect[:n] == 0).sum().item())
            bad += not ok
            print(f"round {r}: n={n} dup={dup / 10:.0f}% pieces={pieces} {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{a.rounds} rounds, {bad} mismatches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
