#!/usr/bin/env python3
"""bench.py — Mreads/s of exact deduplication on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic 150-bp reads already
resident in HBM: empty the set, encode + insert every read, produce keep flags.
  N = 1 : BASELINE.json configs[1] — 100 M single-end 150 bp reads, ~20 % duplicates.
  N > 1 : weak scaling, the same per-GPU batch on every rank; reads are sharded by hash
          prefix with an all-to-all over RCCL (fastq-dupaway_amd/sharded.py).
Launch: python bench.py [--gpus N --steps K --warmup W]; for N>1 under
python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads (pairs) per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--paired", action="store_true", help="configs[2]: 2x150 bp pair-hash")
    ap.add_argument("--dup-permille", type=int, default=200)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--cpu-sample", type=int, default=8_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args()


def cpu_baseline(bases, n_sample, L, paired, bases2):
    """The oracle (CPU restatement of the reference's --fast loop: base-5 17-mer keys,
    std::unordered_set, find-then-insert) timed single-threaded on a prefix of the SAME
    workload.  Reported beside the GPU number; never the thing measured."""
    import numpy as np
    from oracle import binding
    oracle = binding.load_oracle()
    host = bases[: n_sample * L].cpu().numpy()
    offs = np.arange(n_sample, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n_sample, L, np.uint32)
    t0 = time.perf_counter()
    if paired:
        host2 = bases2[: n_sample * L].cpu().numpy()
        keep = oracle.dedup_paired(host, offs, lens, host2, offs, lens)
    else:
        keep = oracle.dedup_single(host, offs, lens)
    dt = time.perf_counter() - t0
    return {"value": round(n_sample / dt / 1e6, 4), "unit": "Mpairs/s" if paired else "Mreads/s", "cores": 1, "kind": "port",
            "sample": f"first {n_sample} reads of the same synthetic workload, in-memory arrays "
                      f"(no file parsing or output), {dt:.1f} s; host has {os.cpu_count()} cores, the reference "
                      f"path is single-threaded"}, keep


def main():
    a = parse()
    import torch
    import fastq_dupaway_amd as fqd
    from fastq_dupaway_amd import Engine, Reads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    dist = None
    force_sharded = os.environ.get("FQD_BENCH_FORCE_SHARDED") == "1"      # rehearse the N>1 path on one GPU
    if world > 1 or force_sharded:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    n, L, S = a.reads, a.read_len, (2 if a.paired else 1)
    dev = torch.device("cuda", local)
    bases = [torch.empty(n * L + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
    expect = torch.empty(n, dtype=torch.uint8, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)

    eng = Engine(segments=S, device=local, capacity_reads=n, capacity_bases=S * n * L, profile=True)
    sharded_mode = dist is not None
    # Sharded runs move a step's reads in rounds sized so that no rank-to-rank message of the
    # all-to-all exceeds 256 MiB (64-byte keys for 150 bp): 3 rounds per step on 8 GPUs, enough for
    # the round pipeline to hide most of the exchange.  Measured on this image (RCCL 2.26.6, torch
    # 2.10): a single all_to_all_single message above 1 GiB arrives corrupted.  The
    # job's input order is (round, rank, position) — file blocks dealt round-robin to the ranks —
    # so round k of rank r holds the global indices below.
    rec_bytes = 8 * (eng.key_words(L, L if S == 2 else 0) + (1 if os.environ.get("FQD_SHARDED_WITH_HASH") == "1" else 0))   # keys travel without their hash
    lazy = sharded_mode and os.environ.get("FQD_SHARDED_LAZY") == "1"       # hashes first, keys only for candidates
    if lazy:
        rec_bytes = 16
    rounds = max(1, -(-(n * rec_bytes // world) // (256 << 20))) if sharded_mode else 1
    if sharded_mode and os.environ.get("FQD_BENCH_ROUNDS"):
        rounds = int(os.environ["FQD_BENCH_ROUNDS"])
    m = -(-n // rounds)
    spans = [(k * m, min(m, n - k * m)) for k in range(rounds)]
    for k, (lo, cnt) in enumerate(spans):
        first = (k * world * m + rank * cnt) if sharded_mode else 0
        for mate in range(S):
            eng.synth_reads(a.seed, first, cnt, L, a.dup_permille, mate, bases[mate][lo * L:],
                            expect[lo:] if mate == S - 1 else None)
    eng.sync()
    segs = [Reads(bases[mate], uniform_len=L, uniform_stride=L) for mate in range(S)]

    if lazy:
        from fastq_dupaway_amd.sharded import LazyShardedDedup
        owner = Engine(segments=1, device=local, capacity_reads=int(n * 1.1))
        sharded = LazyShardedDedup(eng, owner, dist, dev, n_max=m, len0=L, len1=(L if S == 2 else 0))

        def step():
            eng.reset(); owner.reset(); sharded.reset()
            for lo, cnt in spans:
                sharded.dedup([Reads(bases[mate][lo * L:], uniform_len=L, uniform_stride=L) for mate in range(S)], cnt, keep[lo:])
            eng.sync()
    elif sharded_mode:
        from fastq_dupaway_amd.sharded import HipOps, ShardedDedup
        sharded = ShardedDedup(HipOps(eng), dist, dev, n_max=m, len0=L, len1=(L if S == 2 else 0))

        round_list = [([Reads(bases[mate][lo * L:], uniform_len=L, uniform_stride=L) for mate in range(S)], cnt, keep[lo:])
                      for lo, cnt in spans]

        def step():
            eng.reset()
            sharded.dedup_rounds(round_list)
            eng.sync()
    else:
        def step():
            eng.reset()
            eng.submit(segs, n, keep=keep)
            eng.sync()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    parity = "skipped"
    if not a.no_verify:
        # world == 1: closed-form flags of the generator.  world > 1: copies may point at other
        # ranks' reads, the closed form still holds globally (first occurrence = the non-copy).
        ok = bool(torch.equal(keep, expect)) if a.warmup > 0 else None
        if ok is False:
            sys.exit(f"rank {rank}: keep flags differ from the generator's closed form — result invalid")
        parity = "keep flags == closed-form flags of the generator on every rank" if ok else "skipped (no warmup step)"
    eng.reset_profile()
    if lazy and sharded.timing is not None:
        sharded.timing.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.profile()
    if lazy and rank == 0 and sharded.timing is not None:
        print({k: round(v / a.steps, 3) for k, v in sharded.timing.items()}, file=sys.stderr)

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * n * a.steps / dt / 1e6
        bytes_per_unit = S * L                       # sequence bytes read once from HBM
        kernels = {}
        names = {"encode": "fqd::encode_staged_kernel", "insert": "fqd::insert_kernel",
                 "partition": "fqd::bulk_hist/scatter passes (4 kernels, timed as one group)", "dedup": "fqd::bucket_dedup_kernel"}
        for k in ("encode", "insert", "partition", "dedup"):
            if prof[f"{k}_launches"]:
                avg_ms = prof[f"{k}_ms"] / prof[f"{k}_launches"]
                per_launch = prof[f"{k}_reads"] / prof[f"{k}_launches"]
                kernels[k] = {"avg_ms": round(avg_ms, 4), "launches": prof[f"{k}_launches"],
                              "GBps": round(per_launch * bytes_per_unit / (avg_ms * 1e-3) / 1e9, 1)}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (two separate
        # rocprofv3 --pmc runs of this same command, corrected per MI355X_MICROARCH.md §HBM:
        # tools/summarize_pmc.py).  Only valid for the workload those passes were taken on.
        traffic = None
        pmc = sorted((ROOT / "profiles").glob("*_pmc_summary.json"))
        if pmc and world == 1 and n == 100_000_000 and L == 150 and not a.paired:
            for name, rec in json.loads(pmc[-1].read_text())["kernels"].items():
                if name.split("<")[0] == names[dom].split("<")[0].split(" ")[0]:
                    traffic = rec["hbm_traffic"]
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_unit": bytes_per_unit, "kernels": kernels,
                    "whole_step_frac": round(value * 1e6 / world * bytes_per_unit / 1e9 / HBM_PEAK_GBS, 4)}
        out = {"metric": "Mreads/s dedup, 150 bp %s FASTQ" % ("PE" if a.paired else "SE"),
               "value": round(value, 2), "unit": "Mreads/s" if not a.paired else "Mpairs/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
               "data": "synthetic",
               "config": {"workload": ("100M paired-end 2x150 bp, --fast pair-hash" if a.paired and n == 100_000_000 else
                                       "100M single-end 150 bp FASTQ (~20% dups), --fast" if n == 100_000_000 else
                                       f"{n} {'pairs' if a.paired else 'reads'} x {L} bp per GPU"),
                          "reads_per_gpu": n, "read_len": L, "dup_fraction": a.dup_permille / 1000.0,
                          "sharding": "none" if not sharded_mode else f"hash-prefix all-to-all over {world} GPU(s), {rounds} round(s) per step"
                                      + (", hashes first / keys for candidates only" if lazy else "")},
               "parity": parity, "roofline": roofline}
        if world == 1 and a.cpu_sample > 0:
            m = min(a.cpu_sample, n)
            cb, cpu_keep = cpu_baseline(bases[0], m, L, a.paired, bases[1] if S == 2 else None)
            import numpy as np
            if not np.array_equal(cpu_keep, keep[:m].cpu().numpy()):
                sys.exit("GPU keep flags differ from the CPU oracle on the baseline sample — result invalid")
            out["cpu_baseline"] = cb
            out["parity"] += f"; == CPU oracle on the first {m}"
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
