#!/usr/bin/env python3
"""bench.py — Mreads/s of exact deduplication on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic 150-bp reads already
resident in HBM: empty the set, encode + insert every read, produce keep flags.
  N = 1 : BASELINE.json configs[1] — 100 M single-end 150 bp reads, ~20 % duplicates (the headline),
          and beside it in the same JSON line: configs[2] (100 M pairs 2x150 bp, `pe`), the
          PCIe-inclusive rate (`pcie_inclusive`: sequences start in pinned host memory) and the
          end-to-end rate of the CLI, FASTQ file in -> FASTQ file out (`end_to_end`; `.gz`: BGZF in, .gz out;
          `.ordinary_gzip`: a gzip file without member sizes in; `end_to_end_unordered` likewise for --unordered).
  N > 1 : weak scaling, the same per-GPU batch on every rank; reads are sharded by hash
          prefix with an all-to-all over RCCL (the library's shard group: csrc/fqd_shard.hip, fastq-dupaway_amd/shard.py).
  --config se|pe|sharded1 : only that device-phase measurement (sharded1 = the N > 1 path
          rehearsed on one rank under RCCL).
Launch: python bench.py [--gpus N --steps K --warmup W].  With N > 1 and no launcher around it the script
starts its own N ranks (child processes under torch.distributed.run, before any GPU call) and relays rank 0's
line; started under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one
of those ranks.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured copy ceiling
HBM_COPY_GBS = 6290.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["all", "se", "pe", "sharded1", "virtual"], default="all",
                    help="all (default, N=1): headline configs[1] + pe + pcie_inclusive + end_to_end; sharded1: the N > 1 path on one rank under RCCL; "
                         "virtual: --virtual-ranks ranks of the N > 1 path sharing this one GPU (peer copies)")
    ap.add_argument("--virtual-ranks", type=int, default=4)
    ap.add_argument("--round-reads", type=int, default=0, help="reads a rank brings to one round of the sharded exchange (0: 50 Mi — rounds large enough for the owners' partitioned insert — or all of them if fewer)")
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads (pairs) per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--paired", action="store_true", help="same as --config pe")
    ap.add_argument("--dup-permille", type=int, default=200)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--cpu-sample", type=int, default=8_000_000, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--pe-cpu-sample", type=int, default=2_000_000, help="pairs of the pe run checked against the CPU oracle")
    ap.add_argument("--e2e-reads", type=int, default=30_000_000, help="reads of the end-to-end CLI run (0 = skip)")
    ap.add_argument("--e2e-pairs", type=int, default=4_000_000, help="pairs of the end-to-end --unordered CLI run (0 = skip)")
    ap.add_argument("--e2e-dir", default="", help="where the end-to-end FASTQ files go (default: a temp dir under /tmp)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--send-hash", type=int, default=-1, help="sharded configs: 1 = every key's placement hash travels with it (FQD_SHARD_SEND_HASH), 0 = the owners hash arrived keys again; default: 1 from 4 ranks on")
    ap.add_argument("--no-final", action="store_true", help="A/B: plain fqd_submit instead of fqd_submit_final (the set is written back to HBM although nobody reads it)")
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
            "sample": f"first {n_sample} {'pairs' if paired else 'reads'} of the same synthetic workload, in-memory arrays "
                      f"(no file parsing or output), {dt:.1f} s; host has {os.cpu_count()} cores, the reference "
                      f"path is single-threaded"}, keep


def committed_traffic():
    """HBM bytes per launch from the newest committed PMC summary (two separate rocprofv3 --pmc passes of
    `python bench.py --config se`, corrected per MI355X_MICROARCH.md §HBM by tools/summarize_pmc.py).
    NOT measured in this run: the line names the file it comes from."""
    pmc = sorted((ROOT / "profiles").glob("*_pmc_summary.json"))
    if not pmc:
        return None, None
    return json.loads(pmc[-1].read_text())["kernels"], f"profiles/{pmc[-1].name}"


def device_phase(a, torch, dist, dev, local, rank, world, paired, sharded_mode, cpu_sample):
    """Times exactly a.steps steps of the hot path on device-resident input.  Returns the result
    dict (rank 0) plus the tensors the follow-up measurements reuse."""
    from fastq_dupaway_amd import Engine, Reads
    n, L, S = a.reads, a.read_len, (2 if paired else 1)
    bases = [torch.empty(n * L + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
    expect = torch.empty(n, dtype=torch.uint8, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)

    # Sharded runs (fastq-dupaway_amd/shard.py over csrc/fqd_shard.hip): a step's reads go through the exchange in
    # rounds of `m` reads per rank; the job's input order is (round, rank, position) — file blocks dealt round-robin
    # to the ranks — so round k of rank r holds the global indices below.  `virtual`: V ranks share this GPU.
    V = a.virtual_ranks if sharded_mode == "virtual" else 1
    ranks_here = V if sharded_mode == "virtual" else 1
    job_world = V if sharded_mode == "virtual" else world
    n_rank = n // V if sharded_mode == "virtual" else n            # reads per rank and step
    m = min(n_rank, a.round_reads or (50 << 20)) if sharded_mode else n
    rounds = -(-n_rank // m) if sharded_mode else 1
    spans = [(k * m, min(m, n_rank - k * m)) for k in range(rounds)]
    if sharded_mode:
        from fastq_dupaway_amd._lib import load_library
        cap = int(load_library().fqd_shard_slab_capacity(m, job_world, 0))
        own_reads = rounds * job_world * cap + 4096               # slab slots an owner takes in per step, unused ones included
    engines = [Engine(segments=S, device=local, capacity_reads=own_reads if sharded_mode else n,
                      capacity_bases=S * (own_reads if sharded_mode else n) * L, profile=True) for _ in range(ranks_here)]
    eng = engines[0]
    for v in range(ranks_here):
        for k, (lo, cnt) in enumerate(spans):
            r = v if sharded_mode == "virtual" else rank
            first = (k * job_world * m + r * cnt) if sharded_mode else 0
            at = v * n_rank + lo
            for mate in range(S):
                eng.synth_reads(a.seed, first, cnt, L, a.dup_permille, mate, bases[mate][at * L:],
                                expect[at:] if mate == S - 1 else None)
    eng.sync()
    segs = [Reads(bases[mate], uniform_len=L, uniform_stride=L) for mate in range(S)]

    sharded = None
    send_hash = False
    if sharded_mode:
        from fastq_dupaway_amd.shard import ShardGroup, unique_id
        uid = None
        if sharded_mode != "virtual":
            box = [unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)            # control plane only; the keys travel over the library's own RCCL communicator
            uid = box[0]
        # with the hash: 72 instead of 64 bytes per read on the links, 1.75 ms per 100 M reads less at the owners.  At 2 ranks ONE
        # link carries half of a GPU's keys and bounds the step, so fewer bytes win there; from 4 ranks on the links have room
        send_hash = (job_world >= 4) if a.send_hash < 0 else bool(a.send_hash)
        sharded = ShardGroup(engines, world=job_world, first_rank=0 if sharded_mode == "virtual" else rank, round_reads=m,
                             len0=L, len1=(L if S == 2 else 0), transport="copy" if sharded_mode == "virtual" else "rccl", uid=uid, send_hash=send_hash)
        round_args = []
        for lo, cnt in spans:
            sg = [[Reads(bases[mate][(v * n_rank + lo) * L:], uniform_len=L, uniform_stride=L) for mate in range(S)] for v in range(ranks_here)]
            round_args.append((sg, [cnt] * ranks_here, [keep[v * n_rank + lo:] for v in range(ranks_here)]))

        def step():
            for e in engines:
                e.reset()
            for sg, cnts, keeps in round_args:
                sharded.round(sg, cnts, keeps)
            sharded.flush()
    else:
        def step():
            # one batch per reset: the job IS its last batch, and says so (fqd_submit_final), like the CLI's resident runs do
            eng.reset()
            eng.submit(segs, n, keep=keep, final=not a.no_final)
            eng.sync()

    def fence():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
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
    for e in engines:
        e.reset_profile()
    shard_before = sharded.stats(0) if sharded else None
    fence()
    t0 = time.perf_counter()
    marks = []
    for _ in range(a.steps):
        step()
        marks.append(time.perf_counter())                          # (a step ends with the engine's sync: its own time is known, nothing is added to the region)
    fence()
    dt = time.perf_counter() - t0
    each = sorted((b - a_) * 1e3 for a_, b in zip([t0] + marks[:-1], marks))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = eng.profile()
    # what this rank's GPU did, for the per-GPU table of a sharded run
    mine = {"rank": rank, "device": local, "reads_per_step": n, "ms_per_step": round(dt / a.steps * 1e3, 3)}
    if sharded:
        st = sharded.stats(0)
        d = {k: st[k] - shard_before[k] for k in ("rounds", "overflow_rounds", "bytes_sent", "bytes_received", "exchange_ms")}
        per_round = max(1, d["rounds"])
        mine["exchange"] = {"transport": "RCCL: grouped ncclSend/ncclRecv on the library's own communicator" if st["transport"] == 0 else "peer copies (ranks share this process)",
                            "rccl_ranks": st["ranks_in_comm"], "slab_records": st["slab_records"], "rounds_per_step": rounds,
                            "overflow_rounds": d["overflow_rounds"],
                            "bytes_sent_per_round": d["bytes_sent"] // per_round, "bytes_received_per_round": d["bytes_received"] // per_round,
                            "forward_all_to_all_ms_per_round": round(d["exchange_ms"] / per_round, 3)}
        mine["kernels_ms_per_step"] = {k: round(prof[f"{k}_ms"] / a.steps, 3) for k in ("encode", "partition", "dedup", "insert", "other") if prof.get(f"{k}_ms")}
        gbps = n / (dt / a.steps) * S * L / 1e9
        mine["roofline"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 4),
                            "scope": "this GPU's share of the step: its reads x algorithmic bytes / step time"}
    per_gpu = [mine]
    if dist is not None and world > 1:
        per_gpu = [None] * world
        dist.all_gather_object(per_gpu, mine)

    res = None
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = world * n * a.steps / dt / 1e6
        bytes_per_unit = S * L                       # sequence bytes read once from HBM
        names = {"encode": "fqd::encode_staged_pe_kernel" if paired else "fqd::encode_staged_kernel", "insert": "fqd::insert_kernel",
                 "partition": "fqd::bulk_hist/scatter passes (4 kernels, timed as one group)", "dedup": "fqd::bucket_dedup_kernel"}
        kernels = {}
        for k in ("encode", "insert", "partition", "dedup"):
            if prof[f"{k}_launches"]:
                avg_ms = prof[f"{k}_ms"] / prof[f"{k}_launches"]
                kernels[k] = {"name": names[k], "avg_ms": round(avg_ms, 4), "launches": prof[f"{k}_launches"]}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        per_launch = prof[f"{dom}_reads"] / prof[f"{dom}_launches"]
        dom_gbps = per_launch * bytes_per_unit / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        step_gbps = value * 1e6 / world * bytes_per_unit / 1e9
        traffic, source, step_traffic = None, None, None
        if world == 1 and not sharded_mode and n == 100_000_000 and L == 150 and not paired:
            per_kernel, source = committed_traffic()
            if per_kernel:
                for name, rec in per_kernel.items():
                    for k in kernels:
                        if name.split("<")[0] == names[k].split("<")[0].split(" ")[0]:
                            kernels[k]["hbm_bytes_from_committed_profile"] = rec["hbm_traffic"]
                traffic = kernels[dom].get("hbm_bytes_from_committed_profile")
                step_traffic = sum(rec["hbm_traffic"] for name, rec in per_kernel.items() if "synth" not in name)
        # SURVEY §8(d) defines the contract figure over the whole device phase: reads/s x 150 B / 8 TB/s.
        # That is `frac`; the dominant kernel's own figure stands beside it.
        roofline = {"bound": "hbm", "achieved": round(step_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(step_gbps / HBM_PEAK_GBS, 4),
                    "frac_of_measured_copy_ceiling": round(step_gbps / HBM_COPY_GBS, 4),
                    "scope": "whole device phase (all kernels of a step): algorithmic bytes / step time",
                    "traffic": traffic, "traffic_scope": "dominant kernel, per launch" if traffic else None,
                    "traffic_whole_step": step_traffic, "traffic_source": (source + " (committed rocprofv3 --pmc passes of this command; not measured in this run)") if traffic else None,
                    "algorithmic_bytes_per_unit": bytes_per_unit,
                    "dominant_kernel": {"name": names[dom], "avg_ms": kernels[dom]["avg_ms"], "achieved": round(dom_gbps, 1),
                                        "frac": round(dom_gbps / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic bytes per launch / average launch duration (HIP events on the engine's stream)"},
                    "kernels": kernels}
        what = "pairs" if paired else "reads"
        res = {"value": round(value, 2), "unit": "Mpairs/s" if paired else "Mreads/s", "ms_per_step": round(ms_step, 3),
               "config": {"workload": ("100M paired-end 2x150 bp, --fast pair-hash" if paired and n == 100_000_000 else
                                       "100M single-end 150 bp FASTQ (~20% dups), --fast" if n == 100_000_000 else
                                       f"{n} {what} x {L} bp per GPU"),
                          "input": "150-byte sequence lines extracted from the FASTQ records, resident in HBM when the timed region starts",
                          "reads_per_gpu": n, "read_len": L, "dup_fraction": a.dup_permille / 1000.0,
                          "batches_per_step": 1 if not sharded_mode else rounds,
                          "last_batch_declared": bool(not sharded_mode and not a.no_final),
                          "sharding": "none" if not sharded_mode else
                                      (f"hash-prefix sharding over {job_world} rank(s)" + (f" sharing this GPU ({n_rank} reads each)" if sharded_mode == "virtual" else f" = {world} GPU(s), one process each")
                                       + f", fixed-size all-to-all slabs, {rounds} round(s) of {m} reads per rank and step"
                                       + (", hashes travel with the keys" if send_hash else ", owners hash arrived keys again"))},
               "parity": parity, "roofline": roofline,
               "step_ms_spread": {"min": round(each[0], 3), "median": round(each[len(each) // 2], 3), "max": round(each[-1], 3)}}
        if sharded_mode:
            res["per_gpu"] = per_gpu
        if world == 1 and cpu_sample > 0 and sharded_mode != "virtual":
            ms = min(cpu_sample, n)
            cb, cpu_keep = cpu_baseline(bases[0], ms, L, paired, bases[1] if S == 2 else None)
            import numpy as np
            if not np.array_equal(cpu_keep, keep[:ms].cpu().numpy()):
                sys.exit("GPU keep flags differ from the CPU oracle on the baseline sample — result invalid")
            res["cpu_baseline"] = cb
            res["parity"] += f"; == CPU oracle on the first {ms}"
    if sharded:
        sharded.close()
    for e in engines[1:]:
        e.close()
    return res, eng, bases, expect, keep


def pcie_inclusive(torch, eng, bases, expect, keep, n, L, reps=2):
    """Sequences start in PINNED HOST memory, are copied to HBM in chunks and deduplicated; flags come
    back to the host.  Never the bench `value`."""
    from fastq_dupaway_amd import Reads
    h_bases = torch.empty(n * L, dtype=torch.uint8).pin_memory()
    h_bases.copy_(bases[: n * L]); torch.cuda.synchronize()
    h_keep = torch.empty(n, dtype=torch.uint8).pin_memory()
    chunks = 8
    m = n // chunks
    best = None
    for _ in range(reps):
        eng.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(chunks):
            cnt = m if k + 1 < chunks else n - k * m
            bases[k * m * L:(k * m + cnt) * L].copy_(h_bases[k * m * L:(k * m + cnt) * L], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            eng.submit([Reads(bases[k * m * L:], uniform_len=L, uniform_stride=L)], cnt, keep=keep[k * m:])
        eng.sync()
        h_keep.copy_(keep, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ok = bool(torch.equal(keep, expect))
    del h_bases
    return {"value": round(n / best / 1e6, 1), "unit": "Mreads/s", "sequence_GBps_over_pcie": round(n * L / best / 1e9, 1),
            "what": f"{n} reads: 150-byte sequences in pinned host memory -> {chunks} hipMemcpyAsync chunks -> dedup -> flags back to the host; best of {reps}",
            "parity": "keep flags == closed form" if ok else "MISMATCH"}


def end_to_end(a, torch, bases, expect, L):
    """The CLI, FASTQ file in -> FASTQ file out, on the first --e2e-reads reads of the same workload."""
    import numpy as np
    from fastq_dupaway_amd import _lib
    n = min(a.e2e_reads, a.reads)
    seqs = bases[: n * L].cpu().numpy().reshape(n, L)
    exp_keep = expect[:n].cpu().numpy()
    d = Path(a.e2e_dir) if a.e2e_dir else Path(tempfile.mkdtemp(prefix="fqd_e2e_", dir="/tmp"))
    d.mkdir(parents=True, exist_ok=True)
    src, dst = d / "in.fq", d / "out.fq"
    rec_len = 12 + L + 1 + 2 + L + 1                         # "@r%09d\n" + seq\n + "+\n" + qual\n = 316 B at 150 bp
    with open(src, "wb") as f:
        step = 1_000_000
        for lo in range(0, n, step):
            cnt = min(step, n - lo)
            rec = np.empty((cnt, rec_len), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            idx = np.arange(lo, lo + cnt, dtype=np.int64)
            for p in range(9):
                rec[:, 10 - p] = 48 + (idx % 10); idx //= 10
            rec[:, 11] = 10
            rec[:, 12:12 + L] = seqs[lo:lo + cnt]
            rec[:, 12 + L] = 10; rec[:, 13 + L] = ord("+"); rec[:, 14 + L] = 10
            rec[:, 15 + L:15 + 2 * L] = ord("I"); rec[:, 15 + 2 * L] = 10
            f.write(rec.tobytes())
    size_in = src.stat().st_size
    runs = []
    out = None
    for _ in range(2):
        dst.unlink(missing_ok=True)
        t0 = time.perf_counter()
        r = subprocess.run([str(_lib.CLI_PATH), "-i", str(src), "-o", str(dst), "--fast", "-v"], capture_output=True, text=True)
        runs.append(time.perf_counter() - t0)
        out = r
        if r.returncode != 0:
            break
    dups = int((exp_keep == 0).sum())
    expected_line = f"{n} reads processed, out of which {dups} duplicates were removed.\n"
    ok = out.returncode == 0 and out.stdout == expected_line and dst.stat().st_size == (n - dups) * rec_len
    res = {"value": round(n / min(runs) / 1e6, 2), "unit": "Mreads/s", "seconds": [round(t, 3) for t in runs],
           "input_GBps": round(size_in / min(runs) / 1e9, 2),
           "what": f"fastq-dupaway -i in.fq -o out.fq --fast -v on {n} reads ({size_in / 1e9:.2f} GB FASTQ, 316 B/record), plain files on {d}, "
                   f"page cache warm (the input was just written), process start-up and output close included; best of 2 runs",
           "size_note": (f"{n} of the workload's {a.reads} reads: the full size needs about 70 GB of scratch files and four more minutes "
                         f"(python bench.py --e2e-reads {a.reads} [--e2e-dir DIR]; profiles/r03_bench_e2e_100m_reads.json is such a run)") if n < a.reads else "the workload's full size",
           "parity": "-v line and output size == closed form" if ok else f"MISMATCH rc={out.returncode} {out.stdout!r} {out.stderr[-300:]!r}"}
    # the CPU baseline SURVEY §8(d) specifies, beside it: the oracle's FILE driver (500 MiB block reader, record parser,
    # base-5/17 keys, std::unordered_set, find-then-insert, survivors written; reference hash_dup_remover.hpp:105-148) on
    # a prefix of the same input file, one thread.  Its output must be the head of the CLI's output, byte for byte.
    if ok and a.cpu_sample > 0:
        try:
            from oracle import binding
            m = min(a.cpu_sample, n)
            pre, pre_out = d / "prefix.fq", d / "prefix_out.fq"
            with open(src, "rb") as f, open(pre, "wb") as g:
                left = m * rec_len
                while left:
                    blk = f.read(min(left, 64 << 20)); g.write(blk); left -= len(blk)
            t0 = time.perf_counter()
            tot, dup = binding.load_oracle().filter_single(pre, pre_out)
            dt = time.perf_counter() - t0
            size = pre_out.stat().st_size
            same = tot == m and dup == int((exp_keep[:m] == 0).sum()) and size == (m - dup) * rec_len
            if same:
                with open(pre_out, "rb") as f, open(dst, "rb") as g:
                    while same:
                        x = f.read(64 << 20)
                        if not x:
                            break
                        same = x == g.read(len(x))
            res["cpu_baseline"] = {"value": round(m / dt / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port, file driver",
                                   "sample": f"first {m} records of the same FASTQ file ({m * rec_len / 1e9:.2f} GB), file in -> file out through the oracle's "
                                             f"restatement of filterSE (500 MiB block reader, parser, keys, unordered_set, writer), {dt:.1f} s, page cache warm; "
                                             f"host has {os.cpu_count()} cores, the reference path is single-threaded",
                                   "parity": "oracle output == head of the CLI's output, byte for byte" if same else "MISMATCH between the oracle's and the CLI's output"}
            pre.unlink(missing_ok=True); pre_out.unlink(missing_ok=True)
        except Exception as ex:
            res["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # the same reads as a BGZF file in and a .gz file out: inflated, cut into records, deduplicated and deflated on the GPU
    try:
        packer = Path("/tmp") / f"fqd_bgzf_pack_{os.getpid()}"
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(packer), str(ROOT / "tools" / "bgzf_pack.cpp"),
                        str(ROOT / "fastq-dupaway_amd" / "host" / "file_io.cpp"), "-lz", "-lpthread"], check=True, capture_output=True)
        gz_src, gz_dst = d / "in.fq.gz", d / "out.fq.gz"
        subprocess.run([str(packer), str(src), str(gz_src)], check=True, env=dict(os.environ, FQD_GZ_LEVEL="1"))
        gz_runs, rg = [], None
        for _ in range(2):
            gz_dst.unlink(missing_ok=True)
            t0 = time.perf_counter()
            rg = subprocess.run([str(_lib.CLI_PATH), "-i", str(gz_src), "-o", str(gz_dst), "--fast", "-v"], capture_output=True, text=True,
                                env={k: v for k, v in os.environ.items() if k not in ("FQD_GZ_LEVEL", "FQD_GZ_DEVICE", "FQD_GUNZIP_DEVICE", "FQD_ORDERED_RESIDENT")})
            gz_runs.append(time.perf_counter() - t0)
            if rg.returncode != 0:
                break
        inflated = subprocess.run(f"gzip -dc '{gz_dst}' | wc -c", shell=True, capture_output=True, text=True)
        same = rg.returncode == 0 and rg.stdout == expected_line and inflated.stdout.strip() == str((n - dups) * rec_len)
        res["gz"] = {"value": round(n / min(gz_runs) / 1e6, 2), "unit": "Mreads/s", "seconds": [round(t, 3) for t in gz_runs],
                     "what": f"the same reads as a BGZF file in ({gz_src.stat().st_size / 1e9:.2f} GB) and a .gz file out ({gz_dst.stat().st_size / 1e9:.2f} GB): "
                             "inflate, record scan, dedup and deflate on the GPU",
                     "parity": "-v line and inflated output size == closed form" if same else f"MISMATCH rc={rg.returncode} {rg.stdout!r} {rg.stderr[-300:]!r}"}
        packer.unlink(missing_ok=True)
    except Exception as ex:
        res["gz"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # and as an ORDINARY gzip file (no member sizes: what gzip, pigz and sequencers write) in.  Like its BGZF sibling above: .gz out —
    # the file goes to HBM as it lies on disk and is inflated there (fqd_gunzip: block starts guessed per unit, two decodes over made-up
    # windows, windows chained), dedup and deflate on the GPU.  Beside it: a plain file out, and the same with the host's several-thread
    # reader (host/pgzip.hpp, round 3's way) instead of the GPU's.
    try:
        og_src, og_dst, og_gz = d / "in_ordinary.fq.gz", d / "out_ordinary.fq", d / "out_ordinary.fq.gz"
        subprocess.run([sys.executable, str(ROOT / "tools" / "config4_at_size.py"), "--gzip-helper", str(src), str(og_src)], check=True)
        clean = {k: v for k, v in os.environ.items() if k not in ("FQD_GZ_LEVEL", "FQD_GZ_DEVICE", "FQD_GUNZIP_DEVICE", "FQD_ORDERED_RESIDENT", "FQD_GUNZIP_ORDINARY_DEVICE")}

        def run_once(dst, env):
            dst.unlink(missing_ok=True)
            t0 = time.perf_counter()
            r = subprocess.run([str(_lib.CLI_PATH), "-i", str(og_src), "-o", str(dst), "--fast", "-v"], capture_output=True, text=True, env=env)
            return r, time.perf_counter() - t0
        runs, ro = [], None
        for _ in range(2):
            ro, dt = run_once(og_gz, clean)
            runs.append(dt)
            if ro.returncode != 0:
                break
        gz_out = d / "out.fq.gz"                                           # the BGZF leg's output: the same survivors through the same deflater
        same = ro.returncode == 0 and ro.stdout == expected_line
        how = "output byte-identical to the BGZF leg's (whose inflated size == closed form)"
        if same and not (gz_out.exists() and subprocess.run(["cmp", "-s", str(og_gz), str(gz_out)]).returncode == 0):
            how = "inflated output size == closed form"
            same = subprocess.run(f"gzip -dc '{og_gz}' | wc -c", shell=True, capture_output=True, text=True).stdout.strip() == str((n - dups) * rec_len)
        res["ordinary_gzip"] = {"value": round(n / min(runs) / 1e6, 2), "unit": "Mreads/s", "seconds": [round(t, 3) for t in runs],
                                "what": f"the same reads as an ordinary gzip file in ({og_src.stat().st_size / 1e9:.2f} GB, members of 256 MB of text without BGZF fields) "
                                        "and a .gz file out, like the BGZF leg beside it: the file inflated on the GPU (fqd_gunzip), record scan, dedup and deflate there",
                                "parity": f"-v line == closed form, {how}" if same
                                          else f"MISMATCH rc={ro.returncode} {ro.stdout!r} {ro.stderr[-300:]!r}"}
        og_gz.unlink(missing_ok=True)
        for key, env, what in (("plain_out", clean, "the same input, a plain file out (5.4 GB through the page cache)"),
                               ("plain_out_host_reader", dict(clean, FQD_GUNZIP_ORDINARY_DEVICE="0"),
                                "the same, the input inflated by the host's several-thread reader (host/pgzip.hpp) in the streaming run: round 3's way")):
            r, dt = run_once(og_dst, env)
            ok = r.returncode == 0 and r.stdout == expected_line and og_dst.stat().st_size == (n - dups) * rec_len
            res["ordinary_gzip"][key] = {"value": round(n / dt / 1e6, 2), "unit": "Mreads/s", "seconds": [round(dt, 3)], "what": what,
                                         "parity": "-v line and output size == closed form" if ok else f"MISMATCH rc={r.returncode} {r.stdout!r} {r.stderr[-300:]!r}"}
    except Exception as ex:
        res.setdefault("ordinary_gzip", {})["error"] = f"{type(ex).__name__}: {ex}"[:300]
    if not a.e2e_dir:
        shutil.rmtree(d, ignore_errors=True)
    return res


def end_to_end_unordered(a, torch, bases, L):
    """The CLI on the configs[4] SHAPE at a size the oracle checks in seconds: paired FASTQ, file 2 shuffled,
    --unordered; outputs and -v lines compared byte for byte with the CPU oracle's file driver."""
    import filecmp
    import numpy as np
    from fastq_dupaway_amd import _lib
    from oracle import binding
    n = min(a.e2e_pairs, a.reads // 2)
    seqs = bases[: 2 * n * L].cpu().numpy().reshape(2 * n, L)
    d = Path(a.e2e_dir) if a.e2e_dir else Path(tempfile.mkdtemp(prefix="fqd_e2e_un_", dir="/tmp"))
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(a.seed)
    files = []
    for mate in range(2):
        idl = 18                                              # "@r%09d 1:N:0\n"
        rec = np.empty((n, idl + L + 1 + 2 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        idx = np.arange(n, dtype=np.int64)
        for p in range(9):
            rec[:, 10 - p] = 48 + (idx % 10); idx //= 10
        rec[:, 11:18] = np.frombuffer(b" %d:N:0\n" % (mate + 1), dtype=np.uint8)
        rec[:, idl:idl + L] = seqs[mate * n:(mate + 1) * n]
        rec[:, idl + L] = 10; rec[:, idl + L + 1] = ord("+"); rec[:, idl + L + 2] = 10
        rec[:, idl + L + 3:idl + 2 * L + 3] = ord("I"); rec[:, idl + 2 * L + 3] = 10
        if mate == 1:
            rec = rec[rng.permutation(n)]                      # file 2 in another order
        f = d / f"r{mate + 1}.fq"
        rec.tofile(f)
        files.append(f)
    outs = [d / "o1.fq", d / "o2.fq"]; exps = [d / "e1.fq", d / "e2.fq"]
    runs = []
    r = None
    for _ in range(2):
        for o in outs:
            o.unlink(missing_ok=True)
        t0 = time.perf_counter()
        r = subprocess.run([str(_lib.CLI_PATH), "-i", str(files[0]), "-u", str(files[1]), "-o", str(outs[0]), "-p", str(outs[1]),
                            "--fast", "--unordered", "-v"], capture_output=True, text=True, cwd=str(d))
        runs.append(time.perf_counter() - t0)
        if r.returncode != 0:
            break
    t0 = time.perf_counter()
    tot, dup, un = binding.load_oracle().filter_paired(files[0], files[1], exps[0], exps[1], binding.FASTQ, unordered=True, tail_rule=True)
    t_oracle = time.perf_counter() - t0
    line = f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n{un} Non-matching entries from both files were skipped.\n"
    ok = r.returncode == 0 and r.stdout == line and all(filecmp.cmp(o, e, shallow=False) for o, e in zip(outs, exps))
    size = sum(f.stat().st_size for f in files)
    res = {"value": round(n / min(runs) / 1e6, 3), "unit": "Mpairs/s", "seconds": [round(t, 3) for t in runs],
           "what": f"fastq-dupaway -i r1.fq -u r2.fq -o o1.fq -p o2.fq --fast --unordered -v on {n} pairs ({size / 1e9:.2f} GB of FASTQ, file 2 shuffled; "
                   f"one pass, text resident in HBM, host memory within the default --mem-limit), plain files on {d}, page cache warm; best of 2 runs",
           "cpu_oracle_seconds": round(t_oracle, 2),
           "parity": "output bytes and -v lines == CPU oracle's file driver" if ok else f"MISMATCH rc={r.returncode} {r.stdout!r} {r.stderr[-300:]!r}"}
    # the same job as configs[4] has it: BGZF in, .gz out — inflated, cut into records and deflated on the GPU
    try:
        packer = Path("/tmp") / f"fqd_bgzf_pack_{os.getpid()}"        # (an --e2e-dir on /dev/shm is mounted noexec)
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(packer), str(ROOT / "tools" / "bgzf_pack.cpp"),
                        str(ROOT / "fastq-dupaway_amd" / "host" / "file_io.cpp"), "-lz", "-lpthread"], check=True, capture_output=True)
        gz_in = [d / "r1.fq.gz", d / "r2.fq.gz"]; gz_out = [d / "o1.fq.gz", d / "o2.fq.gz"]
        for src, dst in zip(files, gz_in):
            subprocess.run([str(packer), str(src), str(dst)], check=True, env=dict(os.environ, FQD_GZ_LEVEL="1"))
        gz_runs = []
        for _ in range(2):
            for o in gz_out:
                o.unlink(missing_ok=True)
            t0 = time.perf_counter()
            rg = subprocess.run([str(_lib.CLI_PATH), "-i", str(gz_in[0]), "-u", str(gz_in[1]), "-o", str(gz_out[0]), "-p", str(gz_out[1]),
                                 "--fast", "--unordered", "-v"], capture_output=True, text=True, cwd=str(d),
                                env={k: v for k, v in os.environ.items() if k not in ("FQD_GZ_LEVEL", "FQD_GZ_DEVICE", "FQD_GUNZIP_DEVICE")})
            gz_runs.append(time.perf_counter() - t0)
            if rg.returncode != 0:
                break
        same = rg.returncode == 0 and rg.stdout == line and all(
            subprocess.run(f"gzip -dc '{o}' | cmp -s - '{e}'", shell=True).returncode == 0 for o, e in zip(gz_out, exps))
        res["gz"] = {"value": round(n / min(gz_runs) / 1e6, 3), "unit": "Mpairs/s", "seconds": [round(t, 3) for t in gz_runs],
                     "what": "the same pairs as BGZF files in (libdeflate level 1) and .gz out: inflate, record scan and deflate on the GPU "
                             f"({sum(f.stat().st_size for f in gz_in) / 1e9:.2f} GB in, {sum(f.stat().st_size for f in gz_out) / 1e9:.2f} GB out)",
                     "parity": "gzip -dc of both outputs == the CPU oracle's outputs, -v lines equal" if same
                               else f"MISMATCH rc={rg.returncode} {rg.stdout!r} {rg.stderr[-300:]!r}"}
        packer.unlink(missing_ok=True)
    except Exception as ex:                                   # no compiler, no room: the plain-file leg stands
        res["gz"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # and as PLAIN gzip — one member per file, what `gzip`, `pigz` and the reference itself write (file_utils.cpp:83-92): no
    # reader can split such a file, so each input is inflated by ONE host thread (zlib) while the blocks it yields are scanned,
    # uploaded and joined as they come; BGZF is what the numbers above are about, THIS is what files in the wild look like
    try:
        pz_in = [d / "r1.plain.fq.gz", d / "r2.plain.fq.gz"]; pz_out = [d / "o1.plain.fq.gz", d / "o2.plain.fq.gz"]
        packs = [subprocess.Popen(f"gzip -1 -c '{src}' > '{dst}'", shell=True) for src, dst in zip(files, pz_in)]
        if any(p.wait() != 0 for p in packs):
            raise RuntimeError("gzip -1 failed")
        for o in pz_out:
            o.unlink(missing_ok=True)
        t0 = time.perf_counter()
        rp = subprocess.run([str(_lib.CLI_PATH), "-i", str(pz_in[0]), "-u", str(pz_in[1]), "-o", str(pz_out[0]), "-p", str(pz_out[1]),
                             "--fast", "--unordered", "-v"], capture_output=True, text=True, cwd=str(d),
                            env={k: v for k, v in os.environ.items() if k not in ("FQD_GZ_LEVEL", "FQD_GZ_DEVICE", "FQD_GUNZIP_DEVICE")})
        t_plain = time.perf_counter() - t0
        same = rp.returncode == 0 and rp.stdout == line and all(
            subprocess.run(f"gzip -dc '{o}' | cmp -s - '{e}'", shell=True).returncode == 0 for o, e in zip(pz_out, exps))
        res["plain_gzip"] = {"value": round(n / t_plain / 1e6, 3), "unit": "Mpairs/s", "seconds": [round(t_plain, 3)],
                             "what": f"the same pairs as single-member gzip files (`gzip -1`, {sum(f.stat().st_size for f in pz_in) / 1e9:.2f} GB in, no member sizes): "
                                     "inflated on the GPU from guessed block starts (fqd_gunzip); outputs deflated on the GPU as above",
                             "parity": "gzip -dc of both outputs == the CPU oracle's outputs, -v lines equal" if same
                                       else f"MISMATCH rc={rp.returncode} {rp.stdout!r} {rp.stderr[-300:]!r}"}
    except Exception as ex:
        res["plain_gzip"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if not a.e2e_dir:
        shutil.rmtree(d, ignore_errors=True)
    return res


def launch_command(gpus, argv, port):
    """The command `python bench.py --gpus N` (N > 1) turns itself into: one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def rank0_line(stdout_text):
    """The ONE JSON line rank 0 printed, picked out of whatever else the ranks wrote to stdout."""
    for line in reversed(stdout_text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                rec = json.loads(line)
            except ValueError:
                continue
            if "metric" in rec and "value" in rec:
                return line
    return None


def relaunch_as_ranks(a):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes — before this
    process has imported torch or made any GPU call (a process that has initialised the GPU must never exec
    or be replaced) — and relay rank 0's JSON line and the children's exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = launch_command(a.gpus, sys.argv[1:], port)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = rank0_line(r.stdout)
    if line is None:
        sys.stdout.write(r.stdout)
        sys.exit(r.returncode if r.returncode else 1)
    print(line, flush=True)
    sys.exit(r.returncode)


def selftest_ranks(a):
    """FQD_BENCH_SELFTEST=1: the launch path without a GPU — every rank joins a gloo group, the ranks agree on
    their number, rank 0 prints a line of the bench's shape.  tests/test_bench_launch.py runs this on the CPU."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    print(f"rank {rank} chatter on stdout {{not json}}", flush=True)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "selftest", "value": 0.0, "unit": "Mreads/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ranks_seen": int(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_as_ranks(a)                                   # never returns
    if os.environ.get("FQD_BENCH_SELFTEST") == "1":
        return selftest_ranks(a)
    import torch
    import fastq_dupaway_amd as fqd  # noqa: F401  (loads the HIP library or fails loudly)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: the two must agree")
    torch.cuda.set_device(local)
    dist = None
    config = a.config if a.config in ("sharded1", "virtual") else ("pe" if a.paired else a.config)
    sharded_mode = False
    if world > 1:
        sharded_mode = "ranks"
    elif config == "sharded1" or os.environ.get("FQD_BENCH_FORCE_SHARDED") == "1":
        sharded_mode = "ranks"                                  # the N > 1 path rehearsed on one rank: RCCL carries a self exchange
    elif config == "virtual":
        sharded_mode = "virtual"
    if world > 1:
        # torch.distributed is the control plane (the group's unique id, barriers, the max over ranks of the step time):
        # gloo.  The data plane — every key and every flag — is RCCL over xGMI through the library's own communicator
        # (csrc/fqd_shard.hip: ncclCommInitRank + grouped ncclSend/ncclRecv), the same code the CLI's FQD_DEVICES run uses.
        import torch.distributed as dist
        dist.init_process_group("gloo")
    dev = torch.device("cuda", local)
    L = a.read_len

    headline_paired = config == "pe" or a.paired
    res, eng, bases, expect, keep = device_phase(a, torch, dist, dev, local, rank, world, headline_paired, sharded_mode, a.cpu_sample)
    out = None
    if rank == 0:
        out = {"metric": "Mreads/s dedup, 150 bp %s FASTQ (device phase: sequences resident in HBM)" % ("PE" if headline_paired else "SE"),
               "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic", "config": res["config"], "parity": res["parity"], "roofline": res["roofline"],
               "step_ms_spread": res["step_ms_spread"]}
        if "cpu_baseline" in res:
            out["cpu_baseline"] = res["cpu_baseline"]
        if "per_gpu" in res:
            out["per_gpu"] = res["per_gpu"]
    if config == "all" and world == 1 and not sharded_mode:
        # the companions of the headline: a failure in one of them (a full /tmp, say) is reported in its
        # place and never costs the headline line
        def attempt(name, fn):
            try:
                out[name] = fn()
            except (Exception, SystemExit) as ex:      # SystemExit: a parity check of that leg failed
                out[name] = {"error": f"{type(ex).__name__}: {ex}"[:500]}
        attempt("pcie_inclusive", lambda: pcie_inclusive(torch, eng, bases[0], expect, keep, a.reads, L))
        if a.e2e_reads > 0:
            attempt("end_to_end", lambda: end_to_end(a, torch, bases[0], expect, L))
        if a.e2e_pairs > 0:
            attempt("end_to_end_unordered", lambda: end_to_end_unordered(a, torch, bases[0], L))
        eng.close()
        del bases, expect, keep
        torch.cuda.empty_cache()

        def pe_leg():
            # configs[2] beside the headline: 100 M pairs, same step definition, oracle-checked on a sample
            pe, eng2, b2, e2, k2 = device_phase(a, torch, None, dev, local, rank, world, True, False, a.pe_cpu_sample)
            leg = {k: pe[k] for k in ("value", "unit", "ms_per_step", "config", "parity", "cpu_baseline") if k in pe}
            leg["roofline"] = {k: pe["roofline"][k] for k in ("achieved", "frac", "algorithmic_bytes_per_unit", "dominant_kernel", "kernels")}
            eng2.close()
            return leg
        attempt("pe", pe_leg)
    else:
        eng.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None and world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
