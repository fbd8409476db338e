"""The multi-GPU exchange (csrc/fqd_shard.hip) on the CPU, as far as it goes without a GPU: the geometry of its
fixed-size slabs, spills and flag messages (csrc/fqd_shard_plan.hpp — the functions the device code itself uses).
  * tests/native/shard_plan_check plays whole rounds in one process, for 1-8 ranks, overflows included;
  * under REAL torch.distributed (gloo, world_size 2 and 3, separate processes) the ranks move real bytes by the same
    functions: one all_to_all_single of fixed-size slabs queued before any count is known, the counts beside it,
    exactly sized point-to-point spills, flags back the same way.  The owners' "insert" is a Python set here (the
    product has no CPU path); the flags must equal the CPU oracle's on the global input order (round, rank, position).
The GPU side of the same path: tests/test_shard.py, tests/test_cli.py::test_multi_gpu_cli_*."""
import ctypes as C
import os
import socket
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "shard_plan_check.cpp"
EXE = HERE / "native" / "shard_plan_check"
SO = HERE / "native" / "libshard_plan.so"
HDR = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_shard_plan.hpp"
L = 40


def _build():
    newest = max(SRC.stat().st_mtime, HDR.stat().st_mtime)
    if not EXE.exists() or EXE.stat().st_mtime < newest:
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", str(EXE), str(SRC)], check=True)
    if not SO.exists() or SO.stat().st_mtime < newest:
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DSHARD_PLAN_NO_MAIN", "-o", str(SO), str(SRC)], check=True)


@pytest.mark.parametrize("ranks", [1, 2, 3, 4, 8])
def test_slab_plan_rounds_on_the_cpu(ranks):
    _build()
    spilled = 0
    for seed in range(1, 40):
        r = subprocess.run([str(EXE), str(ranks), str(seed)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok "), (seed, r.stdout, r.stderr)
        spilled += int(r.stdout.split()[2])
    assert ranks == 1 or spilled > 0                                    # overflows did happen
    r = subprocess.run([str(EXE), str(ranks), "7", "8"], capture_output=True, text=True)     # tiny slabs: nearly everything spills
    assert r.returncode == 0, r.stdout


def _plan():
    lib = C.CDLL(str(SO))
    u64, u32, p64 = C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)
    lib.plan_slab_slot.argtypes = [u32, u64]; lib.plan_slab_slot.restype = u64
    lib.plan_spill_slot.argtypes = [p64, u32, u32, u64]; lib.plan_spill_slot.restype = u64
    lib.plan_owner_is_compact.argtypes = [p64, u32, u64]; lib.plan_owner_is_compact.restype = C.c_int
    lib.plan_owner_offset.argtypes = [p64, u32, u32, u64]; lib.plan_owner_offset.restype = u64
    lib.plan_owner_records.argtypes = [p64, u32, u64]; lib.plan_owner_records.restype = u64
    return lib


def test_sub_slab_geometry():
    """Slabs cut into one sub-slab per chunk of the input (fqd_plan::geometry): chunks of whole tiles covering the round, room
    for a fair share plus 5.5 sigma, no slack for a group of one; classic_count reads a slab filled from its first slot on as
    full / partial / empty sub-slabs and shows a spill in the last one; a spill takes further sub-slabs at an owner."""
    _build()
    P = _plan()
    P.plan_geometry.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    P.plan_classic_count.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]; P.plan_classic_count.restype = C.c_uint64
    P.plan_owner_sub_slabs.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64]; P.plan_owner_sub_slabs.restype = C.c_uint64
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8, 16):
        for reads in (1, 300, 4096, 50_000, 8 << 20, 50 << 20, 100_000_000):
            cr, ch, sc = C.c_uint64(), C.c_uint32(), C.c_uint64()
            P.plan_geometry(reads, world, 0, C.byref(cr), C.byref(ch), C.byref(sc))
            chunk, chunks, sub = cr.value, ch.value, sc.value
            assert chunk % 256 == 0 and chunk >= 4096 and chunks * chunk >= reads and (chunks - 1) * chunk < reads and chunks <= 1024
            if world == 1:
                assert sub == chunk
            else:
                fair = -(-chunk // world)
                assert fair + 5 * fair ** 0.5 < sub < fair + 6 * fair ** 0.5 + 32
                # a binomial share of a chunk stays inside a sub-slab
                assert (rng.binomial(chunk, 1.0 / world, 20000) <= sub).all()
            for total in (0, 1, sub, sub + 1, chunks * sub, chunks * sub + 5, 3 * chunks * sub + 1):
                got = [P.plan_classic_count(total, c, chunks, sub) for c in range(chunks)]
                assert sum(got) == total and all(v == sub for v in got[:-1] if total >= chunks * sub)
                assert all(got[c] <= sub for c in range(chunks - 1)) and (got[-1] > sub) == (total > chunks * sub)
                over = max(0, total - chunks * sub)
                assert P.plan_owner_sub_slabs(total, chunk, chunks, sub) == chunks + -(-over // sub)
    cr, ch, sc = C.c_uint64(), C.c_uint32(), C.c_uint64()
    P.plan_geometry(20_000, 4, 16, C.byref(cr), C.byref(ch), C.byref(sc))       # tests force tiny slabs: one chunk, one sub-slab
    assert (ch.value, sc.value) == (1, 16) and cr.value >= 20_000


def _gloo_rank(rank, world, port, reads, cap, result_dir):
    """One rank of the exchange, round after round, with the product's geometry and a set for a table."""
    import hashlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = _plan()
        W = world
        as_p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
        seen = set()
        rounds, _, n_per, _ = reads.shape
        out = np.zeros((rounds, n_per), np.uint8)
        for k in range(rounds):
            mine = reads[k, rank]
            n = n_per - (5 * rank if k == 1 else 0)                     # a shorter batch on some ranks
            keys = [mine[i].tobytes() for i in range(n)]
            owner = np.array([int.from_bytes(hashlib.blake2b(s, digest_size=8).digest(), "little") >> 40 for s in keys], dtype=np.uint64) % np.uint64(W)
            out_counts = np.bincount(owner.astype(np.int64), minlength=W).astype(np.uint64)
            slots = P.plan_spill_slot(as_p(out_counts), W, W, cap)
            grouped = np.zeros((slots, L), np.uint8); origin = np.full(slots, -1, np.int64)
            seen_n = [0] * W
            for i in range(n):                                          # fqd_partition_slabs: slab first, then the spill region
                d = int(owner[i]); local = seen_n[d]; seen_n[d] += 1
                at = P.plan_slab_slot(d, cap) + local if local < cap else P.plan_spill_slot(as_p(out_counts), W, d, cap) + (local - cap)
                grouped[at] = mine[i]; origin[at] = i
            # forward: ONE all-to-all of fixed-size slabs, the counts beside it — nothing here depends on a count
            recv = torch.zeros(W * cap * L, dtype=torch.uint8)
            dist.all_to_all_single(recv, torch.from_numpy(grouped[: W * cap].reshape(-1).copy()))
            in_counts_t = torch.zeros(W, dtype=torch.int64)
            dist.all_to_all_single(in_counts_t, torch.from_numpy(out_counts.astype(np.int64)))
            in_counts = in_counts_t.numpy().astype(np.uint64)
            slot = recv.numpy().reshape(W * cap, L)
            # spills: point to point, exactly sized, both ends know the count
            spill = np.zeros((P.plan_spill_slot(as_p(in_counts), W, W, cap), L), np.uint8)
            spill[: W * cap] = slot
            reqs, landing = [], []
            for peer in range(W):
                if out_counts[peer] > cap:
                    a = P.plan_spill_slot(as_p(out_counts), W, peer, cap)
                    t = torch.from_numpy(grouped[a: a + int(out_counts[peer]) - cap].reshape(-1).copy())
                    if peer == rank: landing.append((peer, t))
                    else: reqs.append(dist.isend(t, peer))
                if in_counts[peer] > cap and peer != rank:
                    t = torch.zeros((int(in_counts[peer]) - cap) * L, dtype=torch.uint8)
                    reqs.append(dist.irecv(t, peer)); landing.append((peer, t))
            for q in reqs: q.wait()
            for peer, t in landing:
                a = P.plan_spill_slot(as_p(in_counts), W, peer, cap)
                spill[a: a + t.numel() // L] = t.numpy().reshape(-1, L)
            # the owner inserts in (source rank, position) order
            n_ins = P.plan_owner_records(as_p(in_counts), W, cap)
            keep_recv = np.full(n_ins + cap, 7, np.uint8)
            for s in range(W):
                c = int(in_counts[s]); at = P.plan_owner_offset(as_p(in_counts), W, s, cap)
                rows = [spill[P.plan_slab_slot(s, cap) + i] for i in range(min(c, cap))]
                rows += [spill[P.plan_spill_slot(as_p(in_counts), W, s, cap) + i] for i in range(max(0, c - cap))]
                for i, row in enumerate(rows):
                    key = row.tobytes()
                    keep_recv[at + i] = 0 if key in seen else 1
                    seen.add(key)
            # flags back: cap bytes a pair whatever the counts, plus the spill's
            send_back = np.concatenate([keep_recv[P.plan_owner_offset(as_p(in_counts), W, s, cap):][:cap] for s in range(W)])
            back = torch.zeros(W * cap, dtype=torch.uint8)
            dist.all_to_all_single(back, torch.from_numpy(send_back.copy()))
            keep_back = np.full(slots + cap, 3, np.uint8)
            keep_back[: W * cap] = back.numpy()
            reqs, landing = [], []
            for peer in range(W):
                if in_counts[peer] > cap:
                    a = P.plan_owner_offset(as_p(in_counts), W, peer, cap) + cap
                    t = torch.from_numpy(keep_recv[a: a + int(in_counts[peer]) - cap].copy())
                    if peer == rank: landing.append((peer, t))
                    else: reqs.append(dist.isend(t, peer))
                if out_counts[peer] > cap and peer != rank:
                    t = torch.zeros(int(out_counts[peer]) - cap, dtype=torch.uint8)
                    reqs.append(dist.irecv(t, peer)); landing.append((peer, t))
            for q in reqs: q.wait()
            for peer, t in landing:
                a = P.plan_spill_slot(as_p(out_counts), W, peer, cap)
                keep_back[a: a + t.numel()] = t.numpy()
            ok = origin >= 0
            out[k, origin[ok]] = keep_back[: slots][ok]
        np.save(os.path.join(result_dir, f"keep{rank}.npy"), out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,cap", [(2, 400), (3, 300), (2, 24), (3, 7)])
def test_slab_exchange_under_gloo(oracle, tmp_path, world, cap):
    """cap 400 / 300: slabs hold a fair share (about 300 / 200 of 600 reads) and nothing spills; 24 / 7: every pair spills."""
    import torch.multiprocessing as mp
    _build()
    n_per, rounds = 600, 3
    rng = np.random.default_rng(world * 100 + cap)
    pool = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(world * n_per * rounds // 3 + 1, L))
    reads = pool[rng.integers(0, len(pool), size=(rounds, world, n_per))]
    mp.spawn(_gloo_rank, args=(world, _free_port(), reads, cap, str(tmp_path)), nprocs=world, join=True)
    got = np.stack([np.load(tmp_path / f"keep{r}.npy") for r in range(world)], axis=1)        # [round][rank][i]
    mask = np.ones((rounds, world, n_per), bool)
    for r in range(world):
        if r: mask[1, r, n_per - 5 * r:] = False
    flat = reads[mask].reshape(-1, L)
    n = len(flat)
    exp = oracle.dedup_single(np.concatenate([flat.reshape(-1), np.zeros(8, np.uint8)]),
                              np.arange(n, dtype=np.uint64) * np.uint64(L), np.full(n, L, np.uint32))
    assert np.array_equal(got[mask], exp)
    assert 0 < int((exp == 0).sum()) < n
