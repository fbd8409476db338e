"""The multi-GPU path: hash-prefix sharding with one all-to-all.

CPU part: the exchange logic of fastq-dupaway_amd/sharded.py under REAL torch.distributed
(gloo, world_size 2 and 3, separate processes), with the device halves replaced by a
numpy stand-in that the TEST injects (the product has no CPU path).  Result must equal the
oracle's keep flags on the concatenated global input, over several steps.

GPU part (one MI355X): the same ShardedDedup + the real HipOps for 2 and 4 virtual ranks
run as threads of one process sharing the card, the all-to-all emulated in-process; plus
the partition kernel's stability and totals.
"""
import hashlib
import os
import socket
import threading

import numpy as np
import pytest
import torch

from fastq_dupaway_amd.sharded import ShardedDedup

L = 50


def make_global_reads(world, n_per, steps, seed=0):
    rng = np.random.default_rng(seed)
    pool = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(world * n_per * steps // 3 + 1, L))
    pick = rng.integers(0, len(pool), size=(steps, world, n_per))
    return pool[pick]                       # [step][rank][i][L]


class NumpyOps:
    """Test stand-in for the device halves: any injective key + any hash exercises the exchange."""

    def __init__(self):
        self.seen = set()

    def key_words(self, len0, len1):
        return (len0 + len1 + 7) // 8

    def encode(self, segs, n, records):
        W = self.key_words(L, 0)
        rec = records.numpy().view(np.uint64)[: n * (W + 1)].reshape(n, W + 1)
        bases = segs[0].bases
        for i in range(n):
            s = bases[i * L:(i + 1) * L].tobytes()
            rec[i, 0] = int.from_bytes(hashlib.blake2b(s, digest_size=8).digest(), "little")
            rec[i, 1:] = np.frombuffer(s.ljust(8 * W, b"\0"), dtype=np.uint64)

    def partition(self, records, n, key_words, parts, out, counts, origin):
        rw = key_words + 1
        rec = records.numpy().view(np.uint64)[: n * rw].reshape(n, rw)
        owner = ((rec[:, 0] >> np.uint64(40)) % np.uint64(parts)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        out.numpy().view(np.uint64)[: n * rw].reshape(n, rw)[:] = rec[order]
        origin.numpy()[:n] = order.astype(np.int32)
        counts.numpy()[:] = np.bincount(owner, minlength=parts)

    def insert(self, records, n, len0, len1, keep):
        rw = self.key_words(len0, len1) + 1
        rec = records.numpy().view(np.uint64)[: n * rw].reshape(n, rw)
        k = keep.numpy()
        for i in range(n):                  # arrival order = (source rank, position): first arrival wins
            key = rec[i, 1:].tobytes()
            k[i] = 0 if key in self.seen else 1
            self.seen.add(key)

    def scatter(self, flags, origin, n, keep_out):
        keep_out.numpy()[origin.numpy()[:n]] = flags.numpy()[:n]

    def sync(self):
        pass


class _Seg:
    def __init__(self, bases):
        self.bases = bases


def _gloo_worker(rank, world, port, reads, result_dir, max_message=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        steps, _, n_per, _ = reads.shape
        sd = ShardedDedup(NumpyOps(), dist, torch.device("cpu"), n_max=n_per, len0=L)
        if max_message:
            sd.MAX_MESSAGE = max_message              # force the exchange into several passes of slices
        out = np.zeros((steps, n_per), np.uint8)
        for st in range(steps):
            keep = torch.zeros(n_per, dtype=torch.uint8)
            n_here = n_per if not (st == 1 and rank == 1) else n_per - 7       # ragged step: fewer reads on one rank
            sd.dedup([_Seg(reads[st, rank].reshape(-1))], n_here, keep)
            out[st, :n_here] = keep.numpy()[:n_here]
        np.save(os.path.join(result_dir, f"keep{rank}.npy"), out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,max_message", [(2, None), (3, None), (2, 3000), (3, 1000)])
def test_sharded_exchange_under_gloo(oracle, tmp_path, world, max_message):
    """max_message: a rank-to-rank message is ~600/world records x 64 B here, so a cap of 1-3 kB cuts every
    exchange (keys out, flags back) into 3-20 passes of slices (ShardedDedup._all_to_all)."""
    import torch.multiprocessing as mp
    n_per, steps = 600, 3
    reads = make_global_reads(world, n_per, steps, seed=world)
    mp.spawn(_gloo_worker, args=(world, _free_port(), reads, str(tmp_path), max_message), nprocs=world, join=True)
    got = np.stack([np.load(tmp_path / f"keep{r}.npy") for r in range(world)], axis=1)     # [step][rank][i]
    # the ragged step dropped the last 7 reads of rank 1 in step 1: remove them from the oracle's input too
    mask = np.ones((steps, world, n_per), bool); mask[1, 1, n_per - 7:] = False
    flat = reads[mask].reshape(-1, L)
    n = len(flat)
    exp = oracle.dedup_single(np.concatenate([flat.reshape(-1), np.zeros(8, np.uint8)]),
                              np.arange(n, dtype=np.uint64) * np.uint64(L), np.full(n, L, np.uint32))
    assert np.array_equal(got[mask], exp)
    assert 0 < int((exp == 0).sum()) < n


# ---------------------------------------------------------------- GPU: real kernels, virtual ranks

class ThreadDist:
    """In-process all-to-all between threads that each play one rank (one GPU box has one card)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.local.rank

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None):
        r, w = self.local.rank, self.world
        if input_split_sizes is None:
            per = input.numel() // w
            input_split_sizes = [per] * w
        torch.cuda.synchronize()
        self.slots[r] = (input, np.concatenate([[0], np.cumsum(input_split_sizes)]))
        self.barrier.wait()
        pos = 0
        for src in range(w):
            t, offs = self.slots[src]
            chunk = t[int(offs[r]):int(offs[r + 1])]
            output[pos:pos + chunk.numel()].copy_(chunk)
            pos += chunk.numel()
        torch.cuda.synchronize()
        self.barrier.wait()

    def all_gather_into_tensor(self, output, input):
        torch.cuda.synchronize()
        parts = self._share(input.clone())
        output.copy_(torch.cat([p.reshape(-1) for p in parts]))
        torch.cuda.synchronize()

    def _share(self, obj):
        self.slots[self.local.rank] = obj
        self.barrier.wait()
        everyone = list(self.slots)
        self.barrier.wait()
        return everyone

    def all_reduce(self, t, op=None):
        torch.cuda.synchronize()
        vals = self._share(t.clone())
        stacked = torch.stack(vals)
        t.copy_(stacked.max(dim=0).values if op == torch.distributed.ReduceOp.MAX else stacked.sum(dim=0))
        torch.cuda.synchronize()

    def all_gather_object(self, out, obj):
        out[:] = self._share(obj)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("weak", [False, True])
def test_lazy_exchange_on_gpu_with_virtual_ranks(oracle, world, paired, weak):
    """Hashes first, keys only for candidates: same flags as the oracle, also when unequal keys
    share a hash (weak = the test hash with 26 useful bits, so refuted checks do happen)."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import LazyShardedDedup
    n_per, steps, LL = 20000, 3, 150
    S = 2 if paired else 1
    dev = torch.device("cuda", 0)
    bases = [[[torch.empty(n_per * LL + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
              for _ in range(world)] for _ in range(steps)]
    gen = Engine(segments=S)
    for st in range(steps):
        for r in range(world):
            for m in range(S):
                gen.synth_reads(11, (st * world + r) * n_per, n_per, LL, 300, m, bases[st][r][m], None)
    gen.sync(); gen.close()
    tdist = ThreadDist(world)
    keeps = [[torch.zeros(n_per, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(steps)]
    errors, stats = [], [None] * world

    def worker(rank):
        try:
            tdist.bind(rank)
            torch.cuda.set_device(0)
            with Engine(segments=S, weak_hash=weak) as local, Engine(segments=1) as owner:
                sd = LazyShardedDedup(local, owner, tdist, dev, n_max=n_per, len0=LL, len1=LL if paired else 0)
                sd.MAX_MESSAGE = 64 << 10                  # several request slices per round
                for st in range(steps):
                    segs = [Reads(bases[st][rank][m], uniform_len=LL, uniform_stride=LL) for m in range(S)]
                    sd.dedup(segs, n_per, keeps[st][rank])
                    local.sync()
                stats[rank] = dict(sd.stats)
        except Exception as ex:
            import traceback; traceback.print_exc()
            errors.append(ex)
            tdist.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]; [t.join() for t in threads]
    assert not errors, errors
    n = steps * world * n_per
    offs = np.arange(n, dtype=np.uint64) * np.uint64(LL); lens = np.full(n, LL, np.uint32)
    host = [np.concatenate([bases[st][r][m][: n_per * LL].cpu().numpy() for st in range(steps) for r in range(world)]
                           + [np.zeros(8, np.uint8)]) for m in range(S)]
    exp = oracle.dedup_paired(host[0], offs, lens, host[1], offs, lens) if paired else oracle.dedup_single(host[0], offs, lens)
    got = np.concatenate([keeps[st][r].cpu().numpy() for st in range(steps) for r in range(world)])
    assert np.array_equal(got, exp)
    dups = int((exp == 0).sum())
    assert 0 < dups < n
    refuted = sum(s["refuted"] for s in stats)
    assert dups <= sum(s["requests"] for s in stats) <= dups + refuted      # a refuted candidate may still be a duplicate
    assert (refuted > 0) == weak


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("with_hash", ["0", "1"])
def test_sharded_on_gpu_with_virtual_ranks(oracle, monkeypatch, world, paired, with_hash):
    monkeypatch.setenv("FQD_SHARDED_WITH_HASH", with_hash)      # keys alone on the wire (default) / [hash | key] records
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import HipOps
    n_per, steps, LL = 20000, 2, 150
    S = 2 if paired else 1
    dev = torch.device("cuda", 0)
    bases = [[[torch.empty(n_per * LL + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
              for _ in range(world)] for _ in range(steps)]
    gen = Engine(segments=S)
    for st in range(steps):
        for r in range(world):
            for m in range(S):
                gen.synth_reads(5, (st * world + r) * n_per, n_per, LL, 300, m, bases[st][r][m], None)
    gen.sync(); gen.close()
    tdist = ThreadDist(world)
    keeps = [[torch.zeros(n_per, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(steps)]
    errors = []

    def worker(rank):
        try:
            tdist.bind(rank)
            torch.cuda.set_device(0)
            with Engine(segments=S) as eng:
                sd = ShardedDedup(HipOps(eng), tdist, dev, n_max=n_per, len0=LL, len1=LL if paired else 0)
                for st in range(steps):
                    segs = [Reads(bases[st][rank][m], uniform_len=LL, uniform_stride=LL) for m in range(S)]
                    sd.dedup(segs, n_per, keeps[st][rank])
                    eng.sync()
        except Exception as ex:                       # surface in the main thread
            errors.append(ex)
            tdist.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]; [t.join() for t in threads]
    assert not errors, errors
    n = steps * world * n_per
    offs = np.arange(n, dtype=np.uint64) * np.uint64(LL); lens = np.full(n, LL, np.uint32)
    host = [np.concatenate([bases[st][r][m][: n_per * LL].cpu().numpy() for st in range(steps) for r in range(world)]
                           + [np.zeros(8, np.uint8)]) for m in range(S)]
    exp = oracle.dedup_paired(host[0], offs, lens, host[1], offs, lens) if paired else oracle.dedup_single(host[0], offs, lens)
    got = np.concatenate([keeps[st][r].cpu().numpy() for st in range(steps) for r in range(world)])
    assert np.array_equal(got, exp)
    assert 0 < int((exp == 0).sum()) < n


@pytest.mark.gpu
def test_config3_shape_8_virtual_ranks_100m_reads(oracle):
    """BASELINE configs[3] as far as one card goes: 8 virtual ranks (threads sharing the GPU, the
    all-to-all emulated in-process) with the real kernels behind HipOps, 104 M reads in all = 13 M per
    rank in 4 pipelined-order rounds.  Global input order is (round, rank, position); the generator's
    closed-form flags must come out on every rank, and the first 4 M reads in that order must equal the
    CPU oracle's flags."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import HipOps
    world, rounds, n_round, LL = 8, 4, 3_250_000, 150
    dev = torch.device("cuda", 0)
    gen = Engine(segments=1)
    bases = [[torch.empty(n_round * LL + 16, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    expect = [[torch.empty(n_round, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    for k in range(rounds):
        for r in range(world):
            gen.synth_reads(99, (k * world + r) * n_round, n_round, LL, 200, 0, bases[k][r], expect[k][r])
    gen.sync(); gen.close()
    keeps = [[torch.zeros(n_round, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    tdist = ThreadDist(world)
    errors = []

    def worker(rank):
        try:
            tdist.bind(rank)
            torch.cuda.set_device(0)
            with Engine(segments=1, capacity_reads=int(rounds * n_round * 1.1)) as eng:
                sd = ShardedDedup(HipOps(eng), tdist, dev, n_max=n_round, len0=LL)
                for k in range(rounds):
                    sd.dedup([Reads(bases[k][rank], uniform_len=LL, uniform_stride=LL)], n_round, keeps[k][rank])
                    eng.sync()
        except Exception as ex:                       # surface in the main thread
            errors.append(ex)
            tdist.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in threads]; [t.join() for t in threads]
    assert not errors, errors
    total = rounds * world * n_round
    assert total >= 100_000_000
    dups = 0
    for k in range(rounds):
        for r in range(world):
            assert bool(torch.equal(keeps[k][r], expect[k][r])), (k, r)
            dups += int((expect[k][r] == 0).sum().item())
    assert 0.15 * total < dups < 0.25 * total
    # oracle on a prefix of the global order: round 0, ranks 0 and 1 (6.5 M reads) cut at 4 M
    m = 4_000_000
    host = torch.cat([bases[0][0][: n_round * LL], bases[0][1][: n_round * LL]])[: m * LL].cpu().numpy()
    got = torch.cat([keeps[0][0], keeps[0][1]])[:m].cpu().numpy()
    exp = oracle.dedup_single(np.concatenate([host, np.zeros(8, np.uint8)]), np.arange(m, dtype=np.uint64) * np.uint64(LL), np.full(m, LL, np.uint32))
    assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_partition_is_stable_and_complete():
    from fastq_dupaway_amd import Engine, Reads
    n, LL, parts = 100_003, 150, 8
    dev = torch.device("cuda", 0)
    with Engine(segments=1) as e:
        W = e.key_words(LL); rw = W + 1
        bases = torch.empty(n * LL + 16, dtype=torch.uint8, device=dev)
        e.synth_reads(9, 0, n, LL, 200, 0, bases, None)
        rec = torch.empty(n * rw, dtype=torch.int64, device=dev); out = torch.empty_like(rec)
        counts = torch.zeros(parts, dtype=torch.int64, device=dev); origin = torch.empty(n, dtype=torch.int32, device=dev)
        e.encode_uniform([Reads(bases, uniform_len=LL, uniform_stride=LL)], n, rec)
        e.partition_records(rec, n, W, parts, out, counts, origin)
        e.sync()
    r = rec.cpu().numpy().view(np.uint64).reshape(n, rw); o = out.cpu().numpy().view(np.uint64).reshape(n, rw)
    owner = ((r[:, 0] >> np.uint64(40)) % np.uint64(parts)).astype(np.int64)
    order = np.argsort(owner, kind="stable")
    assert np.array_equal(counts.cpu().numpy(), np.bincount(owner, minlength=parts))
    assert np.array_equal(origin.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(o, r[order])
    with Engine(segments=1) as e:                            # the same partition with the hash word left out of the rows
        keys = torch.empty(n * W, dtype=torch.int64, device=dev)
        e.partition_keys(rec, n, W, parts, keys, counts, origin)
        e.sync()
    assert np.array_equal(counts.cpu().numpy(), np.bincount(owner, minlength=parts))
    assert np.array_equal(origin.cpu().numpy(), order.astype(np.int32))
    assert np.array_equal(keys.cpu().numpy().view(np.uint64).reshape(n, W), r[order][:, 1:])


@pytest.mark.gpu
@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("bulk_min", ["0", "-1", None])
@pytest.mark.parametrize("with_hash", ["0", "1"])
def test_insert_records_over_several_rounds(oracle, monkeypatch, in_place, bulk_min, with_hash):
    """The owner-side half on its own: records ([hash | key], or keys alone) arrive round after
    round (copied in, or received in place at the tail of the key store) and
    first-occurrence-wins holds across rounds."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import HipOps
    monkeypatch.setenv("FQD_SHARDED_WITH_HASH", with_hash)
    if bulk_min is not None:
        monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    n_per, rounds, LL = 1_200_000, 3, 150
    dev = torch.device("cuda", 0)
    with Engine(segments=1) as e:
        ops = HipOps(e)
        W = e.key_words(LL); rw = W + 1
        xw = ops.exchange_words(LL, 0)
        assert xw == (rw if with_hash == "1" else W)
        bases = torch.empty(rounds * n_per * LL + 16, dtype=torch.uint8, device=dev)
        expect = torch.empty(rounds * n_per, dtype=torch.uint8, device=dev)
        e.synth_reads(77, 0, rounds * n_per, LL, 250, 0, bases, expect)
        keep = torch.zeros(rounds * n_per, dtype=torch.uint8, device=dev)
        staging = torch.empty(n_per * rw, dtype=torch.int64, device=dev)
        for k in range(rounds):
            seg = [Reads(bases[k * n_per * LL:], uniform_len=LL, uniform_stride=LL)]
            e.encode_uniform(seg, n_per, staging)
            e.sync()
            wire = staging if with_hash == "1" else staging.view(n_per, rw)[:, 1:].contiguous().view(-1)
            if in_place:
                buf = ops.recv_buffer(n_per, LL, 0, dev)
                buf[: n_per * xw].copy_(wire)                # stands in for the all-to-all writing the records
                torch.cuda.synchronize()
            else:
                buf = wire
            ops.insert(buf, n_per, LL, 0, keep[k * n_per:])
            e.sync()
        assert torch.equal(keep, expect)
        assert e.stats()["duplicates"] == int((expect == 0).sum().item())


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_pipelined_rounds_under_rccl_single_rank(monkeypatch, pipeline):
    """dedup_rounds with real RCCL collectives (one rank: the all-to-all is a copy to self) and the
    real stream/event ordering between the engine's stream and the communication stream."""
    import torch.distributed as dist
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import HipOps
    monkeypatch.setenv("FQD_SHARDED_PIPELINE", pipeline)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        n_per, rounds, LL = 1_500_000, 5, 150
        n = n_per * rounds
        with Engine(segments=1, capacity_reads=n) as e:
            bases = torch.empty(n * LL + 16, dtype=torch.uint8, device=dev)
            expect = torch.empty(n, dtype=torch.uint8, device=dev)
            e.synth_reads(31, 0, n, LL, 250, 0, bases, expect)
            e.sync()
            sd = ShardedDedup(HipOps(e), dist, dev, n_max=n_per, len0=LL)
            keep = torch.zeros(n, dtype=torch.uint8, device=dev)
            for step in range(2):                            # two steps: buffers and events are reused
                e.reset(); keep.zero_()
                sd.dedup_rounds([([Reads(bases[k * n_per * LL:], uniform_len=LL, uniform_stride=LL)], n_per, keep[k * n_per:])
                                 for k in range(rounds)])
                e.sync()
                assert torch.equal(keep, expect), f"step {step}"
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
def test_lazy_exchange_under_rccl_single_rank(paired):
    """The optimistic exchange with real RCCL collectives at sizes where the owner's hash set takes
    the bulk (partitioned) insert: flags equal the generator's analytically known ones."""
    import torch.distributed as dist
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.sharded import LazyShardedDedup
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        S = 2 if paired else 1
        n_per, rounds, LL = 2_500_000, 3, 150
        n = n_per * rounds
        with Engine(segments=S) as local, Engine(segments=1) as owner:
            bases = [torch.empty(n * LL + 16, dtype=torch.uint8, device=dev) for _ in range(S)]
            expect = torch.empty(n, dtype=torch.uint8, device=dev)
            for m in range(S):
                local.synth_reads(41, 0, n, LL, 250, m, bases[m], expect if m == S - 1 else None)
            local.sync()
            sd = LazyShardedDedup(local, owner, dist, dev, n_max=n_per, len0=LL, len1=LL if paired else 0)
            keep = torch.zeros(n, dtype=torch.uint8, device=dev)
            for k in range(rounds):
                segs = [Reads(bases[m][k * n_per * LL:], uniform_len=LL, uniform_stride=LL) for m in range(S)]
                sd.dedup(segs, n_per, keep[k * n_per:])
            local.sync()
            assert torch.equal(keep, expect)
            assert sd.stats["refuted"] == 0
            assert sd.stats["requests"] == int((expect == 0).sum().item())
    finally:
        dist.destroy_process_group()
