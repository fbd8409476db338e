"""GPU parity: the HIP engine, called through the C ABI, against the CPU oracle on the
same inputs — bit-exact keep flags (integer/byte work: no tolerance).

Covers the edge cases the reference's semantics name: empty and
ragged reads, N, chunk/word edges, equal-prefix different-length reads, first
occurrence wins across batches, PE keys where only mate 2 differs, unknown bytes.
Full-size runs (BASELINE.json configs[1], configs[2]) are checked through
size-independent properties: the closed-form keep flags of the synthetic generator,
idempotence, and agreement of the two encoder kernels.
"""
import numpy as np
import pytest

import fastq_dupaway_amd as fqd
from fastq_dupaway_amd import Engine, Reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def make_pool_reads(rng, n, pool_size, len_lo, len_hi, alphabet=b"ACGTN"):
    pool = [bytes(rng.choice(list(alphabet), size=int(rng.integers(len_lo, len_hi + 1))).astype(np.uint8))
            for _ in range(pool_size)]
    return [pool[int(rng.integers(0, pool_size))] for _ in range(n)]


def ragged_arrays(reads, gap=0):
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    pos = 0
    chunks = []
    for i, r in enumerate(reads):
        offs[i] = pos
        chunks.append(r + b"\n" * gap)
        pos += len(r) + gap
    data = np.frombuffer(b"".join(chunks) + b"\0" * 8, dtype=np.uint8).copy()
    return data, offs, lens


def oracle_keep(oracle, reads):
    d, o, l = ragged_arrays(reads)
    return oracle.dedup_single(d, o, l)


# ---- single-end, ragged, host-space submits ----------------------------------------

@pytest.mark.parametrize("gap", [0, 1, 3])
def test_se_ragged_matches_oracle(oracle, gap):
    rng = np.random.default_rng(10 + gap)
    reads = make_pool_reads(rng, 20000, 3000, 0, 200)
    d, o, l = ragged_arrays(reads, gap)
    with Engine(segments=1) as e:
        keep = e.submit([Reads(d, o, l)], len(reads))
        st = e.stats()
    exp = oracle_keep(oracle, reads)
    assert np.array_equal(keep, exp)
    assert st["records"] == len(reads) and st["duplicates"] == int((exp == 0).sum())


def test_se_streaming_batches_first_occurrence_wins(oracle):
    rng = np.random.default_rng(21)
    reads = make_pool_reads(rng, 60000, 8000, 1, 160)
    exp = oracle_keep(oracle, reads)
    got = []
    with Engine(segments=1) as e:          # no capacity hint: the table must grow and rehash
        for a in range(0, len(reads), 7000):
            d, o, l = ragged_arrays(reads[a:a + 7000])
            got.append(e.submit([Reads(d, o, l)], len(o)))
        st = e.stats()
    assert np.array_equal(np.concatenate(got), exp)
    assert st["table_slots"] >= 2 * len(reads)


def test_se_chunk_and_word_edges(oracle):
    base = b"ACGTTGCAAGCTTAGGCTAACGTTAGCATCGATCGGATCCGATTACAGCTAGCTAGGATCGATCGTACGATCGATCGGCTAGCTAGCATCGATGCATGCATGCATCGATCGATCGATGCATGCTAGCTAGCATGCTAGCATCGATGCTAGCTAGCTAG"
    reads = [b"", b"", b"A", b"A", b"N", b"G", b"ACG", b"ACGA", b"AACG", b"ACGN", b"ACGN", b"ACGG"]
    for L in (16, 17, 18, 31, 32, 33, 34, 63, 64, 65, 127, 128, 129, 150, 151):
        s = base[:L]
        reads += [s, s, s[:-1] + (b"A" if s[-1:] != b"A" else b"C"), s[:L // 2] + b"N" + s[L // 2 + 1:], s + b"A"]
    d, o, l = ragged_arrays(reads)
    with Engine(segments=1) as e:
        keep = e.submit([Reads(d, o, l)], len(reads))
    assert np.array_equal(keep, oracle_keep(oracle, reads))


# ---- uniform batches: staged (LDS) encoder vs per-lane encoder vs oracle ---------------

@pytest.mark.parametrize("L,stride", [(150, 150), (150, 151), (151, 151), (100, 100), (1, 1), (33, 40), (64, 64),
                                      (250, 250), (301, 304), (17, 17)])
def test_se_uniform_both_encoders_match_oracle(oracle, L, stride):
    rng = np.random.default_rng(L * 1000 + stride)
    n = 30011
    pool = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=(n // 3, L), p=[.245, .245, .245, .245, .02])
    pick = rng.integers(0, len(pool), size=n)
    buf = np.full((n, stride), ord("\n"), dtype=np.uint8)
    buf[:, :L] = pool[pick]
    flat = np.concatenate([buf.reshape(-1), np.zeros(16, dtype=np.uint8)])
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(stride))
    lens = np.full(n, L, dtype=np.uint32)
    exp = oracle.dedup_single(flat, offs, lens)
    for no_stage in (False, True):
        with Engine(segments=1, no_stage=no_stage) as e:
            keep = e.submit([Reads(flat, uniform_len=L, uniform_stride=stride)], n)
        assert np.array_equal(keep, exp), f"no_stage={no_stage}"


def test_uniform_then_other_length_switches_to_ragged_layout(oracle):
    rng = np.random.default_rng(5)
    a = make_pool_reads(rng, 5000, 900, 50, 50)
    b = make_pool_reads(rng, 5000, 900, 40, 60)
    c = a[:2500]                                     # repeats of batch 1 must still be found
    exp = oracle_keep(oracle, a + b + c)
    with Engine(segments=1) as e:
        da = np.frombuffer(b"".join(a) + b"\0" * 16, dtype=np.uint8).copy()
        k1 = e.submit([Reads(da, uniform_len=50, uniform_stride=50)], len(a))
        d, o, l = ragged_arrays(b)
        k2 = e.submit([Reads(d, o, l)], len(b))
        dc = np.frombuffer(b"".join(c) + b"\0" * 16, dtype=np.uint8).copy()
        k3 = e.submit([Reads(dc, uniform_len=50, uniform_stride=50)], len(c))
    assert np.array_equal(np.concatenate([k1, k2, k3]), exp)


# ---- paired-end -----------------------------------------------------------------------

def test_pe_matches_oracle_ragged_and_uniform(oracle):
    rng = np.random.default_rng(77)
    n = 25000
    r1 = make_pool_reads(rng, n, 1500, 0, 120)
    r2 = make_pool_reads(rng, n, 40, 0, 120)          # few mate-2 variants: many R1-equal, R2-different pairs
    d1, o1, l1 = ragged_arrays(r1)
    d2, o2, l2 = ragged_arrays(r2, gap=2)
    exp = oracle.dedup_paired(d1, o1, l1, d2, o2, l2)
    assert 0 < int((exp == 0).sum()) < n
    with Engine(segments=2) as e:
        keep = e.submit([Reads(d1, o1, l1), Reads(d2, o2, l2)], n)
    assert np.array_equal(keep, exp)
    # uniform 2x150
    L = 150
    p1 = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(2000, L))
    p2 = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=(50, L))
    a = p1[rng.integers(0, 2000, size=n)].reshape(-1)
    b = p2[rng.integers(0, 50, size=n)].reshape(-1)
    a = np.concatenate([a, np.zeros(16, np.uint8)]); b = np.concatenate([b, np.zeros(16, np.uint8)])
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, np.uint32)
    exp = oracle.dedup_paired(a, offs, lens, b, offs, lens)
    for no_stage in (False, True):
        with Engine(segments=2, no_stage=no_stage) as e:
            keep = e.submit([Reads(a, uniform_len=L, uniform_stride=L), Reads(b, uniform_len=L, uniform_stride=L)], n)
        assert np.array_equal(keep, exp)


def test_pe_only_mate2_differs_survives():
    # reference fixture logic (test/test_fast.py paired: 0004 survives because only R2 differs)
    a = np.frombuffer(b"ACGTACGT" * 3 + b"\0" * 16, dtype=np.uint8).copy()
    b = np.frombuffer(b"TTTTAAAA" + b"TTTTAAAC" + b"TTTTAAAA" + b"\0" * 16, dtype=np.uint8).copy()
    with Engine(segments=2) as e:
        keep = e.submit([Reads(a, uniform_len=8, uniform_stride=8), Reads(b, uniform_len=8, uniform_stride=8)], 3)
    assert keep.tolist() == [1, 1, 0]


# ---- unknown bytes ------------------------------------------------------------------------

@pytest.mark.parametrize("no_stage", [False, True])
def test_unknown_base_is_reported_first_in_input_order(no_stage):
    L, n = 150, 5000
    rng = np.random.default_rng(3)
    buf = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, L))
    buf[4000, 17] = ord("x"); buf[1234, 149] = ord("\r"); buf[1234, 60] = ord("n"); buf[3000, 0] = ord("a")
    flat = np.concatenate([buf.reshape(-1), np.zeros(16, np.uint8)])
    with Engine(segments=1, no_stage=no_stage) as e:
        with pytest.raises(fqd.FqdError) as ei:
            e.submit([Reads(flat, uniform_len=L, uniform_stride=L)], n)
        assert ei.value.code == 3
        assert e.bad_base() == (1234, 0, 60, ord("n"))


def test_unknown_base_in_mate2_and_flags_before_it_are_valid(oracle):
    reads1 = [b"ACGT", b"ACGT", b"GGGG", b"ACGT", b"TTTT"]
    reads2 = [b"AAAA", b"AAAA", b"CCCC", b"AAgA", b"CCCC"]
    d1, o1, l1 = ragged_arrays(reads1); d2, o2, l2 = ragged_arrays(reads2)
    keep = np.full(5, 9, dtype=np.uint8)
    with Engine(segments=2) as e:
        with pytest.raises(fqd.FqdError):
            e.submit([Reads(d1, o1, l1), Reads(d2, o2, l2)], 5, keep=keep)
        assert e.bad_base() == (3, 1, 2, ord("g"))
    assert keep[:3].tolist() == [1, 0, 1]


# ---- the reference's own fixtures through the engine ------------------------------------------

def _fasta_records(path):
    lines = path.read_bytes().split(b"\n")
    return [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]


def test_reference_fixture_single_fast(golden_dir):
    fx = golden_dir / "reference_fixtures"
    recs = _fasta_records(fx / "inputs" / "single_fast.fa")
    d, o, l = ragged_arrays([s for _, s in recs])
    with Engine(segments=1) as e:
        keep = e.submit([Reads(d, o, l)], len(recs))
    out = b"".join(i + b"\n" + s + b"\n" for (i, s), k in zip(recs, keep) if k)
    assert out == (fx / "expected" / "single_fast.fa").read_bytes()


def test_reference_fixture_paired_fast(golden_dir):
    fx = golden_dir / "reference_fixtures"
    r1 = _fasta_records(fx / "inputs" / "paired_fast_r1.fa"); r2 = _fasta_records(fx / "inputs" / "paired_fast_r2.fa")
    d1, o1, l1 = ragged_arrays([s for _, s in r1]); d2, o2, l2 = ragged_arrays([s for _, s in r2])
    with Engine(segments=2) as e:
        keep = e.submit([Reads(d1, o1, l1), Reads(d2, o2, l2)], len(r1))
    for recs, name in ((r1, "paired_fast_r1.fa"), (r2, "paired_fast_r2.fa")):
        out = b"".join(i + b"\n" + s + b"\n" for (i, s), k in zip(recs, keep) if k)
        assert out == (fx / "expected" / name).read_bytes()


# ---- device-space submits, synthetic workload, full-size properties -------------------------------

def test_synthetic_1m_matches_oracle_and_closed_form(oracle, torch_cuda):
    torch = torch_cuda
    n, L = 1_000_000, 150
    bases = torch.empty(n * L + 16, dtype=torch.uint8, device="cuda")
    expect = torch.empty(n, dtype=torch.uint8, device="cuda")
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    with Engine(segments=1, capacity_reads=n, capacity_bases=n * L) as e:
        e.synth_reads(1234, 0, n, L, 200, 0, bases, expect)
        e.submit([Reads(bases, uniform_len=L, uniform_stride=L)], n, keep=keep)
        e.sync()
        st = e.stats()
    assert torch.equal(keep, expect)
    host = bases.cpu().numpy()
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    exp = oracle.dedup_single(host, offs, np.full(n, L, np.uint32))
    assert np.array_equal(keep.cpu().numpy(), exp)
    dups = int((exp == 0).sum())
    assert st["duplicates"] == dups and 0.19 * n < dups < 0.21 * n       # ~20 % duplicates (BASELINE configs[0])
    assert (host[: n * L] == ord("N")).sum() > 0                            # the N path is exercised


@pytest.mark.parametrize("paired", [False, True])
def test_table_grows_between_bulk_batches(torch_cuda, paired):
    """No capacity hint: the table is rebuilt (rehash) several times while batches large enough for
    the bulk path keep arriving, so slot tags change width between batches (the tag is cut to what
    fits an 8-byte partition record of the table's size) and old owners must still be found."""
    torch = torch_cuda
    S = 2 if paired else 1
    sizes = [1_200_000, 1_300_000, 2_900_000, 1_100_000, 6_000_000, 5_000_000]
    n, L = sum(sizes), 150
    bases = [torch.empty(n * L + 16, dtype=torch.uint8, device="cuda") for _ in range(S)]
    expect = torch.empty(n, dtype=torch.uint8, device="cuda")
    keep = torch.zeros(n, dtype=torch.uint8, device="cuda")
    with Engine(segments=S) as e:
        for m in range(S):
            e.synth_reads(77, 0, n, L, 300, m, bases[m], expect if m == S - 1 else None)
        at, slots = 0, []
        for k in sizes:
            e.submit([Reads(bases[m][at * L:], uniform_len=L, uniform_stride=L) for m in range(S)], k, keep=keep[at:])
            e.sync()
            slots.append(e.stats()["table_slots"])
            at += k
        st = e.stats()
    assert len(set(slots)) >= 3                                            # the table really grew on the way
    assert torch.equal(keep, expect)
    assert st["duplicates"] == int((expect == 0).sum().item())


def test_synthetic_pairs_match_oracle(oracle, torch_cuda):
    torch = torch_cuda
    n, L = 400_000, 150
    b1 = torch.empty(n * L + 16, dtype=torch.uint8, device="cuda"); b2 = torch.empty_like(b1)
    expect = torch.empty(n, dtype=torch.uint8, device="cuda"); keep = torch.empty_like(expect)
    with Engine(segments=2, capacity_reads=n) as e:
        e.synth_reads(99, 0, n, L, 200, 0, b1, None)
        e.synth_reads(99, 0, n, L, 200, 1, b2, expect)
        e.submit([Reads(b1, uniform_len=L, uniform_stride=L), Reads(b2, uniform_len=L, uniform_stride=L)], n, keep=keep)
        e.sync()
    assert torch.equal(keep, expect)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, np.uint32)
    exp = oracle.dedup_paired(b1.cpu().numpy(), offs, lens, b2.cpu().numpy(), offs, lens)
    assert np.array_equal(keep.cpu().numpy(), exp)
    # about half of the mate-1 duplicates differ in mate 2, so pair-dups ~ 10 %
    assert 0.08 * n < int((exp == 0).sum()) < 0.12 * n


def test_full_size_100m_se_properties(torch_cuda):
    """BASELINE.json configs[1]: 100 M x 150 bp, ~20 % duplicates, one MI355X."""
    torch = torch_cuda
    n, L = 100_000_000, 150
    bases = torch.empty(n * L + 16, dtype=torch.uint8, device="cuda")
    expect = torch.empty(n, dtype=torch.uint8, device="cuda")
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    with Engine(segments=1, capacity_reads=n, capacity_bases=n * L) as e:
        e.synth_reads(2026, 0, n, L, 200, 0, bases, expect)
        e.submit([Reads(bases, uniform_len=L, uniform_stride=L)], n, keep=keep)
        e.sync()
        assert torch.equal(keep, expect)                          # closed-form flags of the generator
        assert e.stats()["duplicates"] == int((expect == 0).sum().item())
        # idempotence: feeding the same reads again finds every one of them
        keep2 = torch.empty_like(keep)
        e.submit([Reads(bases, uniform_len=L, uniform_stride=L)], 10_000_000, keep=keep2[:10_000_000])
        e.sync()
        assert int(keep2[:10_000_000].sum().item()) == 0
    # streamed in 8 batches gives the same flags as one batch
    with Engine(segments=1) as e:
        keep3 = torch.empty_like(keep)
        step = n // 8
        for a in range(0, n, step):
            e.submit([Reads(bases[a * L:], uniform_len=L, uniform_stride=L)], step, keep=keep3[a:a + step])
        e.sync()
        assert torch.equal(keep3, expect)


def test_full_size_100m_pe_properties(oracle, torch_cuda):
    """BASELINE.json configs[2]: 100 M pairs 2 x 150 bp: closed-form flags of the generator over all
    100 M pairs, and the CPU oracle (hash_dup_remover.hpp:228-248 restated) on the first 2 M pairs —
    a prefix is self-contained because copies only ever point at earlier pairs."""
    torch = torch_cuda
    n, L = 100_000_000, 150
    b1 = torch.empty(n * L + 16, dtype=torch.uint8, device="cuda"); b2 = torch.empty_like(b1)
    expect = torch.empty(n, dtype=torch.uint8, device="cuda"); keep = torch.empty_like(expect)
    with Engine(segments=2, capacity_reads=n, capacity_bases=2 * n * L) as e:
        e.synth_reads(7, 0, n, L, 200, 0, b1, None)
        e.synth_reads(7, 0, n, L, 200, 1, b2, expect)
        e.submit([Reads(b1, uniform_len=L, uniform_stride=L), Reads(b2, uniform_len=L, uniform_stride=L)], n, keep=keep)
        e.sync()
        assert bool(torch.equal(keep, expect))
    m = 2_000_000
    offs = np.arange(m, dtype=np.uint64) * np.uint64(L); lens = np.full(m, L, np.uint32)
    exp = oracle.dedup_paired(b1[: m * L].cpu().numpy(), offs, lens, b2[: m * L].cpu().numpy(), offs, lens)
    assert np.array_equal(keep[:m].cpu().numpy(), exp)
    assert 0.05 * m < int((exp == 0).sum()) < 0.15 * m


# ---- the two insert paths (device atomics vs radix partition + LDS segments) agree -----------------

@pytest.mark.parametrize("paired", [False, True])
def test_bulk_and_atomic_insert_paths_agree(oracle, monkeypatch, paired):
    rng = np.random.default_rng(123)
    n, L = 300_000, 100
    S = 2 if paired else 1
    pools = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(n // (4 if m == 0 else 400), L), p=[.24, .24, .24, .24, .04]) for m in range(S)]
    mates = [np.concatenate([pools[m][rng.integers(0, len(pools[m]), n)].reshape(-1), np.zeros(16, np.uint8)]) for m in range(S)]
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, np.uint32)
    exp = (oracle.dedup_paired(mates[0], offs, lens, mates[1], offs, lens) if paired
           else oracle.dedup_single(mates[0], offs, lens))
    for bulk_min in ("0", "999999999999"):
        monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
        # one shot, and streamed in uneven batches into a growing table (fresh and non-fresh segments)
        for cuts in ([0, n], [0, 1000, 60_000, 61_000, 200_000, n]):
            got = []
            with Engine(segments=S) as e:
                for a, b in zip(cuts[:-1], cuts[1:]):
                    segs = [Reads(mates[m][a * L:], uniform_len=L, uniform_stride=L) for m in range(S)]
                    got.append(e.submit(segs, b - a))
                assert e.stats()["duplicates"] == int((exp == 0).sum())
            assert np.array_equal(np.concatenate(got), exp), (bulk_min, cuts)


# ---- forced collisions: the verify-mismatch-then-probe-on branch -------------------------------------

@pytest.mark.parametrize("bulk_min", ["0", "-1"])
@pytest.mark.parametrize("paired", [False, True])
def test_forced_tag_collisions_stay_exact(oracle, monkeypatch, bulk_min, paired):
    """With FQD_FLAG_WEAK_HASH every hash loses its tag and its low position bits, so unequal keys
    collide constantly: each insert meets occupied slots whose tag 'matches', must compare the
    full keys, find them different and keep probing (and wrap inside its segment).  Results must
    still be exact, on both insert paths, across batches and a table rehash."""
    monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    rng = np.random.default_rng(99)
    n, L = 120_000, 75
    S = 2 if paired else 1
    pools = [rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(n // 3, L)) for _ in range(S)]
    mates = [np.concatenate([pools[m][rng.integers(0, len(pools[m]), n)].reshape(-1), np.zeros(16, np.uint8)]) for m in range(S)]
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, np.uint32)
    exp = (oracle.dedup_paired(mates[0], offs, lens, mates[1], offs, lens) if paired
           else oracle.dedup_single(mates[0], offs, lens))
    got = []
    with Engine(segments=S, weak_hash=True) as e:            # no capacity hint: grows and rehashes with the weak hash too
        for a, b in ((0, 20_000), (20_000, 21_000), (21_000, n)):
            segs = [Reads(mates[m][a * L:], uniform_len=L, uniform_stride=L) for m in range(S)]
            got.append(e.submit(segs, b - a))
        assert e.stats()["duplicates"] == int((exp == 0).sum())
    assert np.array_equal(np.concatenate(got), exp)


# ---- skew: one key repeated massively, and very long reads ---------------------------------------

@pytest.mark.parametrize("bulk_min", ["0", "-1", None])
def test_massively_repeated_key(oracle, monkeypatch, torch_cuda, bulk_min):
    """Poly-G style input: one sequence makes up most of the batch.  The bulk path must hand the
    swollen bucket to the atomic protocol (heavy_bucket_insert_kernel) and stay exact."""
    torch = torch_cuda
    if bulk_min is not None:
        monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    rng = np.random.default_rng(4)
    n, L = 1_500_000, 60
    pool = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(5000, L))
    pool[0, :] = ord("G")
    pick = rng.integers(1, 5000, size=n)
    pick[rng.random(n) < 0.7] = 0                              # 70 % of the reads are the same poly-G
    data = np.concatenate([pool[pick].reshape(-1), np.zeros(16, np.uint8)])
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    exp = oracle.dedup_single(data, offs, np.full(n, L, np.uint32))
    with Engine(segments=1, capacity_reads=n) as e:
        keep = e.submit([Reads(data, uniform_len=L, uniform_stride=L)], n)
        assert e.stats()["duplicates"] == int((exp == 0).sum())
    assert np.array_equal(keep, exp)


def test_very_long_and_empty_reads(oracle):
    rng = np.random.default_rng(12)
    pool = [bytes(rng.choice(list(b"ACGTN"), size=int(L)).astype(np.uint8)) for L in (0, 1, 5000, 5001, 20000, 20000, 65, 3)]
    reads = [pool[int(rng.integers(0, len(pool)))] for _ in range(400)]
    reads[7] = reads[5][:-1] + (b"A" if reads[5][-1:] != b"A" else b"C") if len(reads[5]) else reads[7]
    d, o, l = ragged_arrays(reads)
    with Engine(segments=1) as e:
        keep = e.submit([Reads(d, o, l)], len(reads))
    assert np.array_equal(keep, oracle_keep(oracle, reads))
    # uniform long reads: too long for the LDS tile -> per-lane encoder
    L = 3000
    block = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(50, L))
    pick = rng.integers(0, 50, size=2000)
    data = np.concatenate([block[pick].reshape(-1), np.zeros(16, np.uint8)])
    with Engine(segments=1) as e:
        keep = e.submit([Reads(data, uniform_len=L, uniform_stride=L)], 2000)
    exp = oracle.dedup_single(data, np.arange(2000, dtype=np.uint64) * np.uint64(L), np.full(2000, L, np.uint32))
    assert np.array_equal(keep, exp)


@pytest.mark.parametrize("paired", [False, True])
def test_ragged_tiles_that_do_and_do_not_fit_the_staged_span(oracle, paired):
    """The ragged encoder stages a tile's span of the input through LDS when its records lie close
    together and falls back to per-lane loads otherwise: packed reads, a few records far away from
    their neighbours, shuffled offsets and reads longer than the LDS budget all give the oracle's flags."""
    rng = np.random.default_rng(77)
    S = 2 if paired else 1
    n = 9000
    pools = [make_pool_reads(rng, n, 1200 if m == 0 else 40, 0, 180) for m in range(S)]
    for m in range(S):                                       # a few very long reads (longer than a tile's LDS budget)
        for k in (100, 2000, 2001, 7777):
            pools[m][k] = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=70_000))
        pools[m][5000] = pools[m][100]
    segs, host = [], []
    for m in range(S):
        order = np.arange(n)
        if m == 0:
            order[3000:6000] = rng.permutation(order[3000:6000])       # offsets that jump around inside tiles
        chunks, offs, lens, pos = [None] * n, np.zeros(n, np.uint64), np.zeros(n, np.uint32), 0
        for slot in range(n):                                # record order[slot] is laid out at position slot
            k = int(order[slot])
            gap = 5000 if slot % 1500 == 7 else (slot % 3)   # now and then a record far from its neighbours
            pos += gap
            offs[k], lens[k] = pos, len(pools[m][k])
            chunks[slot] = b"#" * gap + pools[m][k]
            pos += len(pools[m][k])
        data = np.frombuffer(b"".join(chunks) + b"\0" * 16, dtype=np.uint8).copy()
        segs.append(Reads(data, offs, lens)); host.append((data, offs, lens))
    with Engine(segments=S) as e:
        keep = e.submit(segs, n)
    if paired:
        exp = oracle.dedup_paired(*host[0], *host[1])
    else:
        exp = oracle.dedup_single(*host[0])
    assert np.array_equal(keep, exp)
    assert 0 < int((exp == 0).sum()) < n


# ---- limits and failures around growth ---------------------------------------------------------------

def test_capacity_limit_and_allocation_failure_are_errors_not_crashes(oracle):
    """An engine holds at most 2^32-2 records (slot = tag:32 | index:32): one more is FQD_ERR_CAPACITY,
    reported before anything is touched.  A table that cannot be allocated is an error carrying the HIP
    text; the library stays usable."""
    from fastq_dupaway_amd import _lib
    rng = np.random.default_rng(1)
    reads = [bytes(rng.choice(list(b"ACGT"), size=30).astype(np.uint8)) for _ in range(1000)]
    data, offs, lens = ragged_arrays(reads)
    with Engine(segments=1) as e:
        keep = e.submit([Reads(data, offs, lens)], len(reads))
        assert np.array_equal(keep, oracle.dedup_single(data, offs, lens))
        with pytest.raises(fqd.FqdError) as err:
            e.submit([Reads(data, uniform_len=1, uniform_stride=1)], 0xFFFFFFFF - 500)       # 1000 + n > 2^32 - 2
        assert err.value.code == _lib.ERR_CAPACITY
        # the engine is still good for more records
        keep2 = e.submit([Reads(data, offs, lens)], len(reads))
        assert int(keep2.sum()) == 0
    with pytest.raises(fqd.FqdError) as err:
        Engine(segments=1, capacity_reads=1 << 40)                                          # a 16 TiB table
    assert err.value.code == _lib.ERR_HIP and "hipMalloc" in err.value.message
    with Engine(segments=1) as e:
        assert np.array_equal(e.submit([Reads(data, offs, lens)], len(reads)), oracle.dedup_single(data, offs, lens))


# ---- ordering against a caller's stream (include/fqdupaway.h, ORDERING RULE) ---------------------

def test_engine_orders_itself_after_the_callers_stream(torch_cuda):
    """Round 3's red GPU run: a buffer torch had just zero-filled on ITS stream was handed to the engine, whose own
    non-blocking stream ran first, and the fill wiped the kernels' output.  fqd_engine_wait_stream closes that: here
    the caller's stream is kept busy with a long queue of large fills of the very buffers the engine is about to
    write, with no host synchronisation anywhere in between; every round must still read back the engine's values.
    fqd_stream_wait_engine is the other direction: torch reads the flags on its stream without a host wait."""
    torch = torch_cuda
    import ctypes as C
    from fastq_dupaway_amd import _lib
    L = fqd.load_library()
    dev = torch.device("cuda", 0)
    n, LL = 4_000_000, 150
    with Engine(segments=1) as e:
        bases = torch.empty(n * LL + 16, dtype=torch.uint8, device=dev)
        expect = torch.empty(n, dtype=torch.uint8, device=dev)
        e.synth_reads(5, 0, n, LL, 200, 0, bases, expect)
        e.sync()
        ballast = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        keep = torch.empty(n, dtype=torch.uint8, device=dev)
        seg = (_lib.ReadsDesc * 2)()
        seg[0].bases, seg[0].uniform_len, seg[0].uniform_stride = bases.data_ptr(), LL, LL
        for step in range(3):
            e.reset()
            for _ in range(8):
                ballast.fill_(step)                      # ≈1 ms of work ahead of the fill below on torch's stream
            keep.fill_(7)                                # must land BEFORE the engine's flags, not after
            stream = int(torch.cuda.current_stream(dev).cuda_stream)
            assert L.fqd_engine_wait_stream(e._h, stream) == _lib.OK          # the raw ABI, not the binding's helper
            assert L.fqd_submit(e._h, seg, n, _lib.MEM_DEVICE, keep.data_ptr()) == _lib.OK
            assert L.fqd_stream_wait_engine(e._h, stream) == _lib.OK
            same = bool(torch.equal(keep, expect))       # on torch's stream, ordered after the engine by the event only
            assert same, f"step {step}: {int((keep != expect).sum())} flags differ ({int((keep == 7).sum())} still hold the fill)"
        # the binding does the first half by itself for every tensor it is handed
        e.reset()
        for _ in range(8):
            ballast.fill_(9)
        keep.fill_(7)
        e.submit([Reads(bases, uniform_len=LL, uniform_stride=LL)], n, keep)
        e.release_to()
        assert torch.equal(keep, expect)
        assert L.fqd_engine_wait_stream(None, None) == _lib.ERR_ARG and L.fqd_stream_wait_engine(None, None) == _lib.ERR_ARG


# ---- the last batch of a run (fqd_submit_final) ---------------------------------------------------

@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("bulk_min", [None, "-1"])
def test_last_batch_declared_gives_the_same_flags(monkeypatch, torch_cuda, paired, bulk_min):
    """fqd_submit_final: same flags and counts as fqd_submit bit for bit — as the only batch (set built on chip and never
    written back), as the last of several (resident set + a bulk batch), on the atomic path (nothing to skip) — then the
    engine refuses more until it is reset, and after the reset builds a correct set over the garbage the skipped
    write-back left in the table."""
    torch = torch_cuda
    if bulk_min is not None:
        monkeypatch.setenv("FQD_BULK_MIN", bulk_min)
    S = 2 if paired else 1
    sizes = [3_000_000, 1_500_000, 2_500_000]
    n, L = sum(sizes), 150
    bases = [torch.empty(n * L + 16, dtype=torch.uint8, device="cuda") for _ in range(S)]
    expect = torch.empty(n, dtype=torch.uint8, device="cuda")
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    with Engine(segments=S, capacity_reads=n) as e:
        for m in range(S):
            e.synth_reads(31, 0, n, L, 250, m, bases[m], expect if m == S - 1 else None)
        segs = lambda at: [Reads(bases[m][at * L:], uniform_len=L, uniform_stride=L) for m in range(S)]
        # (a) the whole job as one declared-last batch
        keep.fill_(9)
        e.submit(segs(0), n, keep=keep, final=True)
        e.sync()
        assert torch.equal(keep, expect)
        assert e.stats()["duplicates"] == int((expect == 0).sum().item())
        # (b) closed until reset, whatever the entry point
        for call in (lambda: e.submit(segs(0), 10, keep=keep), lambda: e.submit(segs(0), 10, keep=keep, final=True),
                     lambda: e.insert_keys(e.reserve_keys(4, L, L if paired else 0), 4, L, L if paired else 0, keep)):
            with pytest.raises(fqd.FqdError) as ei:
                call()
            assert ei.value.code == 1 and "fqd_submit_final" in str(ei.value)
        # (c) after the reset: several batches over the stale table, the last one declared
        e.reset()
        keep.fill_(9)
        at = 0
        for k, size in enumerate(sizes):
            e.submit(segs(at), size, keep=keep[at:], final=(k == len(sizes) - 1))
            at += size
        e.sync()
        assert torch.equal(keep, expect)
        assert e.stats()["duplicates"] == int((expect == 0).sum().item())
        # (d) and once more from the table (c) left behind, undeclared: the plain path must not trust it either
        e.reset()
        keep.fill_(9)
        e.submit(segs(0), n, keep=keep)
        e.submit(segs(0), 1000, keep=keep[:1000])            # every one of these is a duplicate of the resident set
        e.sync()
        assert int(keep[:1000].sum().item()) == 0 and torch.equal(keep[1000:], expect[1000:])


def test_last_batch_with_a_massively_repeated_key(oracle, torch_cuda):
    """The skew guard under fqd_submit_final: the swollen bucket's segment is cleared in HBM and filled by the atomic
    protocol although no other segment is written back."""
    rng = np.random.default_rng(5)
    n, L = 1_500_000, 60
    pool = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(5000, L))
    pool[0, :] = ord("G")
    pick = rng.integers(1, 5000, size=n)
    pick[rng.random(n) < 0.7] = 0
    data = np.concatenate([pool[pick].reshape(-1), np.zeros(16, np.uint8)])
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    exp = oracle.dedup_single(data, offs, np.full(n, L, np.uint32))
    with Engine(segments=1, capacity_reads=n) as e:
        for _ in range(2):                                    # the second time over a table nobody wrote back
            e.reset()
            keep = e.submit([Reads(data, uniform_len=L, uniform_stride=L)], n, final=True)
            assert e.stats()["duplicates"] == int((exp == 0).sum())
            assert np.array_equal(keep, exp)
