"""GPU parity of the `--unordered` join primitives (fqd_extract_tags, fqd_sort_tags, fqd_join_tags,
fqd_gather_seqs: csrc/fqd_join.hip) against the order and equality the reference defines
(FastqViewWithId::cmp / read_new, fastqview.cpp:168-204 — pinned for the oracle in
tests/test_oracle.py) and against the oracle's merge-join (hash_dup_remover.hpp:279-340)."""
import numpy as np
import pytest
import torch

from fastq_dupaway_amd import Engine

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


def tag_arrays(tags):
    lens = np.array([len(t) for t in tags], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) if len(tags) else np.zeros(0, np.uint64)
    data = np.frombuffer(b"".join(tags) + b"\0" * 16, dtype=np.uint8).copy()
    return data, offs, lens


def to_dev(*arrs):
    return [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
            for a in arrs]


def make_tags(rng, n, style):
    if style == "illumina":
        return [b"M0%d:7:FCX:1:%d:%d:%d" % (k % 3, 1100 + int(rng.integers(0, 30)), int(rng.integers(0, 99999)), k) for k in range(n)]
    if style == "sra":
        return [b"%d" % (k + 1) for k in range(n)]
    if style == "newline":
        return [b"read%07d\n" % k for k in range(n)]           # tags that run through the newline (no space in the ID)
    if style == "long":                                       # 200-400 bytes, shared 180-byte prefix, differences far beyond the LDS census range
        prefix = b"instrument:run:flowcell:" * 8
        return [prefix[:180] + bytes(rng.choice(list(b"ACGT0123456789:"), size=int(rng.integers(20, 220))).astype(np.uint8)) + b"/%d" % k for k in range(n)]
    if style == "wide":                                       # random bytes: every position needs 8 bits, keys of several words
        return [bytes(rng.integers(1, 256, size=int(rng.integers(0, 30))).astype(np.uint8)) + b"|%d" % k for k in range(n)]
    # mixed lengths, shared prefixes, prefix-of-each-other cases
    base = [b"a", b"ab", b"abc", b"abcdefgh", b"abcdefghi", b"abcdefgh" * 3, b"abcdefgh" * 3 + b"x", b"b", b"", b"zz" * 20]
    return base + [bytes(rng.choice(list(b"abcXYZ019:"), size=int(rng.integers(0, 40))).astype(np.uint8)) + b"#%d" % k for k in range(n - len(base))]


def run_join(e, a, b):
    """Full device join of two tag lists; returns host arrays."""
    da, oa, la = tag_arrays(a); db, ob, lb = tag_arrays(b)
    A = to_dev(da, oa, la); B = to_dev(db, ob, lb)
    i32 = dict(dtype=torch.int32, device="cuda")
    pa, ma = torch.empty(max(1, len(a)), **i32), torch.empty(max(1, len(a)), **i32)
    pb, mb = torch.empty(max(1, len(b)), **i32), torch.empty(max(1, len(b)), **i32)
    cap = max(1, min(len(a), len(b)))
    qa, qb = torch.empty(cap, **i32), torch.empty(cap, **i32)
    n_pairs = e.join_tags((*A, len(a)), (*B, len(b)), pa, pb, ma, mb, qa, qb)
    h = lambda t, n: t.cpu().numpy().view(np.uint32)[:n]
    return dict(perm_a=h(pa, len(a)), perm_b=h(pb, len(b)), match_a=h(ma, len(a)), match_b=h(mb, len(b)),
                pair_a=h(qa, n_pairs), pair_b=h(qb, n_pairs), n_pairs=n_pairs)


@pytest.mark.parametrize("style", ["illumina", "sra", "newline", "mixed", "wide", "long"])
def test_sort_tags_matches_reference_order(oracle, style):
    rng = np.random.default_rng(5)
    n = 20000 if style != "long" else 4000
    tags = make_tags(rng, n, style)
    order = rng.permutation(n)
    tags = [tags[i] for i in order]
    data, offs, lens = tag_arrays(tags)
    d, o, l = to_dev(data, offs, lens)
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    with Engine(segments=2) as e:
        e.sort_tags(d, o, l, n, perm)
    got = perm.cpu().numpy().view(np.uint32)
    exp = sorted(range(n), key=lambda i: tags[i])            # bytes order == strncmp-then-shorter-first for NUL-free tags
    assert [tags[i] for i in got] == [tags[i] for i in exp]
    assert sorted(got.tolist()) == list(range(n))
    # spot-check the comparator against the oracle's restatement of FastqViewWithId::cmp
    for k in range(0, n - 1, 997):
        assert oracle.compare_tags(tags[got[k]], tags[got[k + 1]]) <= 0


def test_sort_is_stable_for_equal_tags():
    tags = [b"x", b"a", b"x", b"a", b"m", b"x"]
    data, offs, lens = tag_arrays(tags)
    d, o, l = to_dev(data, offs, lens)
    perm = torch.empty(len(tags), dtype=torch.int32, device="cuda")
    with Engine(segments=2) as e:
        e.sort_tags(d, o, l, len(tags), perm)
    assert perm.cpu().tolist() == [1, 3, 4, 0, 2, 5]


@pytest.mark.parametrize("style", ["illumina", "wide", "long"])
def test_join_gives_the_oracles_full_join(oracle, style):
    rng = np.random.default_rng(9)
    n = 30000 if style != "long" else 5000
    ids = make_tags(rng, n, style)
    a = [ids[i] for i in rng.permutation(n) if rng.random() < 0.9]
    b = [ids[i] for i in rng.permutation(n) if rng.random() < 0.8]
    with Engine(segments=2) as e:
        j = run_join(e, a, b)
    pos_b = {b[r]: k for k, r in enumerate(j["perm_b"])}
    pos_a = {a[r]: k for k, r in enumerate(j["perm_a"])}
    assert all(j["match_a"][k] == pos_b.get(a[j["perm_a"][k]], NONE) for k in range(len(a)))
    assert all(j["match_b"][k] == pos_a.get(b[j["perm_b"][k]], NONE) for k in range(len(b)))
    i1, i2, un = oracle.join_tags(*tag_arrays(a), *tag_arrays(b), tail_rule=False)
    assert list(zip(j["pair_a"].tolist(), j["pair_b"].tolist())) == list(zip(i1.tolist(), i2.tolist()))
    assert un == len(a) + len(b) - 2 * j["n_pairs"]


def test_join_pairs_repeated_ids_rank_by_rank(oracle):
    """IDs repeated within a file (the reference's std::sort leaves their order unspecified; the
    oracle's stable merge-join pairs the k-th with the k-th): runs longer than a scan tile included."""
    rng = np.random.default_rng(17)
    universe = [b"id%03d" % k for k in range(40)]
    a = [universe[i] for i in rng.integers(0, 40, 3000)] + [b"hot"] * 5000 + [b"x", b"x", b"x"]
    b = [universe[i] for i in rng.integers(0, 30, 2500)] + [b"hot"] * 7000 + [b"x"]
    order_a, order_b = rng.permutation(len(a)), rng.permutation(len(b))
    a = [a[i] for i in order_a]; b = [b[i] for i in order_b]
    with Engine(segments=2) as e:
        j = run_join(e, a, b)
    i1, i2, un = oracle.join_tags(*tag_arrays(a), *tag_arrays(b), tail_rule=False)
    assert list(zip(j["pair_a"].tolist(), j["pair_b"].tolist())) == list(zip(i1.tolist(), i2.tolist()))
    assert un == len(a) + len(b) - 2 * j["n_pairs"]
    # one-sided inputs
    with Engine(segments=2) as e:
        assert run_join(e, a, [])["n_pairs"] == 0
        assert run_join(e, [], b)["n_pairs"] == 0
        assert run_join(e, [b"same"] * 300, [b"same"] * 200)["n_pairs"] == 200
        assert run_join(e, [b""] * 3, [b"", b"a"])["n_pairs"] == 1


def ref_tag(line: bytes):
    """FastqViewWithId::read_new (fastqview.cpp:190-204) restated: (offset, length) of the tag in the ID line."""
    dot = line.find(b".")
    start = dot + 1 if dot >= 0 else 1
    sp = line.find(b" ", start)
    return start, (sp if sp >= 0 else len(line)) - start


def test_extract_tags_follows_the_reference_rule():
    lines = [b"@r1\n", b"@SRR1.10 x y\n", b"@a:b:c 1:N:0\n", b"@\n", b"@x.\n", b"@x. y\n", b"@no_space_but.dot\n",
             b"@two words.here z\n", b">fa.7 len=5\n", b"@M01:7:FC:1:1101:1000:2000 2:N:0:ACGT\n", b"@.\n", b"@. \n", b"@a b.c\n"]
    rng = np.random.default_rng(3)
    lines += [bytes(rng.choice(list(b"ab. :1"), size=int(rng.integers(0, 25))).astype(np.uint8)).replace(b"\n", b"") for _ in range(500)]
    lines = [(l if l.startswith((b"@", b">")) else b"@" + l) for l in lines]
    lines = [l if l.endswith(b"\n") else l + b"\n" for l in lines]
    text = b"".join(l + b"ACGT\n+\nIIII\n" for l in lines)
    starts, at = [], 0
    for l in lines:
        starts.append(at); at += len(l) + 12
    d_text = torch.from_numpy(np.frombuffer(text + b"\0" * 16, dtype=np.uint8).copy()).cuda()
    d_start = torch.tensor(starts, dtype=torch.int64, device="cuda")
    d_len = torch.tensor([len(l) for l in lines], dtype=torch.int32, device="cuda")
    off = torch.empty(len(lines), dtype=torch.int64, device="cuda"); ln = torch.empty(len(lines), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with Engine(segments=2) as e:
        e.extract_tags(d_text, d_start, d_len, len(lines), off, ln)
        e.sync()
    for k, l in enumerate(lines):
        o, n = ref_tag(l)
        assert (int(off[k]) - starts[k], int(ln[k])) == (o, n), l


def digits(x, width):
    """ASCII decimal, fixed width, of an int64 tensor -> uint8 [n, width]."""
    cols = []
    for _ in range(width):
        cols.append((x % 10 + 48).to(torch.uint8)); x = x // 10
    return torch.stack(cols[::-1], dim=1)


@pytest.mark.parametrize("style", ["one_word", "two_words"])
def test_join_at_50m_tags_per_side(style):
    """configs[4] scale: ~50 M tags in each file, shuffled, ~10 % orphans on each side; expected order,
    matches and pairs in closed form (torch sort of the numeric keys).  one_word: "r%09d\\n" (36 key
    bits); two_words: a constant flow-cell prefix + 9 + 9 varying digits (72 key bits, two 64-bit
    words, so the second word's pass re-gathers the tags through the permutation).
    (Comparisons go through same(): a bare `assert torch.equal(...)` that fails makes pytest format two
    50 M-element tensors, which looks like a hang.)"""
    def same(x, y):
        return bool(torch.equal(x, y))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(77)
    n_ids = 56_000_000
    ids = torch.arange(n_ids, dtype=torch.int64, device=dev)
    if style == "one_word":
        key = ids
        body = torch.cat([torch.full((n_ids, 1), ord("r"), dtype=torch.uint8, device=dev), digits(ids, 9),
                          torch.full((n_ids, 1), 10, dtype=torch.uint8, device=dev)], dim=1)
    else:
        f1 = (ids * 2654435761 + 12345) % 1_000_000_000      # 9 digits: f1 * 1e9 + ids stays inside int64
        key = f1 * 1_000_000_000 + ids                       # fixed-width decimal fields: byte order = numeric order
        prefix = torch.tensor(list(b"M01234:55:000000000-ABCDE:1:"), dtype=torch.uint8, device=dev).repeat(n_ids, 1)
        body = torch.cat([prefix, digits(f1, 9), torch.full((n_ids, 1), ord("#"), dtype=torch.uint8, device=dev), digits(ids, 9)], dim=1)
        del prefix, f1
    width = body.shape[1]

    def side(drop_mod, drop_rem):
        keep = ids % drop_mod != drop_rem                    # ~10 % orphans, different ones per side
        mine = ids[keep]
        mine = mine[torch.randperm(mine.numel(), device=dev, generator=g)]
        return mine
    ida, idb = side(10, 3), side(10, 7)
    na, nb = ida.numel(), idb.numel()
    assert na > 50_000_000 and nb > 50_000_000
    tags_a = body[ida].contiguous().view(-1); tags_b = body[idb].contiguous().view(-1)
    del body
    off_a = torch.arange(na, dtype=torch.int64, device=dev) * width; off_b = torch.arange(nb, dtype=torch.int64, device=dev) * width
    len_a = torch.full((na,), width, dtype=torch.int32, device=dev); len_b = torch.full((nb,), width, dtype=torch.int32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    pa, ma, pb, mb = torch.empty(na, **i32), torch.empty(na, **i32), torch.empty(nb, **i32), torch.empty(nb, **i32)
    qa, qb = torch.empty(min(na, nb), **i32), torch.empty(min(na, nb), **i32)
    torch.cuda.synchronize()              # the engine runs on its own stream: torch's kernels that built the tags must be done
    with Engine(segments=2) as e:
        n_pairs = e.join_tags((tags_a, off_a, len_a, na), (tags_b, off_b, len_b, nb), pa, pb, ma, mb, qa, qb)
    # expected: each side ordered by key; a record has a partner iff its id survives on the other side
    exp_pa = torch.argsort(key[ida]); exp_pb = torch.argsort(key[idb])
    assert same(pa.long(), exp_pa) and same(pb.long(), exp_pb)
    common = (ids % 10 != 3) & (ids % 10 != 7)
    assert n_pairs == int(common.sum())
    sorted_ids_a = ida[exp_pa]; sorted_ids_b = idb[exp_pb]
    has_a = common[sorted_ids_a]; has_b = common[sorted_ids_b]
    assert same(ma.long() != -1, has_a)                                             # int32 view: -1 = none
    # the k-th pair is the k-th common id in key order on both sides
    assert same(ida[qa[:n_pairs].long()], sorted_ids_a[has_a])
    assert same(idb[qb[:n_pairs].long()], sorted_ids_b[has_b])
    # partners point at each other
    k = torch.nonzero(has_a)[:, 0]
    assert same(mb[ma[k].long()].long(), k)


def test_device_halves_of_the_streaming_run():
    """fqd_copy_spans, fqd_count_tags_le, fqd_output_offsets (the bounded-memory --unordered run) against numpy."""
    rng = np.random.default_rng(21)
    n = 50000
    src = rng.integers(0, 256, size=4_000_000, dtype=np.uint8)
    lens = rng.integers(0, 70, size=n).astype(np.uint32); lens[:5] = [0, 1, 7, 8, 9]
    lens[5:21] = [15, 16, 17, 31, 32, 33, 127, 128, 129, 143, 144, 145, 300, 322, 1000, 1001]   # eight lanes, sixteen bytes each: every edge of that
    lens[1000:3000] = rng.integers(100, 400, size=2000)                                           # (and records of the usual size)
    src_off = rng.integers(0, len(src) - 1100, size=n).astype(np.uint64)
    dst_off = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    total = int(lens.sum())
    d_src, d_so, d_ln, d_do = to_dev(src, src_off, lens, dst_off)
    d_dst = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with Engine(segments=2) as e:
        e.copy_spans(d_src, d_so, d_ln, n, d_dst, d_do)
        e.sync()
        got = d_dst.cpu().numpy()
        exp = np.concatenate([src[int(o):int(o) + int(l)] for o, l in zip(src_off, lens)])
        assert np.array_equal(got[:total], exp) and not got[total:].any()

        # tag counts: how many tags of t are <= one tag of `other`, in the reference's order
        tags = make_tags(rng, 3000, "mixed") + [b"dup", b"dup", b""]
        other = [b"abc", b"", b"zz", b"dup", b"abcdefgh", b"M", tags[100], tags[2000]]
        T = to_dev(*tag_arrays(tags)); O = to_dev(*tag_arrays(other))
        torch.cuda.synchronize()
        for k, probe in enumerate(other):
            assert e.count_tags_le((*T, len(tags)), (*O, len(other)), k) == sum(t <= probe for t in tags), probe

        # output offsets: kept pairs' record sizes, running sum in pair order, scattered to the records
        n_rec, n_pairs = 70000, 60000
        sizes = rng.integers(20, 400, size=n_rec).astype(np.uint32)
        idx = rng.permutation(n_rec)[:n_pairs].astype(np.uint32)
        keep = (rng.random(n_pairs) < 0.8).astype(np.uint8)
        d_sz, d_idx = to_dev(sizes, idx); d_keep = torch.from_numpy(keep).cuda()
        d_dest = torch.full((n_rec,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        tot = e.output_offsets(d_keep, d_idx, n_pairs, d_sz, d_dest)
        kept_sizes = np.where(keep == 1, sizes[idx], 0).astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(kept_sizes)[:-1]])
        exp_dest = np.full(n_rec, -1, dtype=np.int64)
        exp_dest[idx[keep == 1]] = starts[keep == 1]
        assert tot == int(kept_sizes.sum())
        assert np.array_equal(d_dest.cpu().numpy(), exp_dest)

        # the same per pair (text resident in HBM): source offset, length (0 when dropped), destination offset;
        # then one window of the output assembled by fqd_copy_spans
        rec_start = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.uint64)]).astype(np.uint64)
        text = rng.integers(32, 127, size=int(sizes.sum()) + 16, dtype=np.uint8)
        d_start, d_text = to_dev(rec_start, text)
        i64 = dict(dtype=torch.int64, device="cuda")
        d_src, d_dst = torch.empty(n_pairs, **i64), torch.empty(n_pairs, **i64)
        d_len = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        tot2 = e.output_plan(d_keep, d_idx, n_pairs, d_start, d_sz, d_src, d_len, d_dst)
        assert tot2 == tot
        assert np.array_equal(d_src.cpu().numpy().view(np.uint64), rec_start[idx])
        assert np.array_equal(d_len.cpu().numpy().view(np.uint32), kept_sizes.astype(np.uint32))
        assert np.array_equal(d_dst.cpu().numpy(), starts)
        a, b = 10000, 30000                                   # pairs [a, b) -> bytes [starts[a], starts[b])
        lo, hi = int(starts[a]), int(starts[b])
        d_win = torch.zeros(hi - lo + 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        e.copy_spans(d_text, d_src[a:], d_len[a:], b - a, d_win.data_ptr() - lo, d_dst[a:])
        e.sync()
        exp_win = np.concatenate([text[int(rec_start[idx[k]]):int(rec_start[idx[k]]) + int(sizes[idx[k]])] for k in range(a, b) if keep[k]])
        assert np.array_equal(d_win.cpu().numpy()[: hi - lo], exp_win)

        # no index list: pair k is record k (the ordered run's plan)
        m = min(n_pairs, n_rec)
        tot3 = e.output_plan(d_keep, None, m, d_start, d_sz, d_src, d_len, d_dst)
        kept3 = np.where(keep[:m] == 1, sizes[:m], 0).astype(np.int64)
        assert tot3 == int(kept3.sum())
        assert np.array_equal(d_src.cpu().numpy().view(np.uint64)[:m], rec_start[:m])
        assert np.array_equal(d_len.cpu().numpy().view(np.uint32)[:m], kept3.astype(np.uint32))
        assert np.array_equal(d_dst.cpu().numpy()[:m], np.concatenate([[0], np.cumsum(kept3)[:-1]]))


def test_join_fuzz_small_cases_against_the_oracle(oracle):
    """Many small joins — tags drawn from tiny alphabets so that repeats, prefixes of each other, empty tags
    and one-sided runs are the rule — against the oracle's stable merge-join (full join)."""
    rng = np.random.default_rng(123)
    with Engine(segments=2) as e:
        for case in range(250):
            alphabet = [b"a", b"b", b"ab", b"", b"a\n", b"b:1", b"b:10", b"b:2", b"zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz", b"\xff", b"\x01"]
            k = int(rng.integers(1, len(alphabet) + 1))
            pick = [alphabet[i] for i in rng.permutation(len(alphabet))[:k]]
            a = [pick[int(i)] for i in rng.integers(0, k, size=int(rng.integers(0, 40)))]
            b = [pick[int(i)] for i in rng.integers(0, k, size=int(rng.integers(0, 40)))]
            j = run_join(e, a, b)
            if a and b:
                i1, i2, un = oracle.join_tags(*tag_arrays(a), *tag_arrays(b), tail_rule=False)
                exp = list(zip(i1.tolist(), i2.tolist()))
            else:
                exp = []
            got = list(zip(j["pair_a"].tolist(), j["pair_b"].tolist()))
            assert got == exp, (case, a, b)
            assert sorted(j["perm_a"].tolist()) == list(range(len(a))) and sorted(j["perm_b"].tolist()) == list(range(len(b)))
