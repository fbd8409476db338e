"""GPU parity of the `--unordered` join primitives (fqd_sort_tags, fqd_match_sorted_tags)
against the order and equality the reference defines (FastqViewWithId::cmp,
fastqview.cpp:168-178 — pinned for the oracle in tests/test_oracle.py) and against the
oracle's merge-join."""
import numpy as np
import pytest
import torch

from fastq_dupaway_amd import Engine

pytestmark = pytest.mark.gpu


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
    # mixed lengths, shared prefixes, prefix-of-each-other cases
    base = [b"a", b"ab", b"abc", b"abcdefgh", b"abcdefghi", b"abcdefgh" * 3, b"abcdefgh" * 3 + b"x", b"b", b"", b"zz" * 20]
    return base + [bytes(rng.choice(list(b"abcXYZ019:"), size=int(rng.integers(0, 40))).astype(np.uint8)) + b"#%d" % k for k in range(n - len(base))]


@pytest.mark.parametrize("style", ["illumina", "sra", "newline", "mixed"])
def test_sort_tags_matches_reference_order(oracle, style):
    rng = np.random.default_rng(5)
    n = 20000
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


@pytest.mark.parametrize("tail_rule", [False, True])
def test_match_gives_the_oracles_join(oracle, tail_rule):
    rng = np.random.default_rng(9)
    n = 30000
    ids = make_tags(rng, n, "illumina")
    a = [ids[i] for i in rng.permutation(n) if rng.random() < 0.9]
    b = [ids[i] for i in rng.permutation(n) if rng.random() < 0.8]
    da, oa, la = tag_arrays(a); db, ob, lb = tag_arrays(b)
    A = to_dev(da, oa, la); B = to_dev(db, ob, lb)
    pa = torch.empty(len(a), dtype=torch.int32, device="cuda"); pb = torch.empty(len(b), dtype=torch.int32, device="cuda")
    match = torch.empty(len(a), dtype=torch.int32, device="cuda")
    with Engine(segments=2) as e:
        e.sort_tags(*A, len(a), pa); e.sort_tags(*B, len(b), pb)
        e.match_sorted_tags((*A, len(a)), pa, (*B, len(b)), pb, match)
        e.sync()
    pa_h = pa.cpu().numpy().view(np.uint32); pb_h = pb.cpu().numpy().view(np.uint32); m = match.cpu().numpy().view(np.uint32)
    pos_b = {b[r]: k for k, r in enumerate(pb_h)}
    for k in range(len(a)):
        exp = pos_b.get(a[pa_h[k]], 0xFFFFFFFF)
        assert m[k] == exp
    # full join built from (perm, match) == the oracle's merge-join without the tail rule
    if not tail_rule:
        i1, i2, un = oracle.join_tags(da, oa, la, db, ob, lb, tail_rule=False)
        got = [(int(pa_h[k]), int(pb_h[m[k]])) for k in range(len(a)) if m[k] != 0xFFFFFFFF]
        assert got == list(zip(i1.tolist(), i2.tolist()))
        assert un == len(a) + len(b) - 2 * len(got)
