"""fqd_bgzf_inflate / fqd_count_lines / fqd_scan_records on the GPU: BGZF members of every deflate block type
must inflate to what zlib makes of them (tests/test_inflate_core.py holds the same decoder to that on the CPU),
damaged members must be counted, and the record arrays must be those of the host scanner's rule."""
import random
import struct
import time

import numpy as np
import pytest

from bgzf_cases import fasta_text, fastq_text
from inflate_cases import bgzf, cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from fastq_dupaway_amd import Engine
    with Engine(segments=1, device=0) as e:
        yield e


def walk(raw: bytes):
    """comp_off, comp_len, out_off, out_len, crc of every member with data (what the host driver reads off the headers)."""
    rows, at, out = [], 0, 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 14] == b"BC"
        total = struct.unpack_from("<H", raw, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", raw, at + total - 8)
        if isize:
            rows.append((at + 18, total - 26, out, isize, crc))
            out += isize
        at += total
    a = np.array(rows, dtype=np.uint64).reshape(-1, 5)
    return a[:, 0].copy(), a[:, 1].astype(np.uint32), a[:, 2].copy(), a[:, 3].astype(np.uint32), a[:, 4].astype(np.uint32), out


def device_inflate(eng, raw: bytes):
    import torch
    dev = torch.device("cuda", 0)
    co, cl, oo, ol, crc, total = walk(raw)
    comp = torch.frombuffer(bytearray(raw + b"\0" * 16), dtype=torch.uint8).to(dev)
    t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
    text = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    args = [t(co), t(cl), t(oo), t(ol), t(crc)]
    torch.cuda.synchronize()
    bad = eng.bgzf_inflate(comp, *args, len(co), text)
    return bad, text[:total].cpu().numpy().tobytes()


@pytest.mark.parametrize("name,data,raw", list(cases()), ids=[c[0] for c in cases()])
def test_every_block_type_inflates_as_zlib_does(eng, name, data, raw):
    bad, got = device_inflate(eng, raw)
    assert bad == 0 and got == data


def test_whatever_zlib_writes_comes_back(eng):
    """The property test of tests/test_inflate_core.py on the kernels: the same random members, other seeds too."""
    from inflate_cases import random_cases
    for seed in (77, 5):
        for trial, (data, raw) in enumerate(random_cases(seed, 40)):
            bad, got = device_inflate(eng, raw)
            assert bad == 0 and got == data, (seed, trial)


def test_damaged_members_are_counted(eng):
    data = fastq_text(1200, 3)
    raw = bgzf(data)
    rnd = random.Random(5)
    caught = 0
    for trial in range(40):
        dmg = bytearray(raw)
        at = rnd.randrange(18, len(dmg) - 40)
        if dmg[at - 18:at - 14] == b"\x1f\x8b\x08\x04" or at % 1 != 0:
            continue
        dmg[at] ^= 1 << rnd.randrange(8)
        try:
            bad, got = device_inflate(eng, bytes(dmg))
        except AssertionError:                               # the flip hit a header field the walk itself rejects
            continue
        assert bad > 0 or got == data                        # a flip in a trailer's ISIZE/CRC or the deflate bits: counted
        caught += bad > 0
    assert caught >= 20


def test_deflate_then_inflate_600_mb_on_the_device(eng):
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(6)
    n, L = 1_900_000, 150
    rec = torch.empty((n, 18 + L + 3 + L + 1), dtype=torch.uint8, device=dev)
    x = torch.arange(n, device=dev, dtype=torch.int64)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    for p in range(9):
        rec[:, 10 - p] = (48 + x % 10).to(torch.uint8); x = x // 10
    rec[:, 11:18] = torch.tensor(list(b" 1:N:0\n"), dtype=torch.uint8, device=dev)
    rec[:, 18:18 + L] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, L), device=dev, generator=g)]
    rec[:, 18 + L] = 10; rec[:, 19 + L] = ord("+"); rec[:, 20 + L] = 10
    rec[:, 21 + L:21 + 2 * L] = torch.tensor(list(b"FFFFFFFF:,#"), dtype=torch.uint8, device=dev)[torch.randint(0, 11, (n, L), device=dev, generator=g)]
    rec[:, 21 + 2 * L] = 10
    src = rec.reshape(-1)
    nbytes = src.numel()
    dst = torch.empty(eng.bgzf_bound(nbytes), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    size = eng.bgzf_deflate(src, nbytes, dst, 4)
    raw = dst[:size].cpu().numpy().tobytes()
    co, cl, oo, ol, crc, total = walk(raw)
    assert total == nbytes
    t = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32)).to(dev)
    args = [t(co), t(cl), t(oo), t(ol), t(crc)]
    text = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.bgzf_inflate(dst, *args, len(co), text)           # warm-up
    text.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = eng.bgzf_inflate(dst, *args, len(co), text)
    dt = time.perf_counter() - t0
    print(f"\n[inflate] {size / 1e6:.0f} MB -> {nbytes / 1e6:.0f} MB, {len(co)} members in {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s of text")
    assert bad == 0 and torch.equal(text[:total], src)
    # and the records of that text
    t0 = time.perf_counter()
    lines = eng.count_lines(text, total)
    assert lines == 4 * n
    start = torch.empty(n, dtype=torch.int64, device=dev); seq_off = torch.empty(n, dtype=torch.int64, device=dev)
    id_len = torch.empty(n, dtype=torch.int32, device=dev); seq_len = torch.empty(n, dtype=torch.int32, device=dev); sz = torch.empty(n, dtype=torch.int32, device=dev)
    ok = eng.scan_records(text, total, 4, n, start, seq_off, id_len, seq_len, sz)
    print(f"[scan] {n} records in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    R = rec.shape[1]
    want = torch.arange(n, device=dev, dtype=torch.int64) * R
    assert ok and torch.equal(start, want) and torch.equal(seq_off, want + 18)
    assert bool((id_len == 18).all()) and bool((seq_len == L).all()) and bool((sz == R).all())


def numpy_records(data: bytes, k: int):
    nl = np.flatnonzero(np.frombuffer(data, dtype=np.uint8) == 10)
    n = len(nl) // k
    ends = nl[: n * k].reshape(n, k)
    start = np.concatenate([[0], ends[:-1, -1] + 1]) if n else np.zeros(0, np.int64)
    return start, ends[:, 0] + 1, ends[:, 0] - start + 1, ends[:, 1] - ends[:, 0] - 1, ends[:, -1] - start + 1


@pytest.mark.parametrize("kind", ["fastq", "fastq_ragged", "fasta", "one_record"])
def test_records_are_the_host_scanners(eng, kind):
    import torch
    dev = torch.device("cuda", 0)
    data, k = {"fastq": (fastq_text(5000, 41), 4), "fastq_ragged": (fastq_text(5000, 42, style="ragged"), 4),
               "fasta": (fasta_text(7000, 43), 2), "one_record": (b"@a\nAC\n+\nII\n", 4)}[kind]
    text = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    lines = eng.count_lines(text, len(data))
    assert lines == data.count(b"\n")
    n = lines // k
    out = [torch.empty(n, dtype=d, device=dev) for d in (torch.int64, torch.int64, torch.int32, torch.int32, torch.int32)]
    assert eng.scan_records(text, len(data), k, n, *out)
    for got, want in zip(out, numpy_records(data, k)):
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want.astype(np.int64))


@pytest.mark.parametrize("damage", ["no_final_newline", "bad_lead", "length_mismatch", "extra_line", "blank_tail"])
def test_text_that_is_not_whole_records_is_reported(eng, damage):
    import torch
    dev = torch.device("cuda", 0)
    recs = [b"@r%d\nACGTAC\n+\nIIIIII\n" % i for i in range(3000)]
    if damage == "no_final_newline":
        data = b"".join(recs)[:-1]
    elif damage == "bad_lead":
        recs[1234] = b"r1234\nACGTAC\n+\nIIIIII\n"; data = b"".join(recs)
    elif damage == "length_mismatch":
        recs[2999] = b"@r2999\nACGTAC\n+\nIIIII\n"; data = b"".join(recs)
    elif damage == "extra_line":
        data = b"".join(recs) + b"@x\n"
    else:
        data = b"".join(recs) + b"\n\n\n\n"
    text = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    lines = eng.count_lines(text, len(data))
    n = lines // 4
    out = [torch.empty(max(n, 1), dtype=d, device=dev) for d in (torch.int64, torch.int64, torch.int32, torch.int32, torch.int32)]
    assert not eng.scan_records(text, len(data), 4, n, *out)
