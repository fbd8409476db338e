"""fqd_gunzip on the GPU: ordinary gzip members (tests/gunzip_cases.py: what zlib writes at several levels and strategies,
stored and fixed blocks, flushes, header fields) must come back as zlib inflates them, with the CRC-32 and the stream length
the trailer holds; whatever is irregular must be REPORTED (ok = 0), never crash, hang or write outside the text."""
import random
import struct
import zlib

import numpy as np
import pytest

from bgzf_cases import fastq_text
from gunzip_cases import cases, member

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from fastq_dupaway_amd import Engine
    with Engine(segments=1, device=0) as e:
        yield e


def header_len(raw: bytes) -> int:
    flg, at = raw[3], 10
    if flg & 4:
        at += 2 + struct.unpack_from("<H", raw, at)[0]
    if flg & 8:
        at = raw.index(b"\0", at) + 1
    if flg & 16:
        at = raw.index(b"\0", at) + 1
    if flg & 2:
        at += 2
    return at


def device_gunzip(eng, raw: bytes, room: int, misalign: int = 0):
    import torch
    dev = torch.device("cuda", 0)
    h = header_len(raw)
    buf = torch.frombuffer(bytearray(b"\xAA" * misalign + raw[h:] + b"\0" * 32), dtype=torch.uint8).to(dev)
    text = torch.full((room + 64,), 0xEE, dtype=torch.uint8, device=dev)
    ok, n, deflate_bytes, crc = eng.gunzip(buf[misalign:], len(raw) - h, text[:room])
    eng.sync()
    assert bool((text[room:] == 0xEE).all()), "wrote beyond the room it was given"
    return ok, text[:n].cpu().numpy().tobytes(), deflate_bytes, crc, h


@pytest.mark.parametrize("name,data,raw", list(cases()), ids=[c[0] for c in cases()])
def test_ordinary_gzip_inflates_as_zlib_does(eng, monkeypatch, name, data, raw):
    if name == "long_runs":
        monkeypatch.setenv("FQD_GUNZIP_RATIO", "2000")
    for k, unit_kb in enumerate((None, "64", "8")):
        if unit_kb:
            monkeypatch.setenv("FQD_GUNZIP_UNIT_KB", unit_kb)
        ok, got, deflate_bytes, crc, h = device_gunzip(eng, raw, len(data) + 100, misalign=(0, 3, 5)[k])
        assert ok and got == data, (name, unit_kb)
        want_crc, isize = struct.unpack_from("<II", raw, h + deflate_bytes)
        assert h + deflate_bytes + 8 == len(raw)
        assert crc == want_crc == (zlib.crc32(data) & 0xFFFFFFFF) and isize == len(data) & 0xFFFFFFFF


def test_several_members_are_walked(eng, monkeypatch):
    """`cat a.gz b.gz c.gz`: where a final block ends the trailer is read and, behind it, the next member's header; every member's
    CRC-32 and ISIZE are held against its own trailer; a member that starts behind the last guessed block start is still found."""
    a, b, c = fastq_text(30000, 5), fastq_text(9000, 6), b"@t\nACGT\n+\nIIII\n"
    from gunzip_cases import header_with_fields
    raw = member(a, 6) + member(b, 1, header=header_with_fields()) + member(c, 9) + member(b"", 6) + member(b[:70000], 6)
    data = a + b + c + b[:70000]
    for unit_kb in (None, "64", "8"):
        if unit_kb:
            monkeypatch.setenv("FQD_GUNZIP_UNIT_KB", unit_kb)
        ok, got, deflate_bytes, crc, h = device_gunzip(eng, raw, len(data) + 10)
        assert ok and got == data
        assert h + deflate_bytes + 8 == len(raw) and crc == zlib.crc32(b[:70000]) & 0xFFFFFFFF
    dmg = bytearray(raw); dmg[len(member(a, 6)) + 2] ^= 1                      # the second member's method byte: no member
    ok, _, _, _, _ = device_gunzip(eng, bytes(dmg), len(data) + 10)
    assert not ok
    wrong = bytearray(raw); wrong[len(member(a, 6)) - 8] ^= 1                  # the first member's CRC in its trailer
    ok, _, _, _, _ = device_gunzip(eng, bytes(wrong), len(data) + 10)
    assert not ok
    ok, _, _, _, _ = device_gunzip(eng, raw + b"\0\0\0\0", len(data) + 10)       # bytes behind the last member that are no member
    assert not ok


def test_batches_of_units_carry_the_window_over(eng, monkeypatch):
    """A scratch far too small for all units at once: several batches, the 32 KiB window handed from one to the next."""
    data = fastq_text(60000, 7)
    raw = member(data, 6)
    monkeypatch.setenv("FQD_GUNZIP_UNIT_KB", "64")
    monkeypatch.setenv("FQD_GUNZIP_SCRATCH_MB", "4")
    ok, got, _, crc, _ = device_gunzip(eng, raw, len(data))
    assert ok and got == data and crc == zlib.crc32(data) & 0xFFFFFFFF


def test_what_does_not_fit_or_is_damaged_is_reported(eng, monkeypatch):
    data = fastq_text(20000, 4)
    raw = member(data, 6)
    ok, got, _, _, _ = device_gunzip(eng, raw, len(data) - 1)                # a byte of room too few
    assert not ok
    monkeypatch.setenv("FQD_GUNZIP_RATIO", "2")                              # units that outgrow their symbol room
    ok, _, _, _, _ = device_gunzip(eng, raw, len(data))
    assert not ok
    monkeypatch.delenv("FQD_GUNZIP_RATIO")
    rnd = random.Random(3)
    caught = 0
    for trial in range(30):
        dmg = bytearray(raw)
        for _ in range(rnd.randrange(1, 4)):
            at = rnd.randrange(12, len(dmg) - 10)
            dmg[at] ^= 1 << rnd.randrange(8)
        ok, got, deflate_bytes, crc, h = device_gunzip(eng, bytes(dmg), len(data) + 1000)
        trailer_ok = ok and h + deflate_bytes + 8 <= len(dmg) and struct.unpack_from("<II", dmg, h + deflate_bytes) == (crc, len(got) & 0xFFFFFFFF)
        if ok and trailer_ok:
            assert got == data                                               # the flip hit nothing that matters (a header field)
        else:
            caught += 1
    assert caught >= 20
    ok, _, _, _, _ = device_gunzip(eng, raw[: len(raw) // 2], len(data))       # cut short
    assert not ok


@pytest.mark.parametrize("step,unit_kb", [(50_000, "8"), (333_333, None), (3_000_000, "64")])
def test_a_file_that_is_still_arriving(eng, monkeypatch, step, unit_kb):
    """fqd_gunzip_arriving: the packed bytes reach HBM piece by piece from another thread (what lies beyond the mark is garbage
    until then) while the call runs; the text, the CRC and the end must be those of the plain call — several members, a member
    end inside a piece, units that wait for bytes that are not there yet.  ~0 in the counter ends the call with ok = 0."""
    import ctypes
    import threading
    import time
    import torch
    if unit_kb:
        monkeypatch.setenv("FQD_GUNZIP_UNIT_KB", unit_kb)
    parts = [fastq_text(9000, 21), fastq_text(50, 22), fastq_text(14000, 23)]
    data = b"".join(parts)
    raw = b"".join(member(p, 6 if k != 1 else 1) for k, p in enumerate(parts))
    h = header_len(raw)
    dev = torch.device("cuda", 0)
    real = torch.frombuffer(bytearray(raw[h:] + b"\0" * 32), dtype=torch.uint8).to(dev)
    n = len(raw) - h
    for give_up in (False, True):
        buf = torch.randint(0, 256, (n + 32,), dtype=torch.uint8, device=dev)          # garbage where nothing has arrived
        text = torch.full((len(data) + 64,), 0xEE, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        arrived = ctypes.c_uint64(0)
        side = torch.cuda.Stream(device=dev)

        def feed():
            at = 0
            while at < n:
                hi = min(n, at + step)
                if give_up and hi > n // 2:
                    arrived.value = 0xFFFFFFFFFFFFFFFF
                    return
                with torch.cuda.stream(side):
                    buf[at:hi + (32 if hi == n else 0)].copy_(real[at:hi + (32 if hi == n else 0)])
                side.synchronize()
                arrived.value = hi
                at = hi
                time.sleep(0.002)
        t = threading.Thread(target=feed)
        t.start()
        ok, nb, deflate_bytes, crc = eng.gunzip(buf, n, text[: len(data)], arrived=arrived)
        t.join()
        eng.sync()
        if give_up:
            assert not ok
            continue
        assert ok and nb == len(data) and text[:nb].cpu().numpy().tobytes() == data
        assert h + deflate_bytes + 8 == len(raw) and crc == zlib.crc32(parts[-1]) & 0xFFFFFFFF
        assert bool((text[len(data):] == 0xEE).all())


def test_a_long_stretch_without_a_guessable_start_is_left_to_the_host(eng):
    """Stored blocks only (level 0): no dynamic block to guess, the whole member would be one wave's work.  Up to 32 MiB that is
    what happens (and the text is right); beyond, the call says "not ok" at once and the caller's host reader takes the file."""
    small = fastq_text(20000, 6)
    ok, got, _, crc, _ = device_gunzip(eng, member(small, 0), len(small))
    assert ok and got == small and crc == zlib.crc32(small) & 0xFFFFFFFF
    big = small * (-(-(40 << 20) // len(small)))
    ok, got, _, _, _ = device_gunzip(eng, member(big, 0), len(big))
    assert not ok and got == b""


def test_large_member_is_fast_enough_to_matter(eng):
    """~200 MB of FASTQ text as one member: times the call (a report, not a bar) and checks CRC and length."""
    import time
    data = fastq_text(300000, 9) * 2
    raw = member(data, 6)
    t0 = time.perf_counter()
    ok, got, deflate_bytes, crc, h = device_gunzip(eng, raw, len(data))
    dt = time.perf_counter() - t0
    assert ok and len(got) == len(data) and crc == zlib.crc32(data) & 0xFFFFFFFF
    print(f"\n[gunzip] {len(data) / 1e6:.0f} MB of text from {len(raw) / 1e6:.0f} MB in {dt * 1e3:.0f} ms (upload and read-back included)")
