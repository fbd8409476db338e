"""File layer of the host driver (fastq-dupaway_amd/host/file_io.cpp), no GPU involved:
".gz" outputs are BGZF (any gzip reader takes them, members carry their size), BGZF inputs are
inflated member by member on several threads, ordinary gzip inputs still go through zlib's
gzread, plain files are read with several preads — always the same bytes."""
import gzip
import os
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
SRC = HERE / "native" / "io_check.cpp"
EXE = HERE / "native" / "io_check"
HOST = ROOT / "fastq-dupaway_amd" / "host"


def fnv(data: bytes) -> int:
    h = 1469598103934665603
    for c in data:
        h = ((h ^ c) * 1099511628211) & (2 ** 64 - 1)
    return h


@pytest.fixture(scope="module")
def build():
    deps = [SRC, HOST / "file_io.cpp", HOST / "file_io.hpp"]
    if not EXE.exists() or EXE.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(EXE), str(SRC), str(HOST / "file_io.cpp"), "-lz", "-lpthread"],
                       check=True, capture_output=True)
    return EXE


def runner(codec):
    def run(*args, ok=True):
        r = subprocess.run([str(EXE), *map(str, args)], capture_output=True, text=True, env=dict(os.environ, FQD_CODEC=codec))
        assert (r.returncode == 0) == ok, r.stderr
        return r.stdout.split(), r.stderr
    return run


# BGZF members go through libdeflate when the system has it and through zlib otherwise (FQD_CODEC=zlib
# forces the latter): every test runs under both.
@pytest.fixture(params=["zlib", "auto"])
def io(build, request):
    return runner(request.param)


def test_codecs_read_each_other(build, tmp_path):
    outs = {}
    for codec in ("zlib", "auto"):
        out = tmp_path / f"{codec}.fq.gz"
        (length, h), _ = runner(codec)("w", out, 1_000_003, 7)
        outs[codec] = (out, length, h)
    assert outs["zlib"][1:] == outs["auto"][1:]
    for writer, (out, length, h) in outs.items():
        for reader in ("zlib", "auto"):
            (l2, h2), _ = runner(reader)("r", out, 300_000, 4)
            assert (l2, h2) == (length, h), (writer, reader)
        assert len(gzip.open(out, "rb").read()) == int(length)


@pytest.mark.parametrize("n", [0, 1, 65279, 65280, 65281, 3_000_001])
def test_gz_output_is_bgzf_and_reads_back(io, tmp_path, n):
    out = tmp_path / "o.fq.gz"
    (length, h), _ = io("w", out, n, 11)
    assert int(length) == n
    data = gzip.open(out, "rb").read()                       # any gzip reader
    assert len(data) == n and (n > 300_000 or fnv(data) == int(h))
    raw = out.read_bytes()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:16] == b"BC\x02\x00"       # BGZF member header
    assert raw.endswith(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    for chunk, threads in [(1000, 1), (70_000, 4), (4_000_000, 8)]:
        (l2, h2), _ = io("r", out, chunk, threads)
        assert (int(l2), h2) == (n, h)


def test_plain_gzip_and_plain_files_read_the_same(io, tmp_path):
    (length, h), _ = io("w", tmp_path / "plain.fq", 40_000_000, 3)          # big enough for the parallel pread
    data = (tmp_path / "plain.fq").read_bytes()
    assert len(data) == int(length)
    for chunk, threads in [(1 << 20, 1), (64 << 20, 4), (33 << 20, 3)]:
        (l2, h2), _ = io("r", tmp_path / "plain.fq", chunk, threads)
        assert (l2, h2) == (length, h)
    with gzip.open(tmp_path / "std.fq.gz", "wb", compresslevel=1) as f:     # one ordinary member: zlib's gzread path
        f.write(data[:250_000])
    (l3, h3), _ = io("r", tmp_path / "std.fq.gz", 1 << 16, 4)
    assert (int(l3), int(h3)) == (250_000, fnv(data[:250_000]))
    # the same member cut short: zlib hands out what it could inflate and reports the early end only with the read after —
    # which used to pass for a clean end of file (the reference's decompressor throws: file_utils.cpp:59-66)
    raw = (tmp_path / "std.fq.gz").read_bytes()
    (tmp_path / "cut_std.fq.gz").write_bytes(raw[: len(raw) * 2 // 3])
    _, err = io("r", tmp_path / "cut_std.fq.gz", 1 << 16, 4, ok=False)
    assert "corrupt or truncated" in err


def test_damaged_bgzf_is_reported(io, tmp_path):
    out = tmp_path / "o.fq.gz"
    io("w", out, 500_000, 5)
    raw = bytearray(out.read_bytes())
    (tmp_path / "cut.fq.gz").write_bytes(raw[: len(raw) // 2])
    _, err = io("r", tmp_path / "cut.fq.gz", 1 << 20, 4, ok=False)
    assert "corrupt or truncated" in err
    raw[len(raw) // 3] ^= 0x55
    (tmp_path / "flip.fq.gz").write_bytes(raw)
    _, err = io("r", tmp_path / "flip.fq.gz", 1 << 20, 4, ok=False)
    assert "corrupt or truncated" in err


@pytest.mark.parametrize("threads", [1, 4])
def test_finished_members_are_passed_through(build, tmp_path, threads):
    """write_members (what the device deflater's output goes through): bytes land after what write() still held and
    before what comes later, whether copied by one thread or through the parallel mapping (>= 8 MB per call)."""
    run = runner("auto")
    src, out = tmp_path / "members.fq.gz", tmp_path / "out.fq.gz"
    (length, h), _ = run("w", src, 60_000_000, 9)              # ~20 MB of members
    body = gzip.open(src, "rb").read()
    run("m", src, out, threads)
    assert subprocess.run(["gzip", "-t", str(out)]).returncode == 0
    assert gzip.open(out, "rb").read() == b"@before\n" + body + b"@after\n"


def test_plain_members_after_bgzf_members_are_read_on(io, tmp_path):
    """`cat blocked.gz ordinary.gz blocked.gz`: the reference's gzip_decompressor reads members of any kind one after
    the other (file_utils.hpp:58-69); the member-parallel reader hands the rest of such a file to zlib."""
    (length, h), _ = io("w", tmp_path / "a.fq.gz", 700_001, 5)
    a = (tmp_path / "a.fq.gz").read_bytes()
    text_a = gzip.decompress(a)
    middle = bytes(range(256)) * 3000
    mixed = a[:-28] + gzip.compress(middle, 6) + a            # (the first copy without its end-of-file member, the second with it)
    (tmp_path / "mixed.fq.gz").write_bytes(mixed)
    want = text_a + middle + text_a
    assert gzip.decompress(mixed) == want
    for chunk, threads in [(1000, 1), (70_000, 4), (4_000_000, 8)]:
        (l2, h2), _ = io("r", tmp_path / "mixed.fq.gz", chunk, threads)
        assert (int(l2), int(h2)) == (len(want), fnv(want))
    # damage inside the plain member is still damage
    bad = bytearray(mixed)
    bad[len(a) - 28 + 40] ^= 0x5A
    (tmp_path / "bad.fq.gz").write_bytes(bad)
    _, err = io("r", tmp_path / "bad.fq.gz", 1 << 20, 4, ok=False)
    assert "corrupt or truncated" in err
