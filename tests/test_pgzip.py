"""The parallel reader of ORDINARY gzip files (fastq-dupaway_amd/host/pgzip.hpp: block starts guessed per chunk, symbols
with an unknown window, a chain of block boundaries that only the true start can begin) against zlib: every kind of
deflate block, several members, header fields, what follows the last member, damage of every kind — on its own
(tests/native/pgzip_check.cpp, built with the sanitizers) and behind InputFile (tests/native/io_check.cpp)."""
import gzip
import io
import os
import subprocess
import zlib
from pathlib import Path

import numpy as np
import pytest

from bgzf_cases import fastq_text

HERE = Path(__file__).resolve().parent
HOST = HERE.parent / "fastq-dupaway_amd" / "host"
SRC = HERE / "native" / "pgzip_check.cpp"
EXE = HERE / "native" / "pgzip_check"
IO_SRC = HERE / "native" / "io_check.cpp"
IO_EXE = HERE / "native" / "io_check"


def build():
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, (HOST / "pgzip.hpp").stat().st_mtime):
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-o", str(EXE), str(SRC), "-lz", "-lpthread"], check=True, capture_output=True)
    deps = [IO_SRC, HOST / "file_io.cpp", HOST / "file_io.hpp", HOST / "pgzip.hpp"]
    if not IO_EXE.exists() or IO_EXE.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(IO_EXE), str(IO_SRC), str(HOST / "file_io.cpp"), "-lz", "-lpthread"],
                       check=True, capture_output=True)


@pytest.fixture(scope="module")
def text():
    build()
    base = fastq_text(30000, 5) * 2                           # ~21 MB: some seven chunks of compressed bytes
    arr = np.frombuffer(base, dtype=np.uint8).copy()
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(arr), size=len(arr) // 40)
    arr[idx] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=len(idx))
    return arr.tobytes()


def run(raw: bytes, tmp_path, threads=8, read_size=1 << 20):
    src, out = tmp_path / "in.gz", tmp_path / "out.bin"
    src.write_bytes(raw)
    r = subprocess.run([str(EXE), str(src), str(out), str(threads), str(read_size)], capture_output=True, text=True, timeout=600)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[:2000]
    return r.returncode, r.stdout.strip(), out.read_bytes()


@pytest.mark.parametrize("level", [1, 6, 9])
def test_levels(text, tmp_path, level):
    rc, said, got = run(gzip.compress(text, level), tmp_path)
    assert rc == 0 and got == text


@pytest.mark.parametrize("threads,read_size", [(2, 1 << 20), (8, 12345), (5, 64 << 20)])
def test_thread_counts_and_read_sizes(text, tmp_path, threads, read_size):
    rc, said, got = run(gzip.compress(text, 6), tmp_path, threads, read_size)
    assert rc == 0 and got == text


@pytest.mark.parametrize("kind", ["fixed", "huffman_only", "rle", "stored", "sync_flushes", "full_flushes"])
def test_every_kind_of_block(text, tmp_path, kind):
    data = text[:12_000_000]
    if kind in ("sync_flushes", "full_flushes"):
        how = zlib.Z_SYNC_FLUSH if kind == "sync_flushes" else zlib.Z_FULL_FLUSH
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        raw = b"".join(c.compress(data[i:i + 777_777]) + c.flush(how) for i in range(0, len(data), 777_777)) + c.flush()
    else:
        level = 0 if kind == "stored" else 6
        strategy = {"fixed": zlib.Z_FIXED, "huffman_only": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE, "stored": zlib.Z_DEFAULT_STRATEGY}[kind]
        c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        raw = c.compress(data) + c.flush()
    rc, said, got = run(raw, tmp_path)
    assert rc == 0 and got == data


def test_members_header_fields_and_what_follows(text, tmp_path):
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    first = c.compress(text[:9_000_000]) + c.flush()
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads.fq", mode="wb", fileobj=buf, compresslevel=4) as g:
        g.write(text[9_000_000:])
    both = first + buf.getvalue()
    for raw in (both, both + b"\0" * 100, both + b"no gzip member, this", first + gzip.compress(b"") + buf.getvalue()):
        rc, said, got = run(raw, tmp_path)
        assert rc == 0 and got == text
    for tiny in (b"", b"x", b"hello\n" * 5):
        rc, said, got = run(gzip.compress(tiny), tmp_path)
        assert rc == 0 and got == tiny


def test_damage_is_reported_after_the_text_before_it(text, tmp_path):
    raw = gzip.compress(text, 6)
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x10
    rc, said, got = run(bytes(bad), tmp_path)
    # (what a flipped bit turns the rest of its block into is text as good as any to a decoder, zlib included: the damage
    #  shows where a code is impossible, or in the CRC at the end; what was decoded before the flip is the file's text)
    assert rc == 3 and said == "corrupt" and len(got) > len(text) // 4 and got[: len(text) // 4] == text[: len(text) // 4]
    rc, said, got = run(raw[: len(raw) * 2 // 3], tmp_path)
    assert rc == 3 and len(got) > len(text) // 2 and text.startswith(got)
    bad = bytearray(raw); bad[-6] ^= 1                                                 # the CRC in the trailer
    rc, said, got = run(bytes(bad), tmp_path)
    assert rc == 3
    bad = bytearray(raw); bad[-2] ^= 1                                                 # ISIZE
    rc, said, got = run(bytes(bad), tmp_path)
    assert rc == 3


def test_random_damage_never_crashes(text, tmp_path):
    import random
    rnd = random.Random(9)
    raw = gzip.compress(text[:8_000_000], 6)
    for trial in range(25):
        bad = bytearray(raw)
        for _ in range(rnd.randrange(1, 4)):
            at = rnd.randrange(10, len(bad))
            bad[at] = rnd.randrange(256)
        rc, said, got = run(bytes(bad), tmp_path, threads=rnd.choice([2, 4, 8]))
        assert rc in (0, 3)
        if rc == 0:
            assert got == text[:8_000_000]


def test_behind_the_input_file(text, tmp_path):
    """InputFile picks this reader for ordinary .gz files of some size read with several threads: same bytes as zlib's."""
    src = tmp_path / "in.fq.gz"
    src.write_bytes(gzip.compress(text, 6))
    outs = {}
    for pg in ("1", "0"):
        env = dict(os.environ, FQD_PGZIP=pg, FQD_PGZIP_MIN_MB="1")
        r = subprocess.run([str(IO_EXE), "r", str(src), str(3 << 20), "8"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        outs[pg] = r.stdout.split()
    assert outs["1"] == outs["0"] and int(outs["1"][0]) == len(text)


def test_text_that_packs_a_thousandfold(tmp_path):
    """300 MB of runs and repeats in 0.4 MB of gzip, then ordinary text: pieces end early instead of growing with the ratio."""
    build()
    rng = np.random.default_rng(3)
    line = bytes(rng.choice(list(b"ACGT"), 150).tolist()) + b"\n"
    data = b"\0" * 150_000_000 + line * 1_000_000 + fastq_text(20000, 9)
    rc, said, got = run(gzip.compress(data, 6), tmp_path, threads=4)
    assert rc == 0 and got == data
    # and a stream of stored blocks only, longer than a piece may grow: it has no block a worker would look for
    noise = np.random.default_rng(4).integers(0, 256, size=60_000_000, dtype=np.uint8).tobytes()
    rc, said, got = run(gzip.compress(noise, 0), tmp_path, threads=4)
    assert rc == 0 and got == noise
