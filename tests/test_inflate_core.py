"""The per-wave decoder of the GPU BGZF reader (fastq-dupaway_amd/csrc/fqd_inflate_wave.hpp) on the CPU, its lanes
run as a loop (64 of them as on the GPU, and 8, which cuts a member into more windows): what it makes of a member
must be what zlib makes of it, for every kind of deflate block; damaged members must be reported, never crash or
loop.  The harness is built with the address and undefined-behaviour sanitizers: a decoder that starts from guessed
code boundaries reads garbage as a matter of course and must stay inside its buffers while it does."""
import subprocess
from pathlib import Path

import pytest

from inflate_cases import bgzf, cases
from bgzf_cases import fastq_text

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "inflate_wave_check.cpp"
EXE = HERE / "native" / "inflate_wave_check"
CORE = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_inflate_wave.hpp"


def harness():
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, CORE.stat().st_mtime):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-o", str(EXE), str(SRC)], check=True, capture_output=True)
    return EXE


def run(raw: bytes, tmp_path, lanes=64):
    src, out = tmp_path / "in.gz", tmp_path / "out.bin"
    src.write_bytes(raw)
    r = subprocess.run([str(harness()), str(src), str(out), str(lanes)], check=True, capture_output=True, text=True, timeout=120)
    assert "outside its bytes" not in r.stderr
    members, bad, size = map(int, r.stdout.split())
    return members, bad, out.read_bytes()


@pytest.mark.parametrize("lanes", [64, 8])
@pytest.mark.parametrize("name,data,raw", list(cases()), ids=[c[0] for c in cases()])
def test_every_block_type_inflates_as_zlib_does(tmp_path, name, data, raw, lanes):
    members, bad, got = run(raw, tmp_path, lanes)
    assert bad == 0 and got == data and members >= 2


def test_damaged_members_are_reported(tmp_path):
    data = fastq_text(1200, 3)
    raw = bytearray(bgzf(data))
    import random
    rnd = random.Random(5)
    n_bad_runs = 0
    for trial in range(60):
        dmg = bytearray(raw)
        for _ in range(rnd.randrange(1, 4)):
            at = rnd.randrange(18, len(dmg) - 40)
            dmg[at] ^= 1 << rnd.randrange(8)
        members, bad, got = run(bytes(dmg), tmp_path, (64, 8)[trial & 1])   # must come back; whether the flip shows is up to the CRC check
        n_bad_runs += bad > 0 or got != data
    assert n_bad_runs >= 30                                  # most single-bit flips derail the decode itself


@pytest.mark.parametrize("quality", ["flat", "mixed"])
def test_what_the_device_deflater_writes_comes_back(tmp_path, quality):
    """The two device coders against each other, both on the CPU: records whose quality line is copied from the record
    before (a chain of matches a group long, which the decoder must not walk one link per trip to memory)."""
    import numpy as np
    from test_bgzf_core import harness_bgzf
    rng = np.random.default_rng(6)
    recs = []
    for i in range(1500):
        seq = bytes(rng.choice(list(b"ACGT"), 150).tolist())
        qual = b"I" * 150 if quality == "flat" else bytes(rng.choice(list(b"FFFFFFFF:,#"), 150).tolist())
        recs.append(b"@r%09d 1:N:0\n" % i + seq + b"\n+\n" + qual + b"\n")
    data = b"".join(recs)
    raw = harness_bgzf(data, 4, tmp_path)
    for lanes in (64, 8):
        members, bad, got = run(raw, tmp_path, lanes)
        assert bad == 0 and got == data


def test_whatever_zlib_writes_comes_back(tmp_path):
    """Property test: texts of every texture (runs, short periods, repeats of earlier lines, noise) under every zlib
    level, strategy and memory level, with flushes in odd places, in members of odd sizes (inflate_cases.random_cases)."""
    from inflate_cases import random_cases
    for trial, (data, raw) in enumerate(random_cases(77, 40)):
        for lanes in (64, 8):
            members, bad, got = run(raw, tmp_path, lanes)
            assert bad == 0 and got == data, (trial, lanes)
