"""The per-thread decoder of the GPU BGZF reader (fastq-dupaway_amd/csrc/fqd_inflate_core.hpp) on the CPU:
what it makes of a member must be what zlib makes of it, for every kind of deflate block; damaged members
must be reported, never crash or loop."""
import subprocess
from pathlib import Path

import pytest

from inflate_cases import bgzf, cases
from bgzf_cases import fastq_text

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "inflate_core_check.cpp"
EXE = HERE / "native" / "inflate_core_check"
CORE = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_inflate_core.hpp"


def harness():
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, CORE.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", str(EXE), str(SRC)], check=True, capture_output=True)
    return EXE


def run(raw: bytes, tmp_path):
    src, out = tmp_path / "in.gz", tmp_path / "out.bin"
    src.write_bytes(raw)
    r = subprocess.run([str(harness()), str(src), str(out)], check=True, capture_output=True, text=True, timeout=120)
    members, bad, size = map(int, r.stdout.split())
    return members, bad, out.read_bytes()


@pytest.mark.parametrize("name,data,raw", list(cases()), ids=[c[0] for c in cases()])
def test_every_block_type_inflates_as_zlib_does(tmp_path, name, data, raw):
    members, bad, got = run(raw, tmp_path)
    assert bad == 0 and got == data and members >= 2


def test_damaged_members_are_reported(tmp_path):
    data = fastq_text(1200, 3)
    raw = bytearray(bgzf(data))
    import random
    rnd = random.Random(5)
    n_bad_runs = 0
    for trial in range(60):
        dmg = bytearray(raw)
        at = rnd.randrange(18, len(dmg) - 40)
        dmg[at] ^= 1 << rnd.randrange(8)
        members, bad, got = run(bytes(dmg), tmp_path)        # must come back; whether the flip shows is up to the CRC check
        n_bad_runs += bad > 0 or got != data
    assert n_bad_runs >= 30                                  # most single-bit flips derail the decode itself
