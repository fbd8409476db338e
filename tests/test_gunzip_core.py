"""The GPU reader of ordinary gzip files (fastq-dupaway_amd/csrc/fqd_gunzip_core.hpp) on the CPU, unit by unit as the kernels
run it (tests/native/gunzip_core_check.cpp, built with the address and undefined-behaviour sanitizers): what it makes of a
member must be what zlib makes of it, for units of several sizes (a unit with no block start in it, units far smaller than a
block, one unit for the whole stream); damage must be reported or harmless, never crash or loop."""
import random
import subprocess
from pathlib import Path

import pytest

from gunzip_cases import cases, member
from bgzf_cases import fastq_text

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "gunzip_core_check.cpp"
EXE = HERE / "native" / "gunzip_core_check"
CORE = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_gunzip_core.hpp"
WAVE = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_inflate_wave.hpp"


def harness():
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, CORE.stat().st_mtime, WAVE.stat().st_mtime):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-o", str(EXE), str(SRC)], check=True, capture_output=True)
    return EXE


def run(raw: bytes, tmp_path, unit: int, ratio: int = 16, how: str = "planes"):
    src, out = tmp_path / "in.gz", tmp_path / "out.bin"
    src.write_bytes(raw)
    r = subprocess.run([str(harness()), str(src), str(out), str(unit), str(ratio), how], check=True, capture_output=True, text=True, timeout=600)
    verdict, units, size, deflate_bytes = r.stdout.split()
    return verdict, int(units), out.read_bytes(), int(deflate_bytes)


@pytest.mark.parametrize("how", ["planes", "serial"])
@pytest.mark.parametrize("unit", [1 << 20, 65536, 4096])
@pytest.mark.parametrize("name,data,raw", list(cases()), ids=[c[0] for c in cases()])
def test_ordinary_gzip_inflates_as_zlib_does(tmp_path, name, data, raw, unit, how):
    """planes: what fqd_gunzip.hip runs — the wave decoder of the BGZF reader twice per unit over two made-up windows; serial: the
    reference form of the same scheme, one plain decoder per unit writing 16-bit symbols (fqd_gunzip_core.hpp)."""
    if how == "serial" and unit == 1 << 20:
        pytest.skip("one size less for the reference form")
    ratio = 2000 if name == "long_runs" else 16
    verdict, units, got, deflate_bytes = run(raw, tmp_path, unit, ratio, how)
    assert verdict == "ok" and got == data
    # the stream ends where the trailer begins: header + deflate + 8 bytes = the member
    header = len(raw) - 8 - deflate_bytes
    assert 10 <= header <= 60
    if unit == 65536 and len(raw) > 2_000_000 and name not in ("fastq_fixed_blocks",):
        assert units >= 4                                                  # the stream really was decoded from several guessed starts


def test_a_unit_that_outgrows_its_room_says_so(tmp_path):
    data = (b"A" * 5000 + b"\n") * 400
    verdict, _, got, _ = run(member(data, 6), tmp_path, 4096, ratio=4)
    assert verdict == "full" and got == b""


def test_damage_is_reported_or_harmless(tmp_path):
    data = fastq_text(6000, 4)
    raw = bytearray(member(data, 6))
    rnd = random.Random(9)
    seen = 0
    for trial in range(40):
        dmg = bytearray(raw)
        for _ in range(rnd.randrange(1, 4)):
            at = rnd.randrange(12, len(dmg) - 10)
            dmg[at] ^= 1 << rnd.randrange(8)
        verdict, _, got, _ = run(bytes(dmg), tmp_path, (65536, 4096)[trial & 1])     # must come back; a flipped literal shows only in the CRC
        seen += verdict != "ok" or got != data
    assert seen >= 20
    # cut short
    verdict, _, got, _ = run(bytes(raw[: len(raw) // 2]), tmp_path, 65536)
    assert verdict in ("bad", "chain")
