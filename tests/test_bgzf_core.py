"""The device BGZF coder's logic on the CPU (tests/native/bgzf_core_check.cpp runs the functions of
fastq-dupaway_amd/csrc/fqd_bgzf_core.hpp thread by thread, phase by phase as the kernels do): what it
writes must be BGZF that any gzip reader inflates back to the input.  tests/test_gpu_bgzf.py then
holds the kernels to the very same bytes."""
import gzip
import subprocess
import zlib
from pathlib import Path

import pytest

from bgzf_cases import cases, fastq_text

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "bgzf_core_check.cpp"
EXE = HERE / "native" / "bgzf_core_check"
CORE = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_bgzf_core.hpp"
EOF_MARK = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def build_harness():
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, CORE.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", str(EXE), str(SRC)], check=True, capture_output=True)
    return EXE


def harness_bgzf(data: bytes, lines_per_record: int, tmp_path) -> bytes:
    src, out = tmp_path / "in.bin", tmp_path / "out.gz"
    src.write_bytes(data)
    r = subprocess.run([str(build_harness()), str(src), str(out), str(lines_per_record)], check=True, capture_output=True, text=True)
    members, stored, size = map(int, r.stdout.split())
    raw = out.read_bytes()
    assert len(raw) == size and members == -(-len(data) // 65280)
    return raw


def check_members(raw: bytes, data: bytes):
    """Walks the members by their BSIZE fields; each must inflate to its 65280-byte share with the right CRC."""
    at, pos = 0, 0
    while at < len(raw):
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 10:at + 16] == b"\x06\x00BC\x02\x00"
        total = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        body = zlib.decompress(raw[at + 18:at + total - 8], wbits=-15)
        assert int.from_bytes(raw[at + total - 8:at + total - 4], "little") == zlib.crc32(body)
        assert int.from_bytes(raw[at + total - 4:at + total], "little") == len(body)
        assert body == data[pos:pos + len(body)] and (len(body) == 65280 or pos + len(body) == len(data))
        pos += len(body); at += total
    assert at == len(raw) and pos == len(data)


@pytest.mark.parametrize("name,data,k", list(cases()), ids=[c[0] for c in cases()])
def test_members_inflate_to_the_input(tmp_path, name, data, k):
    raw = harness_bgzf(data, k, tmp_path)
    assert raw.endswith(EOF_MARK)
    assert gzip.decompress(raw) == data
    check_members(raw[:-len(EOF_MARK)], data)
    if name == "random_bytes":
        assert len(raw) <= len(data) + 31 * 4 + 28            # stored members: 5 + 26 bytes each
    if name == "one_symbol":
        assert len(raw) < len(data) // 20


def test_ratio_on_fastq_is_gzip_1_class(tmp_path):
    data = fastq_text(20000, 11)
    raw = harness_bgzf(data, 4, tmp_path)
    assert gzip.decompress(raw) == data
    z1 = len(zlib.compress(data, 1))
    assert len(raw) < 1.10 * z1, (len(raw), z1, len(data))
