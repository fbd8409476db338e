"""The multi-threaded record scanner of the host feeder gives exactly what scanning from the
front gives (fastq-dupaway_amd/host/records.cpp): same records, same consumed bytes, same first
malformed record — for well-formed text, text cut anywhere, and text with a bad record in it."""
import os
import random
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
SRC = HERE / "native" / "scan_check.cpp"
EXE = HERE / "native" / "scan_check"
HOST = ROOT / "fastq-dupaway_amd" / "host"


@pytest.fixture(scope="module")
def scan():
    deps = [SRC, HOST / "records.cpp", HOST / "records.hpp", HOST / "file_io.cpp"]
    if not EXE.exists() or EXE.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(EXE), str(SRC),
                        str(HOST / "records.cpp"), str(HOST / "file_io.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-lpthread",
                        "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)

    def run(text: bytes, fmt="fastq", want_tag=1):
        env = dict(os.environ, FQD_SCAN_MIN_SLICE="64")
        out = subprocess.run([str(EXE), fmt, str(want_tag)], input=text, capture_output=True, check=True, env=env).stdout
        return out.decode("latin-1").splitlines()
    return run


def fastq(rnd, n, ragged=True):
    out = []
    for i in range(n):
        L = rnd.randint(1, 40) if ragged else 30
        seq = "".join(rnd.choice("ACGTN") for _ in range(L))
        ident = rnd.choice(["@r%d" % i, "@SRR1.%d x y" % i, "@a:b:%d 1:N:0" % i, "@"])
        plus = rnd.choice(["+", "+" + ident[1:], "+@looks like an id"])
        qual = "".join(rnd.choice("@+I#>") for _ in range(L))       # quality lines may start with '@' or '+'
        out.append(f"{ident}\n{seq}\n{plus}\n{qual}\n")
    return "".join(out).encode()


def fasta(rnd, n):
    return "".join(">s%d.%d z\n%s\n" % (i, i * 7, "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 50)))) for i in range(n)).encode()


def same(lines):
    assert len(lines) == 7
    assert all(l == lines[0] for l in lines), lines


def test_well_formed_and_truncated_text(scan):
    rnd = random.Random(5)
    text = fastq(rnd, 400)
    same(scan(text))
    for cut in [0, 1, 2, len(text) // 3, len(text) // 2 + 1, len(text) - 1, len(text) - 2, len(text) - 37]:
        same(scan(text[:cut]))
    same(scan(fasta(rnd, 500), "fasta"))
    same(scan(fasta(rnd, 500)[:-3], "fasta", 0))


def test_first_malformed_record_wins(scan):
    rnd = random.Random(6)
    for trial in range(30):
        recs = [fastq(rnd, 1) for _ in range(200)]
        bad = sorted(rnd.sample(range(200), rnd.randint(1, 3)))
        for b in bad:
            kind = rnd.choice(["lead", "qual", "lines"])
            if kind == "lead":
                recs[b] = b"x" + recs[b][1:]
            elif kind == "qual":
                parts = recs[b].split(b"\n")
                parts[3] = parts[3] + b"II"
                recs[b] = b"\n".join(parts)
            else:
                recs[b] = recs[b].rsplit(b"\n", 2)[0] + b"\n"        # a record one line short shifts every later record
        lines = scan(b"".join(recs))
        same(lines)
        assert lines[0].split()[3] == "1"


def test_one_record_longer_than_a_slice(scan):
    rnd = random.Random(7)
    big = b"@big\n" + b"A" * 5000 + b"\n+\n" + b"I" * 5000 + b"\n"
    same(scan(fastq(rnd, 20) + big + fastq(rnd, 20)))
    same(scan(big))
    same(scan(big[:-1]))
