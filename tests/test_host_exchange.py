"""The C++ driver's multi-GPU exchange (fastq-dupaway_amd/host/multi_gpu.cpp) on the CPU: per-pair
offsets of the all-to-all and of the flags' way back, for 1..8 ranks and random (also empty) messages;
and FQD_DEVICES parsing.  The GPU side of the same path is tests/test_cli.py::test_multi_gpu_cli_*."""
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
SRC = HERE / "native" / "exchange_check.cpp"
EXE = HERE / "native" / "exchange_check"
HOST = ROOT / "fastq-dupaway_amd" / "host"


@pytest.fixture(scope="module")
def exe():
    deps = [SRC, HOST / "multi_gpu.cpp", HOST / "multi_gpu.hpp"]
    if not EXE.exists() or EXE.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(EXE), str(SRC),
                        str(HOST / "multi_gpu.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"],
                       check=True, capture_output=True)
    return str(EXE)


@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
def test_exchange_plan_round_trip(exe, ranks):
    for seed in range(5):
        r = subprocess.run([exe, str(ranks), str(seed)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
