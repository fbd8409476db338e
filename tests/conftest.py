"""pytest plumbing: markers, paths, and loaders for the checker libraries.

`-m "not gpu"` covers the oracle (against golden vectors and, when built, the
reference objects in oracle/_ref), host logic and the C-ABI's symbol table.
`-m gpu` tests are the parity tests proper and call through the C-ABI.
Nothing here reads /root/reference at run time.
"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (built on demand with g++)."""
    import oracle_binding
    return oracle_binding.load_oracle()


@pytest.fixture(scope="session")
def reflib():
    """ctypes handle on oracle/_ref/libfqd_ref.so, or skip when it was never built."""
    import oracle_binding
    lib = oracle_binding.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libfqd_ref.so not built (needs /root/reference)")
    return lib
