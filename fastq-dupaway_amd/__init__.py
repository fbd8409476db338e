"""MI355X-native engine for fastq-dupaway's hash-based `--fast` path.

The product is native: csrc/ (HIP kernels for gfx950 + the C ABI declared in
include/fqdupaway.h) and host/ (the C++ driver and `fastq-dupaway` CLI that mirror
HashDupRemover and main.cpp of the reference).  This Python layer is plumbing for
tests and bench.py: ctypes bindings over the C ABI (engine.py) and the
wrapper over the multi-GPU shard group (shard.py; the exchange itself is csrc/fqd_shard.hip).  There is no CPU fallback: every
entry point raises if the HIP library is missing or no GPU is present.
"""
from ._lib import LIB_PATH, REPO_ROOT, FqdError, build_native, declared_symbols, load_library  # noqa: F401
from .engine import Engine, Reads  # noqa: F401

__all__ = ["Engine", "Reads", "FqdError", "load_library", "build_native", "declared_symbols", "LIB_PATH", "REPO_ROOT"]
