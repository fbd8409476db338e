"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads,
exports every symbol include/fqdupaway.h declares, and refuses to run without a GPU
(no CPU fallback).  No compute calls here."""
import ctypes as C
import subprocess

import pytest

import fastq_dupaway_amd as fqd
from fastq_dupaway_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not fqd.LIB_PATH.exists():
        fqd.build_native("lib")
    return fqd.load_library()


def test_header_symbols_are_all_exported(lib):
    names = fqd.declared_symbols()
    assert {"fqd_engine_create", "fqd_submit", "fqd_engine_sync", "fqd_bad_base", "fqd_last_error",
            "fqd_encode_uniform", "fqd_partition_slabs", "fqd_insert_slabs", "fqd_shard_round", "fqd_synth_reads"} <= set(names)
    exported = subprocess.run(["nm", "-D", "--defined-only", str(fqd.LIB_PATH)], capture_output=True, text=True).stdout
    for n in names:
        assert f" T {n}\n" in exported, f"{n} declared in fqdupaway.h but not exported"
        getattr(lib, n)


def test_abi_version_and_key_words(lib):
    assert lib.fqd_abi_version() == 5
    # words(L) = ceil(L/32) + ceil(L/64): 150 bp -> 8 words = 64 B; pairs add up
    assert lib.fqd_key_words(150, 0) == 8
    assert lib.fqd_key_words(150, 150) == 16
    assert lib.fqd_key_words(0, 0) == 0
    assert lib.fqd_key_words(1, 0) == 2
    assert lib.fqd_key_words(64, 0) == 3
    assert lib.fqd_key_words(65, 0) == 5


def test_library_holds_gfx950_code_only():
    # every device code object bundled into the library targets gfx950 and nothing else
    import re
    data = fqd.LIB_PATH.read_bytes()
    assert b"__CLANG_OFFLOAD_BUNDLE__" in data
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--([a-z0-9:+-]+)", data))
    assert targets == {b"gfx950"}, targets
    assert b"nvptx" not in data and b"sm_90" not in data


def test_bad_config_is_rejected_without_touching_a_gpu(lib):
    h = C.c_void_p()
    cfg = _lib.Config(device=0, segments=3)
    assert lib.fqd_engine_create(C.byref(cfg), C.byref(h)) == _lib.ERR_ARG
    assert h.value is None


def test_no_gpu_means_loud_failure_not_fallback(lib):
    n = C.c_int(0)
    rc = lib.fqd_device_count(C.byref(n))
    if rc == _lib.OK and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fqd.FqdError) as ei:
        fqd.Engine(segments=1)
    assert ei.value.code == _lib.ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value)
