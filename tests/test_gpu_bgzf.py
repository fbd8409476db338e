"""fqd_bgzf_deflate on the GPU: the kernels must produce, byte for byte, what the same coder logic
produces when tests/native/bgzf_core_check.cpp runs it thread by thread on the CPU — and that is checked
(tests/test_bgzf_core.py, and again here) to be BGZF which gzip inflates back to the input."""
import gzip
import time

import numpy as np
import pytest

from bgzf_cases import cases, fastq_text
from test_bgzf_core import EOF_MARK, harness_bgzf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from fastq_dupaway_amd import Engine
    with Engine(segments=1, device=0) as e:
        yield e


def device_bgzf(eng, data: bytes, k: int) -> bytes:
    import torch
    dev = torch.device("cuda", 0)
    src = torch.frombuffer(bytearray(data) if data else bytearray(1), dtype=torch.uint8).to(dev)
    dst = torch.empty(max(1, eng.bgzf_bound(len(data))), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    size = eng.bgzf_deflate(src, len(data), dst, k)
    return dst[:size].cpu().numpy().tobytes()


@pytest.mark.parametrize("name,data,k", list(cases()), ids=[c[0] for c in cases()])
def test_kernels_write_what_the_cpu_run_of_the_same_logic_writes(eng, tmp_path, name, data, k):
    got = device_bgzf(eng, data, k)
    want = harness_bgzf(data, k, tmp_path)
    assert want.endswith(EOF_MARK)
    assert got == want[:-len(EOF_MARK)]
    assert gzip.decompress(got + EOF_MARK) == data


def test_unaligned_source_and_reuse_of_the_engine(eng, tmp_path):
    import torch
    data = fastq_text(1500, 21)
    dev = torch.device("cuda", 0)
    buf = torch.frombuffer(bytearray(b"xyz" + data), dtype=torch.uint8).to(dev)
    dst = torch.empty(eng.bgzf_bound(len(data)), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):                                    # scratch reused; slots must be cleared again
        size = eng.bgzf_deflate(buf[3:], len(data), dst, 4)
        got = dst[:size].cpu().numpy().tobytes()
        assert got == harness_bgzf(data, 4, tmp_path)[:-len(EOF_MARK)]


def test_bad_arguments_are_refused(eng):
    import torch
    dev = torch.device("cuda", 0)
    src = torch.zeros(1000, dtype=torch.uint8, device=dev)
    small = torch.zeros(100, dtype=torch.uint8, device=dev)
    with pytest.raises(Exception, match="dst_capacity"):
        eng.bgzf_deflate(src, 1000, small, 4)
    with pytest.raises(Exception, match="bad arguments"):
        eng.bgzf_deflate(src, 1000, torch.zeros(2000, dtype=torch.uint8, device=dev), 0)


def test_600_mb_of_fastq_inflates_to_the_input(eng):
    """A window-sized buffer: ~1.9 M records of 2x150 bp FASTQ text built on the device."""
    import torch
    dev = torch.device("cuda", 0)
    n, L = 1_900_000, 150
    g = torch.Generator(device=dev); g.manual_seed(5)
    rec = torch.empty((n, 18 + L + 3 + L + 1), dtype=torch.uint8, device=dev)
    ids = torch.arange(n, device=dev, dtype=torch.int64)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    x = ids.clone()
    for p in range(9):
        rec[:, 10 - p] = (48 + x % 10).to(torch.uint8); x //= 10
    rec[:, 11:18] = torch.tensor(list(b" 1:N:0\n"), dtype=torch.uint8, device=dev)
    rec[:, 18:18 + L] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, L), device=dev, generator=g)]
    rec[:, 18 + L] = 10; rec[:, 19 + L] = ord("+"); rec[:, 20 + L] = 10
    rec[:, 21 + L:21 + 2 * L] = torch.tensor(list(b"FFFFFFFF:,#"), dtype=torch.uint8, device=dev)[torch.randint(0, 11, (n, L), device=dev, generator=g)]
    rec[:, 21 + 2 * L] = 10
    src = rec.reshape(-1)
    nbytes = src.numel()
    dst = torch.empty(eng.bgzf_bound(nbytes), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.bgzf_deflate(src, nbytes, dst, 4)                 # warm-up (scratch allocation)
    t0 = time.perf_counter()
    size = eng.bgzf_deflate(src, nbytes, dst, 4)
    dt = time.perf_counter() - t0
    print(f"\n[bgzf] {nbytes / 1e6:.0f} MB -> {size / 1e6:.0f} MB ({nbytes / size:.2f}x) in {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.1f} GB/s")
    got = dst[:size].cpu().numpy().tobytes()
    want = src.cpu().numpy().tobytes()
    assert gzip.decompress(got + EOF_MARK) == want
    assert size < 0.36 * nbytes
