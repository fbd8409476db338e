#!/usr/bin/env python3
"""One targeted run for the ">1 GiB all_to_all_single message arrives corrupted" observation of
round 1: the same one-rank exchange (RCCL copies the rank's message to itself)

  plain   : ordinary torch.empty tensors on both sides
  alias   : the receive side is raw device memory seen through __cuda_array_interface__
            (__cuda_array_interface__), pointing into an ordinary torch allocation
  store   : the receive side is the tail of an engine's key store (fqd_reserve_keys), which is
            what ShardedDedup does in production

at message sizes below and above 1 GiB (and above 2 and 4 GiB), every word checked.
Writes one JSON line per case to stdout.
"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch
import torch.distributed as dist


def pattern(n_words, dev, salt):
    x = torch.arange(n_words, dtype=torch.int64, device=dev)
    return x * 0x9E3779B97F4A7C15 % (1 << 62) + salt      # wraps; any fixed bijection-ish pattern does


def first_bad(a, b):
    ne = (a != b)
    n_bad = int(ne.sum().item())
    if n_bad == 0:
        return 0, None, None
    idx = torch.nonzero(ne)[:, 0]
    return n_bad, int(idx[0].item()), int(idx[-1].item())


def main():
    from fastq_dupaway_amd import Engine
    class _DeviceWords:
        def __init__(self, ptr, n_words):
            self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 3, "strides": None}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sizes_mib = [int(s) for s in os.environ.get("A2A_SIZES_MIB", "512,1000,1024,1025,1100,2048,2049,4100").split(",")]
    for mib in sizes_mib:
        n = mib * (1 << 20) // 8
        src = pattern(n, dev, mib)
        for mode in ("plain", "alias", "store", "plain_chunked"):
            eng = None
            keepalive = None
            if mode in ("plain", "plain_chunked"):
                dst = torch.zeros(n, dtype=torch.int64, device=dev)
            elif mode == "alias":
                keepalive = torch.zeros(n + 1024, dtype=torch.int64, device=dev)
                dst = torch.as_tensor(_DeviceWords(keepalive.data_ptr() + 512 * 8, n), device=dev)
            else:
                eng = Engine(segments=1, device=0)
                n_keys = n // 8                               # 150-bp keys: 8 words each
                ptr = eng.reserve_keys(n_keys, 150, 0)
                dst = torch.as_tensor(_DeviceWords(ptr, n_keys * 8), device=dev)
                dst.zero_()
            m = dst.numel()
            torch.cuda.synchronize()
            if mode == "plain_chunked":                        # what a caller-side cap does: 256 MiB slices
                step = (256 << 20) // 8
                for lo in range(0, m, step):
                    hi = min(m, lo + step)
                    dist.all_to_all_single(dst[lo:hi], src[lo:hi], output_split_sizes=[hi - lo], input_split_sizes=[hi - lo])
            else:
                dist.all_to_all_single(dst[:m], src[:m], output_split_sizes=[m], input_split_sizes=[m])
            torch.cuda.synchronize()
            n_bad, lo_bad, hi_bad = first_bad(dst[:m], src[:m])
            print(json.dumps({"mib": mib, "mode": mode, "words": m, "bad_words": n_bad,
                              "first_bad_word": lo_bad, "last_bad_word": hi_bad,
                              "first_bad_byte_offset": None if lo_bad is None else lo_bad * 8}), flush=True)
            del dst, keepalive
            if eng is not None:
                eng.close()
            torch.cuda.empty_cache()
        del src
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
