#!/usr/bin/env python3
"""fqd_gunzip alone: ordinary gzip members of growing size inflated on the GPU, timed and checked against zlib.
Usage: python tools/gunzip_probe.py [records ...]   (default: 10 1000 60000 600000)"""
import struct
import sys
import time
import zlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    import torch
    from fastq_dupaway_amd import Engine
    from bgzf_cases import fastq_text
    from gunzip_cases import member
    sizes = [int(x) for x in sys.argv[1:]] or [10, 1000, 60000, 600000]
    dev = torch.device("cuda", 0)
    with Engine(segments=1) as e:
        for n in sizes:
            data = fastq_text(min(n, 60000), 3) * max(1, n // 60000)
            raw = member(data, 6)
            print(f"records {n}: text {len(data)} bytes, gzip {len(raw)} bytes", flush=True)
            buf = torch.frombuffer(bytearray(raw[10:] + b"\0" * 32), dtype=torch.uint8).to(dev)
            text = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for rep in range(2):
                t0 = time.perf_counter()
                ok, nb, db, crc = e.gunzip(buf, len(raw) - 10, text[: len(data)])
                dt = time.perf_counter() - t0
                same = ok and text[:nb].cpu().numpy().tobytes() == data
                want_crc, isize = struct.unpack_from("<II", raw, 10 + db) if ok else (0, 0)
                print(f"  ok={ok} bytes={nb} same={same} crc_ok={crc == want_crc == (zlib.crc32(data) & 0xFFFFFFFF)} {dt * 1e3:.1f} ms = {len(data) / dt / 1e9:.2f} GB/s of text", flush=True)


if __name__ == "__main__":
    main()
