#!/usr/bin/env python3
"""Timeline of one rocprofv3 --kernel-trace --memory-copy-trace run: every launch of the kernels whose name contains
one of the given words (start, duration), and how busy the copy engines were per time bucket.
  python tools/summarize_timeline.py DIR [--match inflate,crc] [--bucket-ms 100]"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--match", default="inflate,crc")
    ap.add_argument("--bucket-ms", type=float, default=100.0)
    ap.add_argument("--longer-ms", type=float, default=5.0, help="also list every launch that took at least this long")
    ap.add_argument("--until-ms", type=float, default=1e12, help="stop the listing and the buckets here")
    a = ap.parse_args()
    words = [w for w in a.match.split(",") if w]
    kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]) for r in rows(os.path.join(a.dir, "**", "*kernel_trace.csv"))]
    copies = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", 0) or 0))
              for r in rows(os.path.join(a.dir, "**", "*memory_copy_trace.csv"))]
    if not kernels and not copies:
        print("no traces under", a.dir); return
    t0 = min([k[0] for k in kernels] + [c[0] for c in copies])
    print(f"{len(kernels)} kernel launches, {len(copies)} copies; times in ms from the first event")
    for path in glob.glob(os.path.join(a.dir, "**", "*memory_copy_trace.csv"), recursive=True)[:1]:
        print("copy trace columns:", open(path).readline().strip())
    kernels = [k for k in kernels if (k[0] - t0) / 1e6 <= a.until_ms]
    copies = [c for c in copies if (c[0] - t0) / 1e6 <= a.until_ms]
    for s, e, name in sorted(kernels):
        if any(w in name for w in words) or (e - s) / 1e6 >= a.longer_ms:
            print(f"  {(s - t0) / 1e6:9.1f}  +{(e - s) / 1e6:8.2f} ms  {name[:70]}")
    b = a.bucket_ms * 1e6
    busy = defaultdict(lambda: defaultdict(float)); moved = defaultdict(lambda: defaultdict(int))
    for s, e, d, n in copies:
        k = int((s - t0) // b)
        busy[k][d] += (e - s); moved[k][d] += n
    kbusy = defaultdict(float)
    for s, e, name in kernels:
        kbusy[int((s - t0) // b)] += (e - s)
    print(f"per {a.bucket_ms:.0f} ms bucket: kernel time (ms, summed over streams) | copies: direction busy-ms GB")
    for k in sorted(set(busy) | set(kbusy)):
        parts = "  ".join(f"{d} {busy[k][d] / 1e6:6.1f} ms {moved[k][d] / 1e9:5.2f} GB" for d in sorted(busy[k]))
        print(f"  {k * a.bucket_ms:8.0f}  kernels {kbusy[k] / 1e6:7.1f} | {parts}")


if __name__ == "__main__":
    main()
