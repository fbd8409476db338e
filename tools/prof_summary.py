#!/usr/bin/env python3
"""Prints the per-kernel table of a `rocprofv3 --kernel-trace --stats --output-format csv` run:
  python tools/prof_summary.py <dir given to rocprofv3 -d> [max rows]"""
import csv, glob, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
files = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit(f"no kernel_stats.csv under {d}")
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':<78} {'calls':>6} {'total ms':>10} {'avg ms':>9} {'%':>6}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = r["Name"].replace("void ", "")
    print(f"{name[:78]:<78} {r['Calls']:>6} {float(r['TotalDurationNs']) / 1e6:>10.3f} {float(r['AverageNs']) / 1e6:>9.4f} {100 * float(r['TotalDurationNs']) / tot:>6.1f}")
