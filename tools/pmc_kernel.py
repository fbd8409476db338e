#!/usr/bin/env python3
"""Counter totals of the kernels whose name contains a word, from `rocprofv3 --kernel-trace --pmc ... --output-format csv`
passes:  python tools/pmc_kernel.py WORD DIR [DIR ...]   (per launch averages; ratios to SQ_WAVE_CYCLES where that helps)"""
import collections
import csv
import glob
import sys


def main():
    word, dirs = sys.argv[1], sys.argv[2:]
    vals = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if word in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            for disp in per.values():
                for n, v in disp.items():
                    vals[n].append(v)
    avg = {n: sum(v) / len(v) for n, v in vals.items()}
    wc = avg.get("SQ_WAVE_CYCLES")
    for n in sorted(avg):
        extra = f"   {avg[n] / wc:8.4f} of SQ_WAVE_CYCLES" if wc and n != "SQ_WAVE_CYCLES" else ""
        print(f"{n:28s} {avg[n]:16.0f}  ({len(vals[n])} launches){extra}")


if __name__ == "__main__":
    main()
