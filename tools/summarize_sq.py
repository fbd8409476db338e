#!/usr/bin/env python3
"""Per-kernel SQ counter fractions from a `rocprofv3 --kernel-trace --pmc SQ_...` pass of bench.py:
  python tools/summarize_sq.py <round-tag> <dir given to rocprofv3 -d> "<command line, for the record>"
Writes profiles/<tag>_sq_counters.json: per launch averages; fractions are of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main():
    tag, directory, command = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fqd::" in r["Kernel_Name"]:
                vals[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"tag": tag, "command": command,
           "note": "per launch averages; fractions are of SQ_WAVE_CYCLES (WAIT_ANY = wave parked on s_waitcnt/barrier, ACTIVE_INST_ANY = issuing)",
           "kernels": {}}
    for k, c in sorted(vals.items()):
        avg = {n: sum(v) / len(v) for n, v in c.items()}
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        rec = {"SQ_WAVE_CYCLES": round(wc)}
        for n, v in sorted(avg.items()):
            if n != "SQ_WAVE_CYCLES" and wc:
                rec[n + "_frac"] = round(v / wc, 4)
        out["kernels"][k] = rec
    (ROOT / "profiles" / f"{tag}_sq_counters.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
