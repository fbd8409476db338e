#!/usr/bin/env python3
"""profiles/<tag>_sq_counters.json from one `rocprofv3 --kernel-trace --pmc SQ_... --output-format csv` run:
  python tools/summarize_sq.py <tag> <dir given to rocprofv3 -d> "<command line profiled>" """
import collections, csv, glob, json, sys
from pathlib import Path
tag, d, cmd = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fqd::" in k or "anonymous" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"tag": tag, "command": cmd, "note": "per launch averages; fractions are of SQ_WAVE_CYCLES (WAIT_ANY = wave parked on s_waitcnt/barrier, ACTIVE_INST_ANY = issuing)", "kernels": {}}
for k, c in sorted(acc.items()):
    avg = {n: sum(v) / len(v) for n, v in c.items()}
    wc = avg.get("SQ_WAVE_CYCLES", 0) or 1
    out["kernels"][k] = {"SQ_WAVE_CYCLES": round(avg.get("SQ_WAVE_CYCLES", 0)), **{n + "_frac": round(v / wc, 4) for n, v in sorted(avg.items()) if n != "SQ_WAVE_CYCLES"}}
Path(f"profiles/{tag}_sq_counters.json").write_text(json.dumps(out, indent=1) + "\n")
for k in ("fqd::encode_staged_kernel<true>", "fqd::bucket_dedup_kernel<true, false, 4>", "fqd::bulk_scatter_kernel<1>", "fqd::bulk_scatter_kernel<2>"):
    print(k, out["kernels"].get(k))
