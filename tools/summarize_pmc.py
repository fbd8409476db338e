#!/usr/bin/env python3
"""Summarises rocprofv3 outputs of `bench.py` into profiles/: per-kernel durations from a
--kernel-trace --stats run and HBM traffic from two --pmc passes (FETCH_SIZE, WRITE_SIZE;
they do not fit one pass: MI355X_MICROARCH.md §rocprofv3 PMC slots).

  python tools/summarize_pmc.py <round-tag> <stats_dir> <fetch_dir> <write_dir>

Corrections applied (MI355X_MICROARCH.md §HBM): counter unit = KiB; on gfx950 FETCH_SIZE
reports exactly half of a wide coalesced streaming read, so the fetch of the streaming kernels
(encode, partition histogram/scatter passes) is doubled; the dedup/insert kernels mix streaming
with random 8/64-byte traffic for which the factor is uncalibrated: their fetch is reported raw
and the doubled value is given as an upper bound.
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def counter_avgs(directory, name):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "fqd::" in r["Kernel_Name"]:
                out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    rows = []
    for f in glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        (prof / f"{tag}_kernel_stats.csv").write_text(open(f).read())
    fetch = counter_avgs(fetch_dir, "FETCH_SIZE")
    write = counter_avgs(write_dir, "WRITE_SIZE")
    summary = {"tag": tag, "unit": "bytes per launch", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        streaming = any(t in k for t in ("encode", "bulk_hist", "bulk_scatter"))     # wide coalesced readers
        f_raw = fetch.get(k, 0.0) * 1024
        w = write.get(k, 0.0) * 1024
        f_corr = f_raw * 2 if streaming else f_raw
        summary["kernels"][k] = {
            "FETCH_SIZE_raw": round(f_raw), "WRITE_SIZE": round(w),
            "fetch_corrected": round(f_corr), "hbm_traffic": round(f_corr + w),
            "note": ("FETCH_SIZE doubled: wide coalesced stream on gfx950" if streaming else
                     "random 8/64-byte traffic: FETCH_SIZE factor uncalibrated, raw value used; "
                     f"upper bound with the streaming factor: {round(f_raw * 2 + w)}")}
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        if name in summary["kernels"]:
            summary["kernels"][name]["avg_ns"] = float(r["AverageNs"])
            summary["kernels"][name]["calls"] = int(r["Calls"])
    (prof / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
