#!/usr/bin/env python3
"""A/B legs of the device phase: runs `bench.py --config se` once per variant (an environment of FQD_* knobs the
engine reads at start-up), each in its own process, and prints step time and per-kernel averages side by side.
Usage: python tools/ab_bench.py [--steps 10] [name=ENV1=v,ENV2=v ...]   (no variants given: the built-in list)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DEFAULT = [
    ("as built", {}),
    ("dedup: eight lanes x 8 bytes per candidate", {"FQD_DEDUP_VL": "0"}),
    ("encoder: 14 KB of LDS more per workgroup (3 per CU)", {"FQD_ENC_EXTRA_LDS": "14336"}),
]


def main():
    args = sys.argv[1:]
    steps, repeat = "10", 1
    while args and args[0] in ("--steps", "--repeat"):
        if args[0] == "--steps": steps = args[1]
        else: repeat = int(args[1])
        args = args[2:]
    variants = DEFAULT
    if args:
        variants = []
        for a in args:
            name, _, envs = a.partition("=")
            variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
    rows = []
    for name, env in [v for _ in range(repeat) for v in variants]:      # the whole list, `repeat` times over: runs of one variant are spread out
        env = dict(env)
        extra = env.pop("ARGS", "").split()                      # ARGS=--no-verify: further bench.py arguments of a variant (diagnostic builds)
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "se", "--steps", steps, "--warmup", "2", "--cpu-sample", "0", *extra],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        if r.returncode != 0 or line is None:
            rows.append((name, None, r.stderr[-400:]))
            print(f"{name}: FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)
            continue
        rec = json.loads(line)
        k = rec["roofline"]["kernels"]
        row = {"variant": name, "env": env, "ms_per_step": rec["ms_per_step"], "frac": rec["roofline"]["frac"], "parity": rec["parity"][:40],
               **{f"{x}_ms": k[x]["avg_ms"] for x in k}}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if repeat > 1:
        import statistics
        for name, _ in variants:
            mine = [r for r in rows if isinstance(r, dict) and r["variant"] == name]
            if mine:
                print(json.dumps({"variant": name, "runs": len(mine), "ms_per_step_min": min(r["ms_per_step"] for r in mine),
                                  "ms_per_step_median": statistics.median(r["ms_per_step"] for r in mine),
                                  **{k + "_median": round(statistics.median(r[k] for r in mine), 4) for k in ("encode_ms", "partition_ms", "dedup_ms") if k in mine[0]}}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
