#!/usr/bin/env python3
"""The ordered single-end run on an ORDINARY gzip input (host/pgzip.hpp against zlib's gzread):
  python tools/ordered_gzip_probe.py [--reads 30000000] [--dir /dev/shm/fqd_og]"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=30_000_000)
    ap.add_argument("--dir", default="/dev/shm/fqd_og")
    ap.add_argument("--quick", action="store_true", help="host reader: the default and zlib only")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--settle", type=float, default=0.0, help="seconds to wait before every timed run (the driver clears what the process before gave back)")
    ap.add_argument("--rocprof", default="", help="directory for a rocprofv3 --kernel-trace --stats run of the device path")
    ap.add_argument("--out-gz", action="store_true", help="write out.fq.gz (deflated on the GPU in the resident run) instead of a plain file")
    a = ap.parse_args()
    import numpy as np
    from fastq_dupaway_amd import _lib
    import config4_at_size as c4
    d = Path(a.dir); d.mkdir(parents=True, exist_ok=True)
    n, L = a.reads, 150
    rng = np.random.default_rng(5)
    pool = rng.integers(0, 4, size=(n * 8 // 10, L), dtype=np.uint8)                 # 20 % of the reads repeat an earlier one
    plain = d / "in.fq"
    with open(plain, "wb") as f:
        step = 1_000_000
        for lo in range(0, n, step):
            cnt = min(step, n - lo)
            rec = np.empty((cnt, 12 + L + 1 + 2 + L + 1), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            idx = np.arange(lo, lo + cnt, dtype=np.int64)
            for p in range(9):
                rec[:, 10 - p] = 48 + (idx % 10); idx //= 10
            rec[:, 11] = 10
            pick = rng.integers(0, len(pool), size=cnt)
            rec[:, 12:12 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[pool[pick]]
            rec[:, 12 + L] = 10; rec[:, 13 + L] = ord("+"); rec[:, 14 + L] = 10
            rec[:, 15 + L:15 + 2 * L] = ord("I"); rec[:, 15 + 2 * L] = 10
            f.write(rec.tobytes())
    gz = d / "in.fq.gz"
    c4.ordinary_gzip(plain, gz)
    print(f"{plain.stat().st_size / 1e9:.2f} GB of FASTQ -> {gz.stat().st_size / 1e9:.2f} GB of ordinary gzip", flush=True)
    plain.unlink()
    said = {}
    sfx = ".gz" if a.out_gz else ""
    # first the default (round 4: the file to HBM as it lies on disk, inflated there by fqd_gunzip), twice, with its stages
    for rep in range(a.reps):
        out = d / ("out_dev.fq" + sfx)
        out.unlink(missing_ok=True)
        time.sleep(a.settle)
        t0 = time.perf_counter()
        r = subprocess.run([str(_lib.CLI_PATH), "-i", str(gz), "-o", str(out), "--fast", "-v"], capture_output=True, text=True,
                           env=dict(os.environ, FQD_HOST_TIMING="1", FQD_GUNZIP_TRACE="1"))
        dt = time.perf_counter() - t0
        said["device"] = (r.returncode, r.stdout, 0 if a.out_gz else out.stat().st_size)
        print(f"inflated on the GPU: rc={r.returncode} {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s | {r.stdout.strip()}", flush=True)
        keep = [l for l in r.stderr.splitlines() if "[host timing]" in l or ("[gunzip" in l and "decoded again" not in l)]
        print("\n".join(keep[:60] if rep else keep[:40]), flush=True)
        print(f"units decoded again from the true boundary: {sum('decoded again' in l for l in r.stderr.splitlines())}", flush=True)
    if a.rocprof:
        # one more run under rocprofv3 (the binary itself after `--`): which kernel the time between the stages belongs to
        out = d / ("out_prof.fq" + sfx)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", a.rocprof, "-o", "cli", "--output-format", "csv", "--",
                            str(_lib.CLI_PATH), "-i", str(gz), "-o", str(out), "--fast", "-v"], capture_output=True, text=True)
        print(f"under rocprofv3: rc={r.returncode} | {r.stdout.strip()[-200:]}", flush=True)
        out.unlink(missing_ok=True)
    for pg in ("1", "0") if a.quick else ("1", "1:12", "1:16", "1:4", "0"):
        out = d / (f"out{pg[0]}.fq" + sfx)
        out.unlink(missing_ok=True)
        env = dict(os.environ, FQD_PGZIP=pg[0], FQD_GUNZIP_ORDINARY_DEVICE="0")
        if ":" in pg:
            env["FQD_PGZIP_THREADS"] = pg.split(":")[1]
        t0 = time.perf_counter()
        r = subprocess.run([str(_lib.CLI_PATH), "-i", str(gz), "-o", str(out), "--fast", "-v"], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        said[pg] = (r.returncode, r.stdout, 0 if a.out_gz else out.stat().st_size)
        print(f"FQD_PGZIP={pg} (reader on/off[:threads]): rc={r.returncode} {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s | {r.stdout.strip()}", flush=True)
    print("same lines and output size:", len(set(said.values())) == 1)
    for f in d.iterdir():
        f.unlink()


if __name__ == "__main__":
    main()
