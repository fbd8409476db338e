#!/usr/bin/env python3
"""End-to-end (file in -> file out) timing of the CLI next to the CPU oracle's file driver.
Not the bench metric (that is bench.py, HBM-resident input): this number includes file
reading, record scanning, PCIe and output writing and is bound by the host.
  python tools/e2e_bench.py [--reads N] [--paired] [--dir /tmp]
"""
import argparse
import filecmp
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def write_fastq(path, n, L, rng, pool_idx, pool, tag):
    with open(path, "wb") as f:
        step = 500_000
        for a in range(0, n, step):
            m = min(step, n - a)
            ids = np.char.add(np.char.add(b"@r", np.char.zfill(np.arange(a, a + m).astype("S9"), 9)), tag.encode())
            rec = np.empty((m, 12 + len(tag) + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
            idw = 11 + len(tag)
            rec[:, :idw] = np.frombuffer(b"".join(ids.tolist()), dtype=np.uint8).reshape(m, idw)
            rec[:, idw] = 10
            rec[:, idw + 1: idw + 1 + L] = pool[pool_idx[a:a + m]]
            rec[:, idw + 1 + L] = 10
            rec[:, idw + 2 + L] = ord("+"); rec[:, idw + 3 + L] = 10
            rec[:, idw + 4 + L: idw + 4 + 2 * L] = ord("I")
            rec[:, idw + 4 + 2 * L] = 10
            f.write(rec[:, : idw + 5 + 2 * L].tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--unordered", action="store_true", help="paired: shuffle the records of file 2 and run --unordered")
    ap.add_argument("--gz", action="store_true", help="gzip-compress inputs (with the CLI itself: BGZF) and outputs")
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--oracle-reads", type=int, default=2_000_000)
    ap.add_argument("--host-threads", default="", help="comma list: also time the CLI under FQD_HOST_THREADS=<each>")
    a = ap.parse_args()
    from fastq_dupaway_amd import _lib
    from oracle import binding
    rng = np.random.default_rng(1)
    n, L = a.reads, 150
    pool = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(int(n * 0.8) + 1, L))
    idx = rng.integers(0, len(pool), size=n)
    d = Path(a.dir)
    f1 = d / "e2e_r1.fq"; write_fastq(f1, n, L, rng, idx, pool, (" 1:N" if a.unordered else "/1") if a.paired else "")
    files = [f1]
    if a.paired:
        f2 = d / "e2e_r2.fq"; write_fastq(f2, n, L, rng, rng.integers(0, len(pool), size=n), pool, " 2:N" if a.unordered else "/2"); files.append(f2)
    if a.unordered:                              # fixed-size records: shuffle file 2 as rows
        rec = files[1].stat().st_size // n
        rows = np.fromfile(files[1], dtype=np.uint8).reshape(n, rec)
        rows[rng.permutation(n)].tofile(files[1])
        del rows
    size = sum(f.stat().st_size for f in files)
    plain_files = list(files)
    ext = ".fq.gz" if a.gz else ".fq"
    if a.gz:                                     # BGZF inputs, packed by the host driver's own file layer
        packer = Path("/tmp") / f"e2e_bgzf_pack_{os.getpid()}"      # not in --dir: /dev/shm is mounted noexec
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(packer), str(ROOT / "tools" / "bgzf_pack.cpp"),
                        str(ROOT / "fastq-dupaway_amd" / "host" / "file_io.cpp"), "-lz", "-lpthread"], check=True)
        for k, f in enumerate(files):
            subprocess.run([str(packer), str(f), str(d / f"e2e_in{k}.fq.gz")], check=True)
        files = [d / f"e2e_in{k}.fq.gz" for k in range(len(files))]
        print("compressed inputs:", [f.stat().st_size for f in files])
    outs = [d / f"e2e_out{k}{ext}" for k in range(len(files))]
    args = [str(_lib.CLI_PATH), "-i", str(files[0]), "-o", str(outs[0]), "--fast", "-v"]
    if a.paired:
        args += ["-u", str(files[1]), "-p", str(outs[1])]
    if a.unordered:
        args += ["--unordered"]
    for threads in [t for t in a.host_threads.split(",") if t] + [""]:
        env = dict(os.environ, FQD_HOST_THREADS=threads) if threads else dict(os.environ)
        for rep in range(2):
            for o in outs:                       # a fresh output file each time (rewriting an existing one pays an extra flush at close on ext4)
                o.unlink(missing_ok=True)
            t0 = time.perf_counter(); r = subprocess.run(args, capture_output=True, text=True, env=env); dt = time.perf_counter() - t0
            print(f"cli run {rep} (host threads {threads or 'default'}): rc={r.returncode} {dt:.2f} s  {n / dt / 1e6:.2f} Mreads/s  "
                  f"{size / dt / 1e9:.2f} GB/s  | {r.stdout.strip()} {r.stderr.strip()[:2000]}")
    oracle = binding.load_oracle()
    files = plain_files
    m = n if a.unordered else min(a.oracle_reads, n)         # a prefix of shuffled files would not pair up
    if m < n:                                   # the oracle is timed on a prefix (one CPU core)
        rec_bytes = files[0].stat().st_size // n
        for k, f in enumerate(files):
            with open(f, "rb") as src, open(d / f"e2e_prefix{k}.fq", "wb") as dst:
                dst.write(src.read(rec_bytes * m))
        files = [d / f"e2e_prefix{k}.fq" for k in range(len(files))]
    exp = [d / f"e2e_exp{k}.fq" for k in range(len(files))]
    t0 = time.perf_counter()
    if a.paired:
        oracle.filter_paired(files[0], files[1], exp[0], exp[1], binding.FASTQ, unordered=a.unordered)
    else:
        oracle.filter_single(files[0], exp[0], binding.FASTQ)
    dt = time.perf_counter() - t0
    print(f"oracle (1 core, {m} reads): {dt:.2f} s  {m / dt / 1e6:.3f} Mreads/s")
    if m == n:
        if a.gz:
            import gzip
            print("outputs identical:", all(gzip.open(o, "rb").read() == e.read_bytes() for o, e in zip(outs, exp)))
        else:
            print("outputs identical:", all(filecmp.cmp(o, e, shallow=False) for o, e in zip(outs, exp)))
    for f in list(d.glob("e2e_*")):
        f.unlink()


if __name__ == "__main__":
    main()
