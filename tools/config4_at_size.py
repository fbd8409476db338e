#!/usr/bin/env python3
"""BASELINE configs[4] at its stated size: N pairs of 2x150 bp FASTQ, both files .gz (BGZF), file 2 in
another order, `--fast --unordered`, one MI355X, through the CLI.

The CPU oracle cannot hold this size in minutes, so the run is checked through what the workload's
construction gives in closed form (the IDs are r%09d, so tag order = generator index order and the
generator's analytically known keep flags ARE the expected survivors):
  * -v lines: pairs processed = N, duplicates = number of zero flags, non-matching = 0;
  * outputs: `gzip -t` clean, exactly (N - duplicates) records each, IDs strictly increasing, the ID list of
    output 1 == the IDs of the flagged survivors == the ID list of output 2, mates kept together.
Prints a progress line per stage (append-friendly for gpurun_out logs).

  python tools/config4_at_size.py [--pairs 100000000] [--dir /dev/shm/fqd_c4] [--gz-level 1]
"""
import argparse
import shlex
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


def _gzip_piece(args):
    import zlib
    path, lo, n = args
    with open(path, "rb") as f:
        f.seek(lo)
        raw = f.read(n)
    c = zlib.compressobj(1, zlib.DEFLATED, 31)                  # a whole gzip member (header, trailer), level 1
    return c.compress(raw) + c.flush()


def ordinary_gzip(plain, out, piece=256 << 20):
    """`plain` as an ordinary .gz: gzip members of `piece` bytes of text each, one after the other (what `cat a.gz b.gz`
    gives; readers take it as one stream), packed by a pool of processes."""
    import multiprocessing as mp
    size = plain.stat().st_size
    jobs = [(str(plain), lo, min(piece, size - lo)) for lo in range(0, size, piece)]
    with mp.get_context("fork").Pool(min(16, len(jobs))) as pool, open(out, "wb") as f:
        for member in pool.imap(_gzip_piece, jobs):
            f.write(member)


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--gzip-helper":   # (a process of its own, which has never touched the GPU, forks the packers)
        ordinary_gzip(Path(sys.argv[2]), Path(sys.argv[3]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100_000_000)
    ap.add_argument("--dir", default="/dev/shm/fqd_c4")
    ap.add_argument("--gz-level", default="device", help="outputs: 'device' (the default run: members deflated on the GPU) or a level 1..9 "
                                                           "of the host codec (the reference's default is 6); inputs are packed at level 1 either way")
    ap.add_argument("--mem-limit", default="", help="-m value in MB (default: the CLI's 2048)")
    ap.add_argument("--keep-inputs", action="store_true", help="leave r1.fq.gz / r2.fq.gz in --dir (for a profiler run of the CLI on them)")
    ap.add_argument("--no-check", action="store_true", help="skip reading the outputs back")
    ap.add_argument("--plain-gzip", action="store_true", help="inputs as ORDINARY gzip (members of 256 MB of text made by zlib level 1, no BGZF fields: "
                                                              "what gzip / pigz / a sequencer's software write): no member-parallel reader can split them")
    ap.add_argument("--settle", type=float, default=0.0, help="seconds to wait before the first timed run (the kernel clears the device memory a process that "
                                                              "has just exited gave back; an allocation waits for that clearing)")
    ap.add_argument("--wrap", default="", help="command prefix for the --also runs that carry WRAP=1, ({i} = the run's index) e.g. 'rocprofv3 --kernel-trace --output-format csv -d DIR/{i} -o c4 --'")
    ap.add_argument("--also", default="", help="further timed runs on the same inputs, each under extra environment settings: "
                                               "'FQD_HOST_THREADS=8;FQD_HOST_THREADS=32,FQD_GZ_LEVEL=6' (only their -v lines are checked)")
    a = ap.parse_args()
    import torch
    from fastq_dupaway_amd import Engine, _lib
    n, L = a.pairs, 150
    d = Path(a.dir); d.mkdir(parents=True, exist_ok=True)
    dev = torch.device("cuda", 0)
    rec_len = 18 + L + 1 + 2 + L + 1                         # "@r%09d 1:N:0\n" + seq\n + "+\n" + qual\n
    packer = Path("/tmp") / f"fqd_bgzf_pack_{os.getpid()}"       # /dev/shm is mounted noexec
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(packer), str(ROOT / "tools" / "bgzf_pack.cpp"),
                    str(ROOT / "fastq-dupaway_amd" / "host" / "file_io.cpp"), "-lz", "-lpthread"], check=True)
    rng = np.random.default_rng(4)
    expect = None
    gz = []
    with Engine(segments=2, device=0) as e:
        for mate in range(2):
            plain = d / f"r{mate + 1}.fq"
            bases = torch.empty(n * L + 16, dtype=torch.uint8, device=dev)
            exp = torch.empty(n, dtype=torch.uint8, device=dev) if mate == 1 else None
            e.synth_reads(404, 0, n, L, 200, mate, bases, exp)
            e.sync()
            seqs = bases[: n * L].cpu().numpy().reshape(n, L)
            if exp is not None:
                expect = exp.cpu().numpy()
            del bases, exp
            torch.cuda.empty_cache()
            order = rng.permutation(n) if mate == 1 else None  # file 2 in another order
            with open(plain, "wb") as f:
                step = 2_000_000
                for lo in range(0, n, step):
                    cnt = min(step, n - lo)
                    idx = np.arange(lo, lo + cnt, dtype=np.int64) if order is None else order[lo:lo + cnt].astype(np.int64)
                    rec = np.empty((cnt, rec_len), dtype=np.uint8)
                    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
                    x = idx.copy()
                    for p in range(9):
                        rec[:, 10 - p] = 48 + (x % 10); x //= 10
                    rec[:, 11:18] = np.frombuffer(b" %d:N:0\n" % (mate + 1), dtype=np.uint8)
                    rec[:, 18:18 + L] = seqs[idx]
                    rec[:, 18 + L] = 10; rec[:, 19 + L] = ord("+"); rec[:, 20 + L] = 10
                    rec[:, 21 + L:21 + 2 * L] = ord("I"); rec[:, 21 + 2 * L] = 10
                    f.write(rec.tobytes())
                    if lo % (10 * step) == 0:
                        log(f"file {mate + 1}: {lo + cnt} records written")
            del seqs
            out = d / f"r{mate + 1}.fq.gz"
            t0 = time.perf_counter()
            if a.plain_gzip:
                subprocess.run([sys.executable, str(Path(__file__).resolve()), "--gzip-helper", str(plain), str(out)], check=True)
            else:
                subprocess.run([str(packer), str(plain), str(out)], check=True, env=dict(os.environ, FQD_GZ_LEVEL="1"))
            log(f"file {mate + 1}: {plain.stat().st_size / 1e9:.1f} GB -> {out.stat().st_size / 1e9:.2f} GB {'ordinary gzip' if a.plain_gzip else 'BGZF'} in {time.perf_counter() - t0:.0f} s")
            plain.unlink()
            gz.append(out)
    dups = int((expect == 0).sum())
    outs = [d / "o1.fq.gz", d / "o2.fq.gz"]
    cmd = [str(_lib.CLI_PATH), "-i", str(gz[0]), "-u", str(gz[1]), "-o", str(outs[0]), "-p", str(outs[1]), "--fast", "--unordered", "-v"]
    if a.mem_limit:
        cmd += ["-m", a.mem_limit]
    if a.settle > 0:
        import gc
        gc.collect(); torch.cuda.empty_cache()
        time.sleep(a.settle)
    t0 = time.perf_counter()
    base_env = dict(os.environ, FQD_HOST_TIMING="1")
    base_env.pop("FQD_GZ_LEVEL", None)
    if a.gz_level != "device":
        base_env["FQD_GZ_LEVEL"] = a.gz_level
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(d), env=base_env)
    dt = time.perf_counter() - t0
    log(f"CLI rc={r.returncode}  {dt:.1f} s  {n / dt / 1e6:.3f} M pairs/s   ({' '.join(cmd[1:])}; outputs: {a.gz_level}; "
        f"{outs[0].stat().st_size / 1e9:.2f} + {outs[1].stat().st_size / 1e9:.2f} GB)")
    print(r.stdout, end="")
    for line in r.stderr.splitlines():
        if any(w in line for w in ("unordered", "on the GPU", "process:", "survivors", "engine timing")) or "error" in line.lower():
            print(line)
    ok = r.returncode == 0 and r.stdout == (f"{n} valid read pairs processed, out of which {dups} duplicates were removed.\n"
                                             f"0 Non-matching entries from both files were skipped.\n")
    log("-v lines == closed form:", ok)
    for run, extra in enumerate(filter(None, a.also.split(";"))):
        env = dict(base_env)
        env.update(kv.split("=", 1) for kv in extra.split(","))
        wrap = shlex.split(a.wrap.replace("{i}", str(run))) if env.pop("WRAP", "") else []   # "WRAP=1,...": this run under --wrap's prefix (a profiler)
        if "SLEEP" in env:                                    # "SLEEP=8,...": let the memory the run before gave back be cleared first
            time.sleep(float(env.pop("SLEEP")))
        for o in outs:                                            # (a run writes new files; emptying 10 GB of tmpfs is not its work)
            o.unlink(missing_ok=True)
        t1 = time.perf_counter()
        r2 = subprocess.run(wrap + cmd, capture_output=True, text=True, cwd=str(d), env=env)
        dt2 = time.perf_counter() - t1
        stages = "; ".join(" ".join(l.split("] ", 1)[1].split("  (")[0].split()) for l in r2.stderr.splitlines() if any(w in l for w in ("unordered", "on the GPU", "process:", "survivors", "engine timing")))
        log(f"also [{extra}]: rc={r2.returncode} {dt2:.1f} s = {n / dt2 / 1e6:.3f} M pairs/s, same -v lines: {r2.stdout == r.stdout} | {stages}")
        ok &= r2.returncode == 0 and r2.stdout == r.stdout
    # outputs: ids of every record, in order
    want = np.nonzero(expect)[0]
    for k, o in enumerate([] if a.no_check else outs):
        t0 = time.perf_counter()
        p = subprocess.Popen(["gzip", "-dc", str(o)], stdout=subprocess.PIPE, bufsize=1 << 24)
        got = np.empty(len(want) + 8, dtype=np.int64); m = 0; good = True
        leftover = b""
        while True:
            chunk = p.stdout.read(rec_len * 200000)
            if not chunk:
                break
            chunk = leftover + chunk
            whole = len(chunk) // rec_len * rec_len
            leftover = chunk[whole:]
            rows = np.frombuffer(chunk[:whole], dtype=np.uint8).reshape(-1, rec_len)
            ids = np.zeros(len(rows), dtype=np.int64)
            for c in range(2, 11):
                ids = ids * 10 + (rows[:, c].astype(np.int64) - 48)
            good &= bool((rows[:, 0] == ord("@")).all() and (rows[:, 12] == 49 + k).all() and (rows[:, 18 + L] == 10).all())
            if m + len(ids) > len(got):
                good = False; break
            got[m:m + len(ids)] = ids; m += len(ids)
        p.wait()
        same = good and not leftover and m == len(want) and bool((got[:m] == want).all())
        ok &= same and p.returncode == 0
        log(f"output {k + 1}: {m} records (expected {len(want)}), survivors' IDs in tag order == closed form: {same}  ({time.perf_counter() - t0:.0f} s)")
    log("RESULT:", "ok" if ok else "MISMATCH", f"| {n} pairs, {dups} duplicate pairs removed, {dt:.1f} s end to end, {n / dt / 1e6:.3f} M pairs/s")
    for f in ([] if a.keep_inputs else gz) + outs:
        f.unlink(missing_ok=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
