"""The `fastq-dupaway` CLI of the MI355X build, tested the way the reference tests its own
binary (reference test/test_basic.py, test_fast.py, test_unordered.py: run the executable,
check the exit code, byte-compare output files) plus byte-for-byte agreement with the CPU
oracle's file drivers on FASTQ inputs, error cases included.

CPU part (`-m "not gpu"`): option surface, validation messages, exit codes, loud failure
without a GPU.  GPU part: everything that dedups.
"""
import filecmp
import gzip
import os
import random
import subprocess

import numpy as np
import pytest

import fastq_dupaway_amd as fqd
from fastq_dupaway_amd import _lib
from oracle_binding import FASTA, FASTQ


@pytest.fixture(scope="module")
def exe():
    if not _lib.CLI_PATH.exists():
        fqd.build_native("all")
    return str(_lib.CLI_PATH)


def run(exe, *args, env=None, cwd=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, *map(str, args)], capture_output=True, env=e, cwd=cwd)
    r.stdout = r.stdout.decode("latin-1")          # no newline translation: a reported '\r' must survive
    r.stderr = r.stderr.decode("latin-1")
    return r


# ---------------------------------------------------------------- CPU: option surface

def test_help_goes_to_stderr_with_exit_code_1(exe):
    # reference test/test_basic.py:11-22 (main.cpp:85-90,185-186)
    r = run(exe, "-h")
    assert r.returncode == 1
    assert r.stderr.startswith("fastq-dupaway V")
    for opt in ("--input-1", "--input-2", "--output-1", "--output-2", "--mem-limit", "--format", "--compare-seq",
                "--distance", "--write-clusters", "--fast", "--unordered", "--verbose"):
        assert opt in r.stderr
    assert r.stdout == ""


@pytest.mark.parametrize("args,msg", [
    (["-i", "a", "-o", "b", "-u", "c", "--fast"], "Both input-2 and output-2 arguments are required for paired-end mode!"),
    (["-i", "a", "-o", "b", "-p", "c", "--fast"], "Both input-2 and output-2 arguments are required for paired-end mode!"),
    (["-i", "a", "-u", "a", "-o", "b", "-p", "c", "--fast"], "Paired input files should not be the same file!"),
    (["-i", "a", "-u", "x", "-o", "b", "-p", "b", "--fast"], "Paired output files should not be the same file!"),
    (["-i", "a", "-o", "b", "--format", "sam", "--fast"], 'Only "fastq" or "fasta" file formats are supported!'),
    (["-i", "a", "-o", "b", "--compare-seq", "fuzzy"], "Unsupported compare-seq type provided!"),
    (["-i", "a", "-o", "b", "-m", "100", "--fast"], "Value of unsupported range provided for --mem-limit option!"),
    (["-i", "a", "-o", "b", "-m", "20000", "--fast"], "Value of unsupported range provided for --mem-limit option!"),
    (["-i", "a", "-o", "b", "--fast", "--compare-seq", "loose"], "--fast mode was enabled, but argument(s) for sequence-based mode were provided!"),
    (["-i", "a", "-o", "b", "--fast", "--distance", "3"], "--fast mode was enabled, but argument(s) for sequence-based mode were provided!"),
    (["-i", "a", "-o", "b", "--fast", "--write-clusters"], "--fast mode was enabled, but argument(s) for sequence-based mode were provided!"),
    (["-i", "a", "-o", "b", "--unordered"], "--unordered argument can only be used with --fast mode!"),
    (["-i", "a", "-o", "b", "--fast", "--unordered"], "--unordered argument can only be used with paired inputs!"),
    (["-o", "b", "--fast"], "the option '--input-1' is required but missing"),
    (["-i", "a", "--fast"], "the option '--output-1' is required but missing"),
    (["-i", "a", "-o", "b", "--bogus"], "unrecognised option '--bogus'"),
    (["-i", "a", "-o", "b", "-m", "abc", "--fast"], "the argument ('abc') for option '--mem-limit' is invalid"),
    (["-i"], "the required argument for option '--input-1' is missing"),
])
def test_argument_errors(exe, args, msg):
    # main.cpp:94-177: message under "An error occured during arguments parsing:", exit 1
    r = run(exe, *args)
    assert r.returncode == 1
    assert r.stderr == "An error occured during arguments parsing:\n" + msg + "\n"


def test_sequence_based_modes_are_refused(exe, tmp_path):
    r = run(exe, "-i", "a", "-o", tmp_path / "b")
    assert r.returncode == 1
    assert r.stderr.startswith("An error occured during fastq-dupaway execution:\n")
    assert "--fast mode only" in r.stderr


def test_option_spellings(exe, tmp_path):
    # --name=value, -xVALUE and unambiguous prefixes parse as in Boost.program_options;
    # the run itself then fails on the missing input with the reference's text
    out = tmp_path / "o.fq"
    for args in (["--input-1=/nonexistent/in.fq", f"--output-1={out}", "--fast"],
                 ["-i/nonexistent/in.fq", f"-o{out}", "--fast", "-v"],
                 ["--input-1", "/nonexistent/in.fq", "--output-1", out, "--fas", "--verb"]):
        r = run(exe, *args)
        assert r.returncode == 1
        assert r.stderr == ("Cannot open file /nonexistent/in.fq\nAn error occured during fastq-dupaway execution:\n"
                            "File does not exist or cannot be opened!\n")
        assert out.exists() and out.stat().st_size == 0          # outputs are created first (hash_dup_remover.hpp:110)
        out.unlink()


def test_no_gpu_is_a_loud_error(exe, tmp_path):
    import ctypes as C
    n = C.c_int(0)
    if fqd.load_library().fqd_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    src = tmp_path / "in.fa"; src.write_bytes(b">1\nACGT\n")
    r = run(exe, "-i", src, "-o", tmp_path / "o.fa", "--format", "fasta", "--fast")
    assert r.returncode == 1
    assert "An error occured during fastq-dupaway execution:" in r.stderr


# ---------------------------------------------------------------- GPU: the reference's own tests

FAST = ("--format", "fasta", "--fast")


@pytest.mark.gpu
def test_single_fast(exe, golden_dir, tmp_path):
    # reference test/test_fast.py:7-26
    fx = golden_dir / "reference_fixtures"
    out = tmp_path / "single_fast.fa"
    r = run(exe, "-i", fx / "inputs" / "single_fast.fa", "-o", out, *FAST)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, fx / "expected" / "single_fast.fa", shallow=False)


@pytest.mark.gpu
def test_paired_fast(exe, golden_dir, tmp_path):
    # reference test/test_fast.py:29-57
    fx = golden_dir / "reference_fixtures"
    o1, o2 = tmp_path / "paired_fast_r1.fa", tmp_path / "paired_fast_r2.fa"
    r = run(exe, "-i", fx / "inputs" / "paired_fast_r1.fa", "-u", fx / "inputs" / "paired_fast_r2.fa",
            "-o", o1, "-p", o2, *FAST)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(o1, fx / "expected" / "paired_fast_r1.fa", shallow=False)
    assert filecmp.cmp(o2, fx / "expected" / "paired_fast_r2.fa", shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["shuffled", "skewed", "deletion", "interleaved", "not_overlapped"])
@pytest.mark.parametrize("full_join", ["0", "1"])
@pytest.mark.parametrize("mode", ["memory", "resident", "twopass"])
def test_unordered(exe, golden_dir, tmp_path, name, full_join, mode):
    # reference test/test_unordered.py:7-48 (both join rules, and both ways of running the join, reproduce the fixtures)
    fx = golden_dir / "reference_fixtures"
    o1, o2 = tmp_path / "r1.fa", tmp_path / "r2.fa"
    r = run(exe, "-i", fx / "inputs" / f"unordered_{name}_r1.fa", "-u", fx / "inputs" / f"unordered_{name}_r2.fa",
            "-o", o1, "-p", o2, *FAST, "--unordered", env={"FQD_FULL_JOIN": full_join, "FQD_UNORDERED_MODE": mode}, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(o1, fx / "expected" / f"unordered_{name}_r1.fa", shallow=False)
    assert filecmp.cmp(o2, fx / "expected" / f"unordered_{name}_r2.fa", shallow=False)


# ---------------------------------------------------------------- GPU: FASTQ, against the oracle's drivers

def fastq(recs, qual=None):
    out = []
    for k, (i, s) in enumerate(recs):
        q = (qual[k] if qual else b"I" * len(s))
        out.append(b"@" + i + b"\n" + s + b"\n+\n" + q + b"\n")
    return b"".join(out)


def random_reads(rnd, n, pool, lo, hi, alphabet=b"ACGTN"):
    seqs = [bytes(rnd.choice(alphabet) for _ in range(rnd.randrange(lo, hi + 1))) for _ in range(pool)]
    return [rnd.choice(seqs) for _ in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("block_mb", [None, "1"])
def test_se_fastq_matches_oracle_bytes(exe, oracle, tmp_path, block_mb):
    rnd = random.Random(31)
    seqs = random_reads(rnd, 30000, 6000, 0, 160)
    quals = [bytes(rnd.choice(b"!#5?IJ") for _ in s) for s in seqs]
    src = tmp_path / "in.fq"
    src.write_bytes(fastq([(b"r%07d some comment" % k, s) for k, s in enumerate(seqs)], quals))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq"
    tot, dup = oracle.filter_single(src, exp, FASTQ)
    r = run(exe, "-i", src, "-o", got, "--fast", "-v", env={"FQD_BLOCK_MB": block_mb} if block_mb else None)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(got, exp, shallow=False)
    assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n"      # hpp:146-147


@pytest.mark.gpu
def test_se_uniform_150bp_fastq(exe, oracle, tmp_path):
    # fixed-size records: the host picks the uniform descriptor and the LDS-staged encoder
    rng = np.random.default_rng(8)
    n, L = 50000, 150
    pool = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=(n // 2, L), p=[.245, .245, .245, .245, .02])
    recs = [(b"r%09d" % k, pool[rng.integers(0, len(pool))].tobytes()) for k in range(n)]
    src = tmp_path / "in.fq"; src.write_bytes(fastq(recs))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq"
    oracle.filter_single(src, exp, FASTQ)
    r = run(exe, "-i", src, "-o", got, "--fast", env={"FQD_BLOCK_MB": "4"})
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(got, exp, shallow=False)


@pytest.mark.gpu
def test_pe_fastq_matches_oracle_bytes(exe, oracle, tmp_path):
    rnd = random.Random(41)
    n = 20000
    s1 = random_reads(rnd, n, 1500, 1, 120)
    s2 = random_reads(rnd, n, 30, 1, 120)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(fastq([(b"p%06d/1" % k, s) for k, s in enumerate(s1)]))
    f2.write_bytes(fastq([(b"p%06d/2" % k, s) for k, s in enumerate(s2[: n - 7])]))      # shorter second file
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tot, dup, _ = oracle.filter_paired(f1, f2, e1, e2, FASTQ)
    assert tot == n - 7
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "-v", env={"FQD_BLOCK_MB": "1"})
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
    assert r.stdout == f"{tot} read pairs processed, out of which {dup} duplicates were removed.\n"   # hpp:253-254


STREAM_ENV = {"memory": {"FQD_UNORDERED_MODE": "memory"},
              # the bounded-memory ways (hash_dup_remover.cpp), in 1 MiB blocks and 64 KiB output windows:
              # resident = one pass, the text of both files stays in HBM, outputs assembled there window by window;
              # twopass  = only tags + sequences stay in HBM, the inputs are read again, windows go through temporary files
              "resident": {"FQD_UNORDERED_MODE": "resident", "FQD_BLOCK_MB": "1", "FQD_STREAM_WINDOW_KB": "64"},
              "twopass": {"FQD_UNORDERED_MODE": "twopass", "FQD_BLOCK_MB": "1", "FQD_STREAM_WINDOW_KB": "64"}}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["memory", "resident", "twopass"])
@pytest.mark.parametrize("style", ["illumina", "sra", "slash"])
@pytest.mark.parametrize("full_join", ["0", "1"])
def test_unordered_fastq_matches_oracle_bytes(exe, oracle, tmp_path, style, full_join, mode):
    rnd = random.Random(51)
    n = 3000
    seqs1 = random_reads(rnd, n, 400, 20, 60); seqs2 = random_reads(rnd, n, 10, 20, 60)

    def ident(k, mate):
        if style == "illumina":
            return b"M01:7:FC:1:%d:%d:%d %d:N:0:ACGT" % (1100 + k % 7, 1000 + k, 2000 + 3 * k, mate)
        if style == "sra":
            return b"SRR99.%d len=%d" % (k + 1, 50 + mate)
        return b"read%d/%d" % (k, mate)                    # no space: never joins
    r1 = [(ident(k, 1), seqs1[k]) for k in range(n)]
    r2 = [(ident(k, 2), seqs2[k]) for k in range(n)]
    del r1[100:130]; del r2[2000:2050]                     # orphans on both sides
    rnd.shuffle(r2)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(fastq(r1)); f2.write_bytes(fastq(r2))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tail = full_join == "0"
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=tail)
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", env={"FQD_FULL_JOIN": full_join, **STREAM_ENV[mode]},
            cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
    assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                        f"{un} Non-matching entries from both files were skipped.\n")                # hpp:342-346
    assert [p.name for p in tmp_path.iterdir() if p.is_dir()] == []                                  # the temporary directory is gone
    if style == "slash":
        assert tot == 0
    else:
        assert tot > 2000


@pytest.mark.gpu
def test_gz_in_and_out(exe, oracle, tmp_path):
    rnd = random.Random(61)
    # > 4 MiB of output: batches of BGZF members deflated on worker threads, written in order
    raw = fastq([(b"g%05d" % k, s) for k, s in enumerate(random_reads(rnd, 60000, 30000, 30, 80))])
    src = tmp_path / "in.fq.gz"
    with gzip.open(src, "wb") as f:
        f.write(raw)
    plain = tmp_path / "in.fq"; plain.write_bytes(raw)
    exp = tmp_path / "exp.fq"; oracle.filter_single(plain, exp, FASTQ)
    got = tmp_path / "got.fq.gz"
    r = run(exe, "-i", src, "-o", got, "--fast")
    assert r.returncode == 0, r.stderr
    assert gzip.open(got, "rb").read() == exp.read_bytes()
    assert subprocess.run(["gzip", "-t", str(got)]).returncode == 0
    # the same ordinary .gz through the several-thread reader (host/pgzip.hpp; files this small go through zlib otherwise)
    got2 = tmp_path / "got2.fq"
    r = run(exe, "-i", src, "-o", got2, "--fast", env={"FQD_PGZIP_MIN_MB": "0"})
    assert r.returncode == 0, r.stderr
    assert got2.read_bytes() == exp.read_bytes()
    # the oracle (zlib's gzread, as Boost's gzip_decompressor) reads the multi-member file back identically
    again = tmp_path / "again.fq"
    oracle.filter_single(got, again, FASTQ)
    assert again.read_bytes() == exp.read_bytes()
    # and the CLI reads its own BGZF output back (members inflated on several threads): nothing left to remove
    third = tmp_path / "third.fq"
    r = run(exe, "-i", got, "-o", third, "--fast", "-v")
    assert r.returncode == 0, r.stderr
    assert third.read_bytes() == exp.read_bytes()
    assert "out of which 0 duplicates were removed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["memory", "resident", "twopass"])
@pytest.mark.parametrize("full_join", ["0", "1"])
def test_unordered_repeated_ids_pair_rank_by_rank(exe, oracle, tmp_path, full_join, mode):
    """IDs repeated inside one file and inside both (every copy used to pair with the same
    partner): the k-th copy in file 1 pairs with the k-th in file 2, leftovers are counted as
    non-matching — the oracle's stable merge-join, byte for byte, -v lines included."""
    rnd = random.Random(71)
    seqs = random_reads(rnd, 400, 50, 20, 40)
    ids1 = [b"x"] * 3 + [b"y"] + [b"dup%d" % (k % 17) for k in range(120)] + [b"u%d" % k for k in range(60)]
    ids2 = [b"x"] + [b"y"] * 4 + [b"dup%d" % (k % 23) for k in range(150)] + [b"u%d" % k for k in range(30, 90)]
    rnd.shuffle(ids1); rnd.shuffle(ids2)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(fastq([(i + b" 1", seqs[k % 400]) for k, i in enumerate(ids1)]))
    f2.write_bytes(fastq([(i + b" 2", seqs[(7 * k) % 400]) for k, i in enumerate(ids2)]))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=(full_join == "0"))
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", env={"FQD_FULL_JOIN": full_join, **STREAM_ENV[mode]},
            cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
    assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                        f"{un} Non-matching entries from both files were skipped.\n")
    assert tot > 100 and un > 50


def write_fastq_np(path, ids, seq_codes, gz):
    """FASTQ of len(ids) records from a uint8 matrix of bases, written in slabs (keeps Python out of the
    per-record loop for the base and quality lines)."""
    import io
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if gz else (lambda p: open(p, "wb"))
    L = seq_codes.shape[1]
    qual = b"I" * L
    with opener(path) as f:
        for lo in range(0, len(ids), 100000):
            buf = io.BytesIO()
            rows = seq_codes[lo:lo + 100000]
            for k, row in enumerate(rows):
                buf.write(b"@"); buf.write(ids[lo + k]); buf.write(b"\n"); buf.write(row.tobytes()); buf.write(b"\n+\n"); buf.write(qual); buf.write(b"\n")
            f.write(buf.getvalue())


@pytest.mark.gpu
def test_config4_shape_gz_unordered_2m_pairs(exe, oracle, tmp_path):
    """BASELINE configs[4] at a size the oracle finishes in seconds: paired-end FASTQ, .gz in AND
    out, --unordered, Illumina-style IDs, file 2 shuffled, orphans on both sides, >= 2 M pairs:
    CLI output bytes (inflated) and -v lines equal the oracle's (hash_dup_remover.hpp:257-347;
    reference test/test_unordered.py:7-48 at 10 records)."""
    rng = np.random.default_rng(404)
    n, L = 2_150_000, 100
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    parent = rng.integers(0, n, size=n)
    is_dup = rng.random(n) < 0.2
    src = np.where(is_dup, np.minimum(parent, np.arange(n)), np.arange(n))     # ~20 % copies of an earlier pair's mate 1
    s1 = pool[src]
    s2 = np.ascontiguousarray(pool[src][:, ::-1])                                 # mate 2 follows mate 1: pair duplicates
    flip = rng.random(n) < 0.5
    s2[is_dup & flip, 0] = ord("N")                                               # half of them differ in mate 2 only
    ids1 = [b"A00123:45:HXXXXXXX:%d:%d:%d:%d 1:N:0:ACGT" % (1 + k % 4, 1101 + (k // 4) % 60, 1000 + (k * 7) % 30000, 1000 + k) for k in range(n)]
    ids2 = [i[:-10] + b"2:N:0:ACGT" for i in ids1]
    keep1 = np.ones(n, bool); keep1[rng.integers(0, n, 60000)] = False           # orphans on both sides
    keep2 = np.ones(n, bool); keep2[rng.integers(0, n, 60000)] = False
    keep1[-1] = keep2[-1] = True                                                  # both files end on the largest ID
    i1 = np.nonzero(keep1)[0]; i2 = np.nonzero(keep2)[0]
    i2 = i2[rng.permutation(len(i2))]                                             # file 2 in another order
    f1, f2 = tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz"
    write_fastq_np(f1, [ids1[k] for k in i1], s1[i1], True)
    write_fastq_np(f2, [ids2[k] for k in i2], s2[i2], True)
    e1, e2 = tmp_path / "e1.fq", tmp_path / "e2.fq"
    g1, g2 = tmp_path / "g1.fq.gz", tmp_path / "g2.fq.gz"
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=True)
    assert tot >= 2_000_000 and dup > 50_000 and un > 50_000
    # round 1's way (inputs also held in host memory), then under "-m 500" (the inputs inflate to ~1 GB) the
    # default — one pass with the text resident in HBM only — and the two-pass fallback with 125 MB output windows through temporary files in a directory created in the working
    # directory (main.cpp:192) and removed at exit
    # ... and the default again with the `.gz` members deflated on the GPU (what happens when no level is asked for)
    for env, extra in (({"FQD_UNORDERED_MODE": "memory"}, []), ({}, ["-m", "500"]), ({"FQD_UNORDERED_MODE": "twopass"}, ["-m", "500"]),
                       ({"FQD_GZ_DEVICE": "1"}, ["-m", "500"])):
        for g in (g1, g2):
            g.unlink(missing_ok=True)
        r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", *extra, env={"FQD_GZ_LEVEL": "1", "FQD_HOST_TIMING": "1", **env}, cwd=tmp_path)
        assert r.returncode == 0, r.stderr
        assert ("unordered/resident: survivors out of HBM" in r.stderr) == ("FQD_UNORDERED_MODE" not in env), r.stderr
        if "FQD_GZ_DEVICE" in env:
            assert g1.stat().st_size < 0.45 * e1.stat().st_size and g2.stat().st_size < 0.45 * e2.stat().st_size
        assert ("unordered/stream: pass 2" in r.stderr) == (env.get("FQD_UNORDERED_MODE") == "twopass"), r.stderr
        assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                            f"{un} Non-matching entries from both files were skipped.\n")
        assert [p.name for p in tmp_path.iterdir() if p.is_dir()] == []
        for got, exp in ((g1, e1), (g2, e2)):
            assert subprocess.run(["gzip", "-t", str(got)]).returncode == 0
            with gzip.open(got, "rb") as a, open(exp, "rb") as b:
                while True:
                    x, y = a.read(1 << 24), b.read(1 << 24)
                    assert x == y
                    if not x:
                        break


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fastq", "fasta", "nothing_joins", "windows_of_64k"])
def test_unordered_gz_outputs_deflated_on_the_device(exe, oracle, tmp_path, case):
    """The resident --unordered run with `.gz` outputs and no level asked for: the members come off the GPU
    (fqd_bgzf_deflate) — any gzip reader must inflate them to the oracle's bytes; an output without records
    is still a valid (empty) gzip file."""
    rnd = random.Random(77)
    n = 4000
    seqs = random_reads(rnd, n, 300, 30, 80)
    fasta = case == "fasta"
    sep = b"/" if case == "nothing_joins" else b" "
    r1 = [(b"M01:7:FC:1:%d:%d%s1:N:0" % (1100 + k % 7, 1000 + k, sep), seqs[k]) for k in range(n)]
    r2 = [(b"M01:7:FC:1:%d:%d%s2:N:0" % (1100 + k % 7, 1000 + k, sep), seqs[(k * 7) % n]) for k in range(n)]
    del r1[50:70]
    rnd.shuffle(r2)

    def text(recs):
        return b"".join(b">" + i + b"\n" + q + b"\n" for i, q in recs) if fasta else fastq(recs)
    ext = "fa" if fasta else "fq"
    f1, f2 = tmp_path / f"r1.{ext}", tmp_path / f"r2.{ext}"
    f1.write_bytes(text(r1)); f2.write_bytes(text(r2))
    e1, e2 = tmp_path / f"e1.{ext}", tmp_path / f"e2.{ext}"
    g1, g2 = tmp_path / f"g1.{ext}.gz", tmp_path / f"g2.{ext}.gz"
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTA if fasta else FASTQ, unordered=True, tail_rule=True)
    env = {"FQD_HOST_TIMING": "1"}
    if case == "windows_of_64k":
        env["FQD_STREAM_WINDOW_KB"] = "64"                  # many windows, each ending in a short member
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", *(["--format", "fasta"] if fasta else []), env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "unordered/resident: survivors out of HBM" in r.stderr
    assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                        f"{un} Non-matching entries from both files were skipped.\n")
    assert (tot == 0) == (case == "nothing_joins")
    for got, exp in ((g1, e1), (g2, e2)):
        assert subprocess.run(["gzip", "-t", str(got)]).returncode == 0
        assert gzip.open(got, "rb").read() == exp.read_bytes()
        raw = got.read_bytes()
        assert raw.endswith(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
        if tot:
            assert len(raw) < 0.6 * exp.stat().st_size


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fastq", "fasta", "stored_and_fixed_blocks", "flipped_bit", "bad_record", "no_final_newline",
                                  "plain_gzip", "truncated", "plain_member_inside"])
def test_unordered_bgzf_inputs_inflated_on_the_device(exe, oracle, tmp_path, case):
    """BGZF inputs of the resident --unordered run are inflated and cut into records on the GPU
    (fqd_bgzf_inflate, fqd_scan_records); anything but a well-formed BGZF file of whole records is read the
    host way instead.  Either way: the oracle's bytes and lines, and for broken inputs exactly what the run
    with FQD_GUNZIP_DEVICE=0 says and leaves behind."""
    import zlib
    from inflate_cases import bgzf, member, EOF_MARK
    rnd = random.Random(91)
    n = 6000
    seqs = random_reads(rnd, n, 500, 40, 90)
    fasta = case == "fasta"
    r1 = [(b"M01:7:FC:1:%d:%d 1:N:0" % (1100 + k % 7, 1000 + k), seqs[k]) for k in range(n)]
    r2 = [(b"M01:7:FC:1:%d:%d 2:N:0" % (1100 + k % 7, 1000 + k), seqs[(k * 11) % n]) for k in range(n)]
    del r2[300:340]
    rnd.shuffle(r2)

    def text(recs):
        return b"".join(b">" + i + b"\n" + q + b"\n" for i, q in recs) if fasta else fastq(recs)
    t1, t2 = text(r1), text(r2)
    if case == "bad_record":
        cut = t2.index(b"\n@", len(t2) // 2) + 1
        t2 = t2[:cut] + b"x" + t2[cut + 1:]                          # a record that does not start with '@'
    if case == "no_final_newline":
        t2 = t2[:-1]
    kw = {"stored_and_fixed_blocks": dict(level=0)}.get(case, dict(level=1))
    z1 = bgzf(t1, **kw)
    z2 = bgzf(t2, strategy=zlib.Z_FIXED) if case == "stored_and_fixed_blocks" else bgzf(t2, **kw)
    if case == "flipped_bit":
        z2 = bytearray(z2); z2[len(z2) // 2] ^= 0x10; z2 = bytes(z2)
    if case == "truncated":
        z2 = z2[: len(z2) * 2 // 3]
    if case == "plain_gzip":
        z2 = gzip.compress(t2, 1)
    if case == "plain_member_inside":                        # `cat blocked.gz ordinary.gz blocked.gz`: the host reads it on, as the reference's decompressor does
        third = len(t2) // 3
        z2 = bgzf(t2[:third])[:-len(EOF_MARK)] + gzip.compress(t2[third:2 * third], 6) + bgzf(t2[2 * third:])
    ext = "fa" if fasta else "fq"
    f1, f2 = tmp_path / f"r1.{ext}.gz", tmp_path / f"r2.{ext}.gz"
    f1.write_bytes(z1); f2.write_bytes(z2)
    fmt = ["--format", "fasta"] if fasta else []
    runs = {}
    for gunzip in ("1", "0"):
        g1, g2 = tmp_path / f"g1_{gunzip}.{ext}", tmp_path / f"g2_{gunzip}.{ext}"
        r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", *fmt,
                env={"FQD_HOST_TIMING": "1", "FQD_GUNZIP_DEVICE": gunzip, "FQD_PGZIP_MIN_MB": "0" if gunzip == "1" else "8"}, cwd=tmp_path)
        # (FQD_PGZIP_MIN_MB=0: ordinary gzip inputs of any size go through the several-thread reader, host/pgzip.hpp, in one of
        #  the two runs and through zlib in the other)
        said = "\n".join(l for l in r.stderr.splitlines() if "[host timing]" not in l)
        runs[gunzip] = (r.returncode, r.stdout, said, g1.read_bytes() if g1.exists() else None, g2.read_bytes() if g2.exists() else None)
        if r.returncode == 0:                               # (the stage clocks are printed by runs that finish)
            assert ("inflate + record scan on the GPU" in r.stderr) == (gunzip == "1"), r.stderr    # file 1 is always good BGZF
    assert runs["1"] == runs["0"]
    rc, out, said, b1, b2 = runs["1"]
    if case in ("fastq", "fasta", "stored_and_fixed_blocks", "plain_gzip", "plain_member_inside"):
        p1, p2 = tmp_path / f"p1.{ext}", tmp_path / f"p2.{ext}"
        p1.write_bytes(t1); p2.write_bytes(t2)
        e1, e2 = tmp_path / f"e1.{ext}", tmp_path / f"e2.{ext}"
        tot, dup, un = oracle.filter_paired(p1, p2, e1, e2, FASTA if fasta else FASTQ, unordered=True, tail_rule=True)
        assert rc == 0 and tot > 5000
        assert out == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                       f"{un} Non-matching entries from both files were skipped.\n")
        assert b1 == e1.read_bytes() and b2 == e2.read_bytes()
    elif case in ("flipped_bit", "truncated"):
        assert rc != 0 and "corrupt or truncated" in said
    elif case == "bad_record":
        assert rc != 0 and "Invalid record start character: x" in said


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("case", ["fastq_to_gz", "fastq_to_plain", "fasta", "small_windows", "bad_base", "bad_record", "uneven_pairs",
                                  "flipped_bit", "plain_gzip", "plain_in_gz_out", "plain_in_gz_out_no_final_newline"])
def test_ordered_runs_on_bgzf_inputs_stay_on_the_device(exe, oracle, tmp_path, case, paired):
    """SE / ordered PE with BGZF inputs: files to HBM compressed, inflate + record scan + dedup there, survivors out in
    input order (run_ordered_resident).  Irregular inputs are left to the streaming run untouched.  Either way the
    oracle's bytes and lines, and exactly what FQD_ORDERED_RESIDENT=0 gives."""
    from inflate_cases import bgzf
    rnd = random.Random(17 + paired)
    n = 5000
    fasta = case == "fasta"
    seqs = random_reads(rnd, n, 700, 30, 100)
    r1 = [(b"M01:7:FC:1:%d:%d 1:N:0" % (1100 + k % 7, 1000 + k), seqs[k]) for k in range(n)]
    r2 = [(b"M01:7:FC:1:%d:%d 2:N:0" % (1100 + k % 7, 1000 + k), seqs[(k * 13) % n]) for k in range(n)]
    if case == "uneven_pairs":
        del r2[-7:]
    if case == "bad_base":
        i, q = r1[3000]; r1[3000] = (i, q[:5] + b"x" + q[6:])

    def text(recs):
        return b"".join(b">" + i + b"\n" + q + b"\n" for i, q in recs) if fasta else fastq(recs)
    t = [text(r1), text(r2)]
    if case == "bad_record":
        cut = t[0].index(b"\n@", len(t[0]) // 2) + 1
        t[0] = t[0][:cut] + b"y" + t[0][cut + 1:]
    plain_in = case.startswith("plain_in")
    if case == "plain_in_gz_out_no_final_newline":
        t[0] = t[0][:-1]
    z = list(t) if plain_in else [bgzf(x, level=1) for x in t]
    if case == "flipped_bit":
        b = bytearray(z[0]); b[len(b) // 2] ^= 4; z[0] = bytes(b)
    if case == "plain_gzip":
        z[0] = gzip.compress(t[0], 1)
    S = 2 if paired else 1
    ext = "fa" if fasta else "fq"
    oext = ext if case == "fastq_to_plain" else ext + ".gz"
    ins = [tmp_path / (f"r{s + 1}.{ext}" + ("" if plain_in else ".gz")) for s in range(S)]
    for s in range(S):
        ins[s].write_bytes(z[s])
    fmt = ["--format", "fasta"] if fasta else []
    env0 = {"FQD_HOST_TIMING": "1"}
    if case == "small_windows":
        env0["FQD_STREAM_WINDOW_KB"] = "96"
    runs = {}
    for resident in ("1", "0"):
        outs = [tmp_path / f"g{s + 1}_{resident}.{oext}" for s in range(S)]
        args = ["-i", ins[0], "-o", outs[0]] + (["-u", ins[1], "-p", outs[1]] if paired else [])
        r = run(exe, *args, "--fast", "-v", *fmt, env={**env0, "FQD_ORDERED_RESIDENT": resident}, cwd=tmp_path)
        said = "\n".join(l for l in r.stderr.splitlines() if "[host timing]" not in l)

        def content(p):
            if not p.exists():
                return None
            return gzip.open(p, "rb").read() if str(p).endswith(".gz") else p.read_bytes()
        runs[resident] = (r.returncode, r.stdout, said, [content(p) for p in outs])
        # (plain_gzip: file 1 is an ORDINARY gzip file — inflated on the GPU too, fqd_gunzip, since round 4)
        good = case in ("fastq_to_gz", "fastq_to_plain", "fasta", "small_windows", "plain_in_gz_out", "plain_gzip") or (case == "uneven_pairs" and not paired)
        if r.returncode == 0:
            assert ("ordered/resident: survivors out of HBM" in r.stderr) == (resident == "1" and good), r.stderr
    assert runs["1"] == runs["0"]
    rc, out, said, got = runs["1"]
    if case in ("fastq_to_gz", "fastq_to_plain", "fasta", "small_windows", "uneven_pairs", "plain_gzip", "plain_in_gz_out",
                "plain_in_gz_out_no_final_newline"):
        ps = [tmp_path / f"p{s + 1}.{ext}" for s in range(S)]
        es = [tmp_path / f"e{s + 1}.{ext}" for s in range(S)]
        for s in range(S):
            ps[s].write_bytes(t[s])
        if paired:
            tot, dup, un = oracle.filter_paired(ps[0], ps[1], es[0], es[1], FASTA if fasta else FASTQ)
            line = f"{tot} read pairs processed, out of which {dup} duplicates were removed.\n"
        else:
            tot, dup = oracle.filter_single(ps[0], es[0], FASTA if fasta else FASTQ)
            line = f"{tot} reads processed, out of which {dup} duplicates were removed.\n"
        assert rc == 0 and out == line and tot > 4000
        assert got == [e.read_bytes() for e in es]
    else:
        assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["single", "paired", "unordered"])
@pytest.mark.parametrize("case", ["good", "header_fields", "flipped_bit", "truncated", "two_members", "bad_base", "very_packable"])
def test_ordinary_gzip_inputs_inflated_on_the_device(exe, oracle, tmp_path, case, mode):
    """FQD_GUNZIP_ORDINARY_DEVICE=1: an ordinary `.gz` (one long deflate stream, what gzip / pigz / sequencers write; reference
    file_utils.cpp:59-66 reads it through the same decompressor) goes to HBM as it lies on disk and is inflated THERE
    (fqd_gunzip).  Good files: the oracle's bytes and lines, and the device really did it (its stage shows in the timing).
    Several members (`cat a.gz b.gz`) are walked on the device too.  Anything irregular — damage, a cut file, an unknown base —
    is left to the host reader: exactly what the run with FQD_GUNZIP_ORDINARY_DEVICE=0 gives, message, exit code and partial
    output alike; so is a text that packs more than the eightfold the device reader keeps room for (one read repeated: 200-fold)."""
    from gunzip_cases import header_with_fields, member
    rnd = random.Random(23)
    n = 30000
    seqs = random_reads(rnd, n, 4000, 60, 100)
    if case == "very_packable":
        seqs = [b"ACGTTGCA" * 12] * n
    r1 = [(b"M01:7:FC:1:%d:%d 1:N:0" % (1100 + k % 7, 1000 + k), seqs[k]) for k in range(n)]
    r2 = [(b"M01:7:FC:1:%d:%d 2:N:0" % (1100 + k % 7, 1000 + k), seqs[(k * 13) % n]) for k in range(n)]
    if case == "bad_base":
        i, q = r1[20000]; r1[20000] = (i, q[:5] + b"x" + q[6:])
    if mode == "unordered":
        rnd.shuffle(r2)
    t = [fastq(r1), fastq(r2)]
    z = [member(x, 6, header=header_with_fields() if case == "header_fields" else b"") for x in t]
    if case == "flipped_bit":
        b = bytearray(z[0]); b[len(b) // 2] ^= 4; z[0] = bytes(b)
    if case == "truncated":
        z[0] = z[0][: len(z[0]) * 2 // 3]
    if case == "two_members":
        half = t[0].index(b"\n@", len(t[0]) // 2) + 1
        z[0] = member(t[0][:half], 6) + member(t[0][half:], 1)
    S = 1 if mode == "single" else 2
    ins = [tmp_path / f"r{s + 1}.fq.gz" for s in range(S)]
    for s in range(S):
        ins[s].write_bytes(z[s])
    runs = {}
    for device in ("1", "0"):
        outs = [tmp_path / f"g{s + 1}_{device}.fq.gz" for s in range(S)]
        args = ["-i", ins[0], "-o", outs[0]] + (["-u", ins[1], "-p", outs[1]] if S == 2 else []) + (["--unordered"] if mode == "unordered" else [])
        r = run(exe, *args, "--fast", "-v", env={"FQD_HOST_TIMING": "1", "FQD_GUNZIP_ORDINARY_DEVICE": device, "FQD_GUNZIP_UNIT_KB": "64"}, cwd=tmp_path)
        said = "\n".join(l for l in r.stderr.splitlines() if "[host timing]" not in l)
        content = [gzip.open(p, "rb").read() if p.exists() else None for p in outs] if r.returncode == 0 else [p.exists() for p in outs]
        runs[device] = (r.returncode, r.stdout, said, content)
        if device == "1" and case in ("good", "header_fields", "two_members") and mode != "unordered":
            assert "ordered/resident: survivors out of HBM" in r.stderr, r.stderr       # the resident run took the files: nothing was left to the host reader
    assert runs["1"] == runs["0"]
    rc, out, said, got = runs["1"]
    if case in ("good", "header_fields", "two_members", "very_packable"):
        ps = [tmp_path / f"p{s + 1}.fq" for s in range(S)]
        es = [tmp_path / f"e{s + 1}.fq" for s in range(S)]
        for s in range(S):
            ps[s].write_bytes(t[s])
        if S == 2:
            tot, dup, un = oracle.filter_paired(ps[0], ps[1], es[0], es[1], FASTQ, unordered=mode == "unordered")
        else:
            tot, dup = oracle.filter_single(ps[0], es[0], FASTQ)
        assert rc == 0 and str(tot) in out and tot > 20000
        assert got == [e.read_bytes() for e in es]
    else:
        assert rc != 0


# ---------------------------------------------------------------- GPU: several engines in one run (FQD_DEVICES)

def uniform_fastq(rnd, n, L, pool, ident):
    seqs = [bytes(rnd.choice(b"ACGTN") for _ in range(L)) for _ in range(pool)]
    return fastq([(ident(k), rnd.choice(seqs)) for k in range(n)])


@pytest.mark.gpu
@pytest.mark.parametrize("devices,exchange,slab", [("0,0", "copy", ""), ("0,0,0,0", "copy", ""), ("0", "rccl", ""), ("0", "copy", ""),
                                                    ("0,0,0", "copy", "64"), ("0", "rccl", "16")])
def test_multi_gpu_cli_single_end_matches_oracle(exe, oracle, tmp_path, devices, exchange, slab):
    """The C++ driver's multi-GPU run (run_ordered_multi over the shard group, csrc/fqd_shard.hip): several ranks — here
    ranks sharing the one card, or one rank under real RCCL — batches dealt round-robin (FQD_BLOCK_MB=1 gives many
    pipelined rounds), keys exchanged by hash prefix in fixed-size slabs (slab: slabs far too small, every round
    spills), flags back, survivors written in input order: same bytes and -v line as the oracle."""
    rnd = random.Random(81)
    src = tmp_path / "in.fq"
    src.write_bytes(uniform_fastq(rnd, 60000, 100, 9000, lambda k: b"r%07d" % k))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq"
    tot, dup = oracle.filter_single(src, exp, FASTQ)
    for send_hash in ("0", "1"):                                 # the owners hash arrived keys again / the hashes travel with the keys
        r = run(exe, "-i", src, "-o", got, "--fast", "-v", env={"FQD_DEVICES": devices, "FQD_EXCHANGE": exchange, "FQD_BLOCK_MB": "1", "FQD_SHARD_SLAB": slab,
                                                                 "FQD_SHARD_SEND_HASH": send_hash})
        assert r.returncode == 0, r.stderr
        assert filecmp.cmp(got, exp, shallow=False)
        assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n"
    assert dup > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0,0", "0"])
def test_multi_gpu_cli_paired_matches_oracle(exe, oracle, tmp_path, devices):
    rnd = random.Random(82)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(uniform_fastq(rnd, 40000, 80, 3000, lambda k: b"p%06d/1" % k))
    f2.write_bytes(uniform_fastq(rnd, 40000, 60, 40, lambda k: b"p%06d/2" % k))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tot, dup, _ = oracle.filter_paired(f1, f2, e1, e2, FASTQ)
    for send_hash in ("0", "1"):
        r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "-v", env={"FQD_DEVICES": devices, "FQD_BLOCK_MB": "1", "FQD_SHARD_SEND_HASH": send_hash})
        assert r.returncode == 0, r.stderr
        assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
        assert r.stdout == f"{tot} read pairs processed, out of which {dup} duplicates were removed.\n"
    assert dup > 1000


@pytest.mark.gpu
def test_multi_gpu_cli_errors(exe, oracle, tmp_path):
    rnd = random.Random(83)
    # an unknown base in the middle: output cut at that record, the reference's two lines, exit 1
    recs = [(b"r%05d" % k, bytes(rnd.choice(b"ACGT") for _ in range(50))) for k in range(30000)]
    recs[17017] = (recs[17017][0], recs[17017][1][:20] + b"x" + recs[17017][1][21:])
    src = tmp_path / "bad.fq"; src.write_bytes(fastq(recs))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq"
    with pytest.raises(RuntimeError):
        oracle.filter_single(src, exp, FASTQ)
    r = run(exe, "-i", src, "-o", got, "--fast", env={"FQD_DEVICES": "0,0", "FQD_BLOCK_MB": "1"})
    assert r.returncode == 1
    assert r.stderr == ("Error: unknown character in DNA sequence: x\nAn error occured during fastq-dupaway execution:\n"
                        "Supported sequence character set: {A, N, C, G, T}!\n")
    assert got.read_bytes() == exp.read_bytes()
    rag = tmp_path / "ragged.fq"
    rag.write_bytes(fastq([(b"a", b"ACGT"), (b"b", b"ACGTA"), (b"c", b"ACG")]))
    r = run(exe, "-i", rag, "-o", tmp_path / "o.fq", "--fast", env={"FQD_DEVICES": "0,x"})
    assert r.returncode == 1 and "FQD_DEVICES" in r.stderr
    # the key shape of the group follows the file: reads of other lengths after blocks of one length, then a read longer
    # than the padded width, make the owners lay their keys out again (fqd_widen_keys) — same bytes as the oracle's, which
    # like the reference keys any length at any point (seq_utils.cpp:35-49); round 3 refused both with exit 1
    recs = [(b"u%05d" % k, bytes(rnd.choice(b"ACGT") for _ in range(60))) for k in range(40000)]
    recs[35000] = (recs[35000][0], recs[35000][1][:41])
    for k in range(36000, 40000, 7):                                                        # copies of reads from before the change of shape
        recs[k] = (recs[k][0], recs[k - 30000][1])
    late = tmp_path / "late.fq"; late.write_bytes(fastq(recs))
    tot, dup = oracle.filter_single(late, exp, FASTQ)
    for devices in ("0,0", "0,0,0", "0,0,0,0"):
        for extra in ({}, {"FQD_SHARD_PADDED": "1"}):
            r = run(exe, "-i", late, "-o", got, "--fast", "-v", env={"FQD_DEVICES": devices, "FQD_BLOCK_MB": "1", **extra})
            assert r.returncode == 0 and got.read_bytes() == exp.read_bytes(), (devices, extra, r.stderr)
            assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n" and dup > 500
    recs[35000] = (recs[35000][0], recs[34999][1] + b"ACGTACGTAC")                      # 70 > 64 = the first round's 60 rounded up
    recs[100] = (recs[100][0], recs[100][1][:33])                                           # (the first round is ragged: padded keys)
    recs[38000] = (recs[38000][0], recs[35000][1])                                          # the long read again, and a still longer one
    recs[39000] = (recs[39000][0], recs[35000][1] * 3)
    late.write_bytes(fastq(recs))
    tot, dup = oracle.filter_single(late, exp, FASTQ)
    for devices in ("0,0", "0,0,0", "0,0,0,0"):
        for extra in ({}, {"FQD_SHARD_MAX_LEN": "70"}, {"FQD_SHARD_SLAB": "64"}):
            r = run(exe, "-i", late, "-o", got, "--fast", "-v", env={"FQD_DEVICES": devices, "FQD_BLOCK_MB": "1", **extra})
            assert r.returncode == 0 and got.read_bytes() == exp.read_bytes(), (devices, extra, r.stderr)
            assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n"
    # paired: mate 2 grows late, mate 1 does not
    p1 = [(b"p%05d/1" % k, bytes(rnd.choice(b"ACGT") for _ in range(50))) for k in range(30000)]
    p2 = [(b"p%05d/2" % k, bytes(rnd.choice(b"AC") for _ in range(4)) * 10) for k in range(30000)]
    for k in range(20000, 30000, 5):
        p1[k] = (p1[k][0], p1[k - 15000][1]); p2[k] = (p2[k][0], p2[k - 15000][1])
    p2[25001] = (p2[25001][0], p2[25001][1] + b"GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG")
    p2[26001] = (p2[26001][0], p2[25001][1]); p1[26001] = (p1[26001][0], p1[25001][1])
    f1, f2 = tmp_path / "l1.fq", tmp_path / "l2.fq"
    f1.write_bytes(fastq(p1)); f2.write_bytes(fastq(p2))
    e1, e2, g1, g2 = (tmp_path / x for x in ("le1.fq", "le2.fq", "lg1.fq", "lg2.fq"))
    tot, dup, _ = oracle.filter_paired(f1, f2, e1, e2, FASTQ)
    for devices in ("0,0", "0,0,0"):
        r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "-v", env={"FQD_DEVICES": devices, "FQD_BLOCK_MB": "1"})
        assert r.returncode == 0 and g1.read_bytes() == e1.read_bytes() and g2.read_bytes() == e2.read_bytes(), (devices, r.stderr)
        assert r.stdout == f"{tot} read pairs processed, out of which {dup} duplicates were removed.\n" and dup > 500


def trimmed_fastq(rnd, n, lo, hi, pool, ident):
    seqs = [bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randint(lo, hi))) for _ in range(pool)]
    return fastq([(ident(k), rnd.choice(seqs)) for k in range(n)])


@pytest.mark.gpu
@pytest.mark.parametrize("devices,slab", [("0,0", ""), ("0,0,0,0", "32"), ("0", "")])
def test_multi_gpu_cli_trimmed_reads(exe, oracle, tmp_path, devices, slab):
    """Reads of lengths 30..160: the multi-GPU run takes them as padded keys; single-end and
    paired, same bytes and -v lines as the oracle."""
    rnd = random.Random(91)
    src = tmp_path / "in.fq"
    src.write_bytes(trimmed_fastq(rnd, 50000, 30, 160, 8000, lambda k: b"t%07d" % k))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq"
    tot, dup = oracle.filter_single(src, exp, FASTQ)
    env = {"FQD_DEVICES": devices, "FQD_EXCHANGE": "copy", "FQD_BLOCK_MB": "1", "FQD_SHARD_SLAB": slab}
    r = run(exe, "-i", src, "-o", got, "--fast", "-v", env=env)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(got, exp, shallow=False)
    assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n" and dup > 5000
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(trimmed_fastq(rnd, 30000, 30, 160, 2500, lambda k: b"p%06d/1" % k))
    f2.write_bytes(trimmed_fastq(rnd, 30000, 40, 120, 30, lambda k: b"p%06d/2" % k))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tot, dup, _ = oracle.filter_paired(f1, f2, e1, e2, FASTQ)
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "-v", env=env)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
    assert r.stdout == f"{tot} read pairs processed, out of which {dup} duplicates were removed.\n" and dup > 500


@pytest.mark.gpu
def test_resident_run_announces_a_device_error_before_handing_over(exe, oracle, tmp_path):
    """a GPU error inside the resident ordered run is not swallowed — one line names it, then the streaming
    run does the job (nothing had been written yet), with the oracle's bytes."""
    from inflate_cases import bgzf
    rnd = random.Random(5)
    recs = [(b"h%05d" % k, s) for k, s in enumerate(random_reads(rnd, 4000, 600, 30, 90))]
    src = tmp_path / "in.fq.gz"; src.write_bytes(bgzf(fastq(recs), level=1))
    plain = tmp_path / "in.fq"; plain.write_bytes(fastq(recs))
    exp, got = tmp_path / "exp.fq", tmp_path / "got.fq.gz"
    tot, dup = oracle.filter_single(plain, exp, FASTQ)
    r = run(exe, "-i", src, "-o", got, "--fast", "-v", env={"FQD_TEST_FAIL_RESIDENT": "1"}, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("[fastq-dupaway] the GPU-resident ordered run gave up on a GPU error (GPU engine: forced by FQD_TEST_FAIL_RESIDENT)") == 1
    assert gzip.decompress(got.read_bytes()) == exp.read_bytes()
    assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n"
    r = run(exe, "-i", src, "-o", got, "--fast", "-v", cwd=tmp_path)                 # and without the forced error: not a word
    assert r.returncode == 0 and r.stderr == "" and gzip.decompress(got.read_bytes()) == exp.read_bytes()


# ---------------------------------------------------------------- GPU: --unordered over several GPUs

MULTI = [("0,0", ""), ("0,0,0,0", ""), ("0,0,0", "8"), ("0", "")]      # (FQD_DEVICES, FQD_SHARD_SLAB): ranks sharing the one card; tiny slabs: every pair exchange spills


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["shuffled", "skewed", "deletion", "interleaved", "not_overlapped"])
@pytest.mark.parametrize("full_join", ["0", "1"])
@pytest.mark.parametrize("devices,slab", MULTI)
def test_multi_gpu_unordered_reference_fixtures(exe, golden_dir, tmp_path, name, full_join, devices, slab):
    """reference test/test_unordered.py:7-48 with the tag order cut into one range per rank (run_unordered_multi)."""
    fx = golden_dir / "reference_fixtures"
    o1, o2 = tmp_path / "r1.fa", tmp_path / "r2.fa"
    r = run(exe, "-i", fx / "inputs" / f"unordered_{name}_r1.fa", "-u", fx / "inputs" / f"unordered_{name}_r2.fa",
            "-o", o1, "-p", o2, *FAST, "--unordered",
            env={"FQD_FULL_JOIN": full_join, "FQD_DEVICES": devices, "FQD_EXCHANGE": "copy", "FQD_SHARD_SLAB": slab}, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(o1, fx / "expected" / f"unordered_{name}_r1.fa", shallow=False)
    assert filecmp.cmp(o2, fx / "expected" / f"unordered_{name}_r2.fa", shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["illumina", "sra", "slash"])
@pytest.mark.parametrize("full_join", ["0", "1"])
@pytest.mark.parametrize("devices,slab", MULTI[:3])
def test_multi_gpu_unordered_fastq_matches_oracle_bytes(exe, oracle, tmp_path, style, full_join, devices, slab):
    rnd = random.Random(52)
    n = 3000
    seqs1 = random_reads(rnd, n, 400, 20, 60); seqs2 = random_reads(rnd, n, 10, 20, 60)

    def ident(k, mate):
        if style == "illumina":
            return b"M01:7:FC:1:%d:%d:%d %d:N:0:ACGT" % (1100 + k % 7, 1000 + k, 2000 + 3 * k, mate)
        if style == "sra":
            return b"SRR99.%d len=%d" % (k + 1, 50 + mate)
        return b"read%d/%d" % (k, mate)
    r1 = [(ident(k, 1), seqs1[k]) for k in range(n)]
    r2 = [(ident(k, 2), seqs2[k]) for k in range(n)]
    del r1[100:130]; del r2[2000:2050]
    r1 += r1[500:520]                                      # repeated IDs in file 1: paired rank by rank with file 2's
    rnd.shuffle(r2)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(fastq(r1)); f2.write_bytes(fastq(r2))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=(full_join == "0"))
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v",
            env={"FQD_FULL_JOIN": full_join, "FQD_DEVICES": devices, "FQD_EXCHANGE": "copy", "FQD_SHARD_SLAB": slab, "FQD_BLOCK_MB": "1", "FQD_STREAM_WINDOW_KB": "64"}, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)
    assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                        f"{un} Non-matching entries from both files were skipped.\n")
    assert (tot == 0) if style == "slash" else (tot > 2000)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_multi_gpu_unordered_tail_rule_fuzz(exe, oracle, tmp_path, devices):
    """Tiny inputs (1..9 records per file): most ranges are empty or hold one file only, and the reference's end-of-file
    rule has to look across them."""
    rnd = random.Random(20260303)
    seqs = [b"ACGT", b"GGCC", b"TTAA", b"ACGT", b"NNAC"]
    for case in range(30):
        ids = rnd.sample(range(1, 13), rnd.randrange(1, 10))
        ids2 = rnd.sample(range(1, 13), rnd.randrange(1, 10))
        def fa(idlist, salt):
            return b"".join(b">%02d x\n%s\n" % (i, seqs[(i * salt) % len(seqs)]) for i in idlist)
        f1, f2 = tmp_path / f"a{case}.fa", tmp_path / f"b{case}.fa"
        f1.write_bytes(fa(ids, 1)); f2.write_bytes(fa(ids2, 3))
        for full in ("0", "1"):
            e1, e2, g1, g2 = (tmp_path / f"{x}{case}{full}" for x in ("e1", "e2", "g1", "g2"))
            tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTA, unordered=True, tail_rule=(full == "0"))
            r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--format", "fasta", "--fast", "--unordered", "-v",
                    env={"FQD_FULL_JOIN": full, "FQD_DEVICES": devices, "FQD_EXCHANGE": "copy"})
            assert r.returncode == 0, r.stderr
            assert g1.read_bytes() == e1.read_bytes() and g2.read_bytes() == e2.read_bytes(), (ids, ids2, full)
            assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                                f"{un} Non-matching entries from both files were skipped.\n"), (ids, ids2, full)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,gz", [("0,0,0,0", False), ("0,0", True)])
def test_multi_gpu_unordered_2m_pairs_shuffled_with_orphans(exe, oracle, tmp_path, devices, gz):
    """2 M pairs, Illumina-style IDs, file 2 shuffled, orphans on both sides, mates of different lengths, about a fifth
    duplicate pairs: outputs and -v lines equal the oracle's, plain and `.gz` out."""
    import numpy as np
    rng = np.random.default_rng(77)
    n, L1, L2 = 2_000_000, 100, 76
    pool1 = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n // 4, L1)); pool2 = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(64, L2))
    pick1 = rng.integers(0, n // 4, n); pick2 = rng.integers(0, 64, n)

    def build(mate, pool, pick, L, keep):
        idx = np.flatnonzero(keep)
        ids = np.char.add(np.char.add("@A00:1:FC:1:", np.char.zfill(idx.astype(str), 8)), f" {mate}:N:0\n").astype("S")
        idl = len(ids[0])
        rec = np.empty((len(idx), idl + L + 1 + 2 + L + 1), np.uint8)
        rec[:, :idl] = np.frombuffer(b"".join(ids.tolist()), np.uint8).reshape(len(idx), idl)
        rec[:, idl:idl + L] = pool[pick[idx]]
        rec[:, idl + L] = 10; rec[:, idl + L + 1] = ord("+"); rec[:, idl + L + 2] = 10
        rec[:, idl + L + 3: idl + 2 * L + 3] = ord("F"); rec[:, idl + 2 * L + 3] = 10
        return rec
    keep1 = rng.random(n) > 0.01; keep2 = rng.random(n) > 0.01
    r1 = build(1, pool1, pick1, L1, keep1); r2 = build(2, pool2, pick2, L2, keep2)
    r2 = r2[rng.permutation(len(r2))]
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    r1.tofile(f1); r2.tofile(f2)
    ext = ".fq.gz" if gz else ".fq"
    e1, e2, g1, g2 = tmp_path / "e1.fq", tmp_path / "e2.fq", tmp_path / ("g1" + ext), tmp_path / ("g2" + ext)
    tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=True)
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", "-v", env={"FQD_DEVICES": devices, "FQD_EXCHANGE": "copy"}, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                        f"{un} Non-matching entries from both files were skipped.\n")
    assert tot > 1_900_000 and dup > 20_000 and un > 10_000
    if gz:
        for g, e in ((g1, e1), (g2, e2)):
            assert subprocess.run(f"gzip -dc '{g}' | cmp -s - '{e}'", shell=True).returncode == 0
    else:
        assert filecmp.cmp(g1, e1, shallow=False) and filecmp.cmp(g2, e2, shallow=False)


@pytest.mark.gpu
def test_multi_gpu_unordered_errors(exe, oracle, tmp_path):
    """An unknown base (the output is cut at that pair, in tag order), an empty file, a malformed record: as the
    single-GPU run and the oracle have it."""
    rnd = random.Random(88)
    n = 4000
    r1 = [(b"q.%04d a" % k, bytes(rnd.choice(b"ACGT") for _ in range(40))) for k in range(n)]
    r2 = [(b"q.%04d b" % k, bytes(rnd.choice(b"ACGT") for _ in range(30))) for k in range(n)]
    r2[2500] = (r2[2500][0], r2[2500][1][:10] + b"x" + r2[2500][1][11:])
    rnd.shuffle(r2)
    f1, f2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    f1.write_bytes(fastq(r1)); f2.write_bytes(fastq(r2))
    e1, e2, g1, g2 = (tmp_path / x for x in ("e1.fq", "e2.fq", "g1.fq", "g2.fq"))
    with pytest.raises(RuntimeError):
        oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=True)
    r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", env={"FQD_DEVICES": "0,0,0", "FQD_EXCHANGE": "copy"}, cwd=tmp_path)
    assert r.returncode == 1
    assert r.stderr == ("Error: unknown character in DNA sequence: x\nAn error occured during fastq-dupaway execution:\n"
                        "Supported sequence character set: {A, N, C, G, T}!\n")
    assert g1.read_bytes() == e1.read_bytes() and g2.read_bytes() == e2.read_bytes()
    empty = tmp_path / "empty.fq"; empty.write_bytes(b"")
    single = run(exe, "-i", f1, "-u", empty, "-o", g1, "-p", g2, "--fast", "--unordered", cwd=tmp_path)
    multi = run(exe, "-i", f1, "-u", empty, "-o", g1, "-p", g2, "--fast", "--unordered", env={"FQD_DEVICES": "0,0"}, cwd=tmp_path)
    assert multi.returncode == single.returncode == 1 and multi.stderr == single.stderr


# ---------------------------------------------------------------- GPU: error behaviour

def both(exe, oracle, tmp_path, data: bytes, fmt=FASTQ):
    """Runs CLI and oracle on the same bytes; returns (cli result, cli output, oracle output, oracle error)."""
    src = tmp_path / "in.txt"; src.write_bytes(data)
    exp, got = tmp_path / "exp.out", tmp_path / "got.out"
    err = None
    try:
        oracle.filter_single(src, exp, fmt)
    except RuntimeError as e:
        err = str(e)
    args = ["-i", src, "-o", got, "--fast"] + (["--format", "fasta"] if fmt == FASTA else [])
    r = run(exe, *args)
    return r, got.read_bytes() if got.exists() else None, exp.read_bytes() if exp.exists() else None, err


@pytest.mark.gpu
def test_error_empty_input(exe, oracle, tmp_path, capfd):
    r, got, exp, err = both(exe, oracle, tmp_path, b"")
    assert r.returncode == 1 and err == "Not enough memory to read a single object!"
    assert r.stderr == "An error occured during fastq-dupaway execution:\nNot enough memory to read a single object!\n"
    assert got == exp == b""


@pytest.mark.gpu
def test_missing_final_newline_drops_last_record(exe, oracle, tmp_path):
    r, got, exp, err = both(exe, oracle, tmp_path, fastq([(b"1", b"AC"), (b"2", b"GT"), (b"3", b"AC")])[:-1])
    assert r.returncode == 0 and err is None and got == exp == fastq([(b"1", b"AC"), (b"2", b"GT")])


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [b"Ag", b"AC\r", b"RYKM", b"ac"])
def test_error_unknown_base_keeps_partial_output(exe, oracle, tmp_path, bad, capfd):
    recs = [(b"%d" % k, s) for k, s in enumerate([b"ACGT", b"ACGT", b"GGCC", bad, b"TTTT", b"ACGT"])]
    r, got, exp, err = both(exe, oracle, tmp_path, fastq(recs))
    capfd.readouterr()
    first = next(c for c in bad if c not in b"ACGTN")
    assert r.returncode == 1 and err == "Supported sequence character set: {A, N, C, G, T}!"
    assert r.stderr == (f"Error: unknown character in DNA sequence: {chr(first)}\n"
                        "An error occured during fastq-dupaway execution:\nSupported sequence character set: {A, N, C, G, T}!\n")
    assert got == exp == fastq([recs[0], recs[2]])           # records before the bad one, deduplicated


@pytest.mark.gpu
def test_error_bad_start_character_lookahead(exe, oracle, tmp_path, capfd):
    # the record BEFORE the malformed one is fetched but never written (one-record lookahead)
    data = fastq([(b"1", b"ACGT"), (b"2", b"GGGG"), (b"3", b"TTTT")]) + b"x4\nAC\n+\nII\n" + fastq([(b"5", b"CC")])
    r, got, exp, err = both(exe, oracle, tmp_path, data)
    capfd.readouterr()
    assert r.returncode == 1 and err == "Fastq record should start with @ symbol!"
    assert r.stderr == ("Invalid record start character: x\nAn error occured during fastq-dupaway execution:\n"
                        "Fastq record should start with @ symbol!\n")
    assert got == exp == fastq([(b"1", b"ACGT"), (b"2", b"GGGG")])


@pytest.mark.gpu
def test_error_quality_length_mismatch(exe, oracle, tmp_path, capfd):
    data = fastq([(b"1", b"ACGT"), (b"2", b"GGGG")]) + b"@3\nACGT\n+\nIII\n"
    r, got, exp, err = both(exe, oracle, tmp_path, data)
    capfd.readouterr()
    assert r.returncode == 1 and err == "Sequence and Quality fields of Fastq record should have the same length!"
    assert r.stderr.startswith("Found sequence ACGT of length 5 and quality string III of length 4\n")
    assert got == exp == fastq([(b"1", b"ACGT")])


@pytest.mark.gpu
def test_error_first_record_malformed(exe, oracle, tmp_path, capfd):
    r, got, exp, err = both(exe, oracle, tmp_path, b"ACGT\n", FASTA)
    capfd.readouterr()
    assert r.returncode == 1 and err == "Fasta record should start with > symbol!"
    assert r.stderr.startswith("Invalid record start character: A\n")
    assert got == exp == b""


@pytest.mark.gpu
def test_bad_base_before_malformed_record_wins(exe, oracle, tmp_path, capfd):
    data = fastq([(b"1", b"ACGT"), (b"2", b"AxGT"), (b"3", b"TTTT")]) + b"bad\n"
    r, got, exp, err = both(exe, oracle, tmp_path, data)
    capfd.readouterr()
    assert err == "Supported sequence character set: {A, N, C, G, T}!"
    assert r.returncode == 1 and "unknown character in DNA sequence: x" in r.stderr
    assert got == exp == fastq([(b"1", b"ACGT")])
    # ...but a bad base in the record right before the malformed one is never reached
    data = fastq([(b"1", b"ACGT"), (b"2", b"AxGT")]) + b"bad\n"
    r, got, exp, err = both(exe, oracle, tmp_path, data)
    capfd.readouterr()
    assert err == "Fastq record should start with @ symbol!"
    assert r.returncode == 1 and "Fastq record should start with @ symbol!" in r.stderr
    assert got == exp == fastq([(b"1", b"ACGT")])


@pytest.mark.gpu
def test_unordered_tail_rule_fuzz_against_oracle(exe, oracle, tmp_path):
    """Many tiny --unordered inputs (1..9 records per file, random overlaps) through the CLI and
    the oracle: the reference's end-of-file rule and the full join must both agree
    byte for byte, including the -v counts."""
    rnd = random.Random(20260101)
    seqs = [b"ACGT", b"GGCC", b"TTAA", b"ACGT", b"NNAC"]
    for case in range(45):
        ids = rnd.sample(range(1, 13), rnd.randrange(1, 10))
        ids2 = rnd.sample(range(1, 13), rnd.randrange(1, 10))
        def fa(idlist, salt):
            return b"".join(b">%02d x\n%s\n" % (i, seqs[(i * salt) % len(seqs)]) for i in idlist)
        f1, f2 = tmp_path / f"a{case}.fa", tmp_path / f"b{case}.fa"
        f1.write_bytes(fa(ids, 1)); f2.write_bytes(fa(ids2, 3))
        for full in ("0", "1"):
            e1, e2, g1, g2 = (tmp_path / f"{x}{case}{full}" for x in ("e1", "e2", "g1", "g2"))
            tot, dup, un = oracle.filter_paired(f1, f2, e1, e2, FASTA, unordered=True, tail_rule=(full == "0"))
            r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--format", "fasta", "--fast", "--unordered", "-v",
                    env={"FQD_FULL_JOIN": full})
            assert r.returncode == 0, r.stderr
            assert g1.read_bytes() == e1.read_bytes() and g2.read_bytes() == e2.read_bytes(), (ids, ids2, full)
            assert r.stdout == (f"{tot} valid read pairs processed, out of which {dup} duplicates were removed.\n"
                                f"{un} Non-matching entries from both files were skipped.\n"), (ids, ids2, full)


@pytest.mark.gpu
def test_differential_fuzz_single_and_paired_with_injected_errors(exe, oracle, tmp_path):
    """Random small inputs — clean, or with one defect somewhere (unknown base, bad lead byte,
    quality of another length, a line missing, no final newline, CRLF) — through the CLI and the
    oracle's file driver: same exit status, same exception text, same bytes on disk."""
    rnd = random.Random(2024)

    def records(n, fmt):
        out = []
        for k in range(n):
            s = bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(1, 70)))
            if rnd.random() < 0.3 and out:
                s = rnd.choice(out)[1]
            out.append((b"id%d" % k, s))
        return out

    def render(recs, fmt, defect):
        parts = []
        for (i, s) in recs:
            parts.append(b"@" + i + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" if fmt == FASTQ else b">" + i + b"\n" + s + b"\n")
        k = rnd.randrange(len(parts))
        if defect == "base":
            i, s = recs[k]
            s = s[: len(s) // 2] + rnd.choice([b"x", b"a", b"R", b"."]) + s[len(s) // 2 + 1:]
            parts[k] = b"@" + i + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" if fmt == FASTQ else b">" + i + b"\n" + s + b"\n"
        elif defect == "lead":
            parts[k] = b"?" + parts[k][1:]
        elif defect == "qual" and fmt == FASTQ:
            parts[k] = parts[k][:-1] + b"I\n"
        elif defect == "line":
            parts[k] = parts[k].split(b"\n", 1)[1]
        data = b"".join(parts)
        if defect == "nonl":
            data = data[:-1]
        if defect == "crlf":
            data = data.replace(b"\n", b"\r\n")
        return data

    defects = [None, None, "base", "lead", "qual", "line", "nonl", "crlf"]
    for case in range(48):
        fmt = rnd.choice([FASTQ, FASTA])
        paired = case % 3 == 0
        fmt_args = ["--format", "fasta"] if fmt == FASTA else []
        d = tmp_path / f"c{case}"; d.mkdir()
        n = rnd.randrange(2, 40)
        f1 = d / "r1.txt"; f1.write_bytes(render(records(n, fmt), fmt, rnd.choice(defects)))
        err = None
        if paired:
            f2 = d / "r2.txt"; f2.write_bytes(render(records(rnd.randrange(2, 40), fmt), fmt, rnd.choice(defects)))
            e1, e2, g1, g2 = (d / x for x in ("e1", "e2", "g1", "g2"))
            try:
                oracle.filter_paired(f1, f2, e1, e2, fmt)
            except RuntimeError as ex:
                err = str(ex)
            r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", *fmt_args)
            pairs = [(g1, e1), (g2, e2)]
        else:
            e1, g1 = d / "e1", d / "g1"
            try:
                oracle.filter_single(f1, e1, fmt)
            except RuntimeError as ex:
                err = str(ex)
            r = run(exe, "-i", f1, "-o", g1, "--fast", *fmt_args)
            pairs = [(g1, e1)]
        what = (case, fmt, paired, f1.read_bytes()[:200])
        assert (r.returncode != 0) == (err is not None), (what, r.stderr, err)
        if err is not None:
            assert r.stderr.endswith("An error occured during fastq-dupaway execution:\n" + err + "\n"), (what, r.stderr, err)
        for g, e in pairs:
            assert g.exists() == e.exists(), what
            if e.exists():
                assert g.read_bytes() == e.read_bytes(), what


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["memory", "resident", "twopass"])
def test_unordered_differential_fuzz_with_injected_errors(exe, oracle, tmp_path, mode):
    """--unordered on random small paired inputs — clean, or with one defect in either file (unknown base,
    bad lead byte, quality of another length, a line missing, empty file) — through the CLI (held in memory
    / streamed twice) and the oracle's driver: same exit status, same exception text, same output files.
    (With an unknown base the reference has already written the pairs before it in tag order.)"""
    rnd = random.Random(77)

    def render(ids, seqs, mate, defect):
        parts = [b"@" + i + b" %d\n" % mate + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in zip(ids, seqs)]
        if not parts:
            return b""
        k = rnd.randrange(len(parts))
        if defect == "base":
            s = seqs[k]; s = s[: len(s) // 2] + b"x" + s[len(s) // 2 + 1:]
            parts[k] = b"@" + ids[k] + b" %d\n" % mate + s + b"\n+\n" + b"I" * len(s) + b"\n"
        elif defect == "lead":
            parts[k] = b"?" + parts[k][1:]
        elif defect == "qual":
            parts[k] = parts[k][:-1] + b"I\n"
        elif defect == "line":
            parts[k] = parts[k].split(b"\n", 1)[1]
        elif defect == "empty":
            return b""
        return b"".join(parts)

    defects = [None, None, None, "base", "lead", "qual", "line", "empty"]
    pool = [bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(1, 50))) for _ in range(12)]
    for case in range(36):
        d = tmp_path / f"c{case}"; d.mkdir()
        universe = [b"q%03d" % k for k in range(rnd.randrange(2, 30))]
        ids1 = rnd.sample(universe, rnd.randrange(1, len(universe) + 1)); ids2 = rnd.sample(universe, rnd.randrange(1, len(universe) + 1))
        f1, f2 = d / "r1.fq", d / "r2.fq"
        f1.write_bytes(render(ids1, [rnd.choice(pool) for _ in ids1], 1, rnd.choice(defects)))
        f2.write_bytes(render(ids2, [rnd.choice(pool) for _ in ids2], 2, rnd.choice(defects)))
        e1, e2, g1, g2 = (d / x for x in ("e1", "e2", "g1", "g2"))
        err = None
        try:
            oracle.filter_paired(f1, f2, e1, e2, FASTQ, unordered=True, tail_rule=True)
        except RuntimeError as ex:
            err = str(ex)
        r = run(exe, "-i", f1, "-u", f2, "-o", g1, "-p", g2, "--fast", "--unordered", env=STREAM_ENV[mode], cwd=d)
        what = (case, f1.read_bytes()[:150], f2.read_bytes()[:150])
        assert (r.returncode != 0) == (err is not None), (what, r.stderr, err)
        if err is not None:
            assert r.stderr.endswith("An error occured during fastq-dupaway execution:\n" + err + "\n"), (what, r.stderr, err)
        for g, e in ((g1, e1), (g2, e2)):
            assert g.exists() == e.exists(), what
            if e.exists():
                assert g.read_bytes() == e.read_bytes(), what


@pytest.mark.gpu
def test_records_longer_than_the_reader_headroom(exe, oracle, tmp_path):
    """Records of ~1.3 MB with 4 MiB input blocks: what a block carries over to the next one (the record cut
    by the block end plus the one before it) exceeds the 1 MiB the two-stage reader leaves free in front of
    the raw bytes, so the slow re-allocation path of RecordStream::finish runs; output == the oracle's."""
    rnd = random.Random(91)
    seqs = [bytes(rnd.choice(b"ACGT") for _ in range(1000)) * 1300 for _ in range(3)]
    recs = [(b"big%d" % k, seqs[k % 3][: 1_300_000 - 7 * (k % 2)]) for k in range(9)]
    src = tmp_path / "big.fa"
    src.write_bytes(b"".join(b">" + i + b"\n" + s + b"\n" for i, s in recs))
    exp, got = tmp_path / "exp.fa", tmp_path / "got.fa"
    tot, dup = oracle.filter_single(src, exp, FASTA)
    r = run(exe, "-i", src, "-o", got, "--fast", "--format", "fasta", "-v", env={"FQD_BLOCK_MB": "4"})
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(got, exp, shallow=False)
    assert r.stdout == f"{tot} reads processed, out of which {dup} duplicates were removed.\n" and dup > 0


@pytest.mark.gpu
def test_pipes_as_input_and_output(exe, oracle, tmp_path):
    """Not regular files: the input comes through /dev/stdin (no size, no pread), the output goes
    to /dev/stdout."""
    rnd = random.Random(77)
    raw = fastq([(b"p%05d" % k, s) for k, s in enumerate(random_reads(rnd, 5000, 900, 10, 90))])
    src = tmp_path / "in.fq"; src.write_bytes(raw)
    exp = tmp_path / "exp.fq"; oracle.filter_single(src, exp, FASTQ)
    r = subprocess.run([exe, "-i", "/dev/stdin", "-o", "/dev/stdout", "--fast"], input=raw, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == exp.read_bytes()
