"""Host side of the `--unordered` join without a GPU (fastq-dupaway_amd/host/id_join.cpp): the
reference's end-of-file rule (hash_dup_remover.hpp:279-340) applied to the full join
the device hands back, and the k-th-with-k-th pairing of repeated IDs, against the oracle's
merge-join (oracle/fqd_oracle.cpp: join_sorted) — exhaustively on small tag sets."""
import itertools
import random
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
SRC = HERE / "native" / "join_check.cpp"
EXE = HERE / "native" / "join_check"
HOST = ROOT / "fastq-dupaway_amd" / "host"


@pytest.fixture(scope="module")
def join_check():
    deps = [SRC, HOST / "id_join.cpp", HOST / "id_join.hpp", HOST / "records.cpp", HOST / "records.hpp", HOST / "file_io.cpp"]
    if not EXE.exists() or EXE.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(EXE), str(SRC),
                        str(HOST / "id_join.cpp"), str(HOST / "records.cpp"), str(HOST / "file_io.cpp"),
                        "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)

    def run(cases):
        """cases: list of (tags_a, tags_b); one process per call, many cases per process would need framing —
        the exhaustive test below batches by spawning once per case set of modest size."""
        out = []
        for a, b in cases:
            text = "".join(f"{len(t)}\n" + "".join((x.hex() or "-") + "\n" for x in t) for t in (a, b))
            res = subprocess.run([str(EXE)], input=text.encode(), capture_output=True, check=True).stdout.decode().splitlines()
            parsed = {}
            for line in res:
                head, _, tail = line.partition(":")
                mode, pairs, unmatched = head.split()
                parsed[mode] = (int(pairs), int(unmatched), [tuple(map(int, p.split(","))) for p in tail.split()])
            out.append(parsed)
        return out
    return run


def tag_arrays(tags):
    lens = np.array([len(t) for t in tags], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) if len(tags) else np.zeros(0, np.uint64)
    data = np.frombuffer(b"".join(tags) + b"\0" * 16, dtype=np.uint8).copy()
    return data, offs, lens


def oracle_join(oracle, a, b, tail):
    if not a or not b:
        return 0, 0, []
    i1, i2, un = oracle.join_tags(*tag_arrays(a), *tag_arrays(b), tail_rule=tail)
    return len(i1), un, list(zip(i1.tolist(), i2.tolist()))


def check(join_check, oracle, cases):
    for (a, b), got in zip(cases, join_check(cases)):
        for mode, tail in (("tail", True), ("full", False)):
            n, un, pairs = oracle_join(oracle, a, b, tail)
            if not tail and a and b:
                assert un == len(a) + len(b) - 2 * n
            assert got[mode] == (n, un, pairs), (mode, a, b, got[mode], (n, un, pairs))


def test_tail_rule_and_repeated_ids_exhaustive_small(join_check, oracle):
    """Every pair of tag multisets over a 3-letter alphabet with up to 3 records per file (repeats included)."""
    alphabet = [b"a", b"b", b"c"]
    lists = [list(c) for k in range(0, 4) for c in itertools.product(alphabet, repeat=k)]
    rnd = random.Random(3)
    cases = [(a, b) for a in lists for b in lists]
    rnd.shuffle(cases)
    check(join_check, oracle, cases[:350])


def test_tail_rule_random_with_orphans_and_repeats(join_check, oracle):
    rnd = random.Random(11)
    cases = []
    for _ in range(150):
        universe = [b"%d" % rnd.randint(0, 40) for _ in range(rnd.randint(1, 12))]
        a = [rnd.choice(universe) for _ in range(rnd.randint(0, 9))]
        b = [rnd.choice(universe) for _ in range(rnd.randint(0, 9))]
        cases.append((a, b))
    # the advisor's example: A = [x, x, x], B = [x]
    cases += [([b"x", b"x", b"x"], [b"x"]), ([b"x"], [b"x", b"x"]), ([b"1", b"2", b"3", b"91"], [b"%d" % k for k in range(1, 10)] + [b"91"])]
    check(join_check, oracle, cases)
