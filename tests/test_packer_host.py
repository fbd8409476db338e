"""Runs the kernels' own packer (csrc/fqd_device.hpp) on the host and checks it.

The packing decides key equality on the device, so before any GPU time is spent:
  * words equal an independent numpy 2-bit + N-mask packing, at every byte alignment;
  * unknown bytes are reported with the reference's first-bad-byte rule (seq_utils.cpp:3-21);
  * two sequences get equal device keys iff the oracle's base-5 keys are equal.
"""
import random
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "host_pack_check.cpp"
EXE = HERE / "native" / "host_pack_check"
CODE = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3, ord("N"): 3}


@pytest.fixture(scope="module")
def packer():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    hdr = HERE.parent / "fastq-dupaway_amd" / "csrc" / "fqd_device.hpp"
    if not EXE.exists() or EXE.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-o", str(EXE), str(SRC)],
                       check=True, capture_output=True)

    def run(cases):
        text = "".join(f"{sh} {seq.hex()}\n" for sh, seq in cases)
        out = subprocess.run([str(EXE)], input=text, capture_output=True, text=True, check=True).stdout
        rows = []
        for line in out.splitlines():
            f = line.split()
            rows.append({"n": int(f[0]), "hash": int(f[1]), "bad_pos": int(f[2]), "bad_byte": int(f[3]),
                         "words": [int(x) for x in f[4:]]})
        return rows
    return run


def expected_words(seq: bytes):
    """Independent statement of the layout in fqd_device.hpp: per 64-base block
    [codes group 0][codes group 1 if any][N mask]; inside a 32-base group, base
    4k+j (dword k, byte j) -> codes bits 32*(k//4) + 8j + 2(k%4), mask bit 8j + k."""
    words = []
    for blk in range(0, len(seq), 64):
        part = seq[blk:blk + 64]
        mask = 0
        for gi, g in enumerate(range(0, len(part), 32)):
            w = 0
            for b, c in enumerate(part[g:g + 32]):
                k, j = divmod(b, 4)
                w |= CODE[c] << (32 * (k // 4) + 8 * j + 2 * (k % 4))
                if c == ord("N"):
                    mask |= 1 << (32 * gi + 8 * j + k)
            words.append(w)
        words.append(mask)
    return words


def test_words_match_independent_packing(packer):
    rnd = random.Random(1)
    cases = []
    for n in list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 149, 150, 151, 250, 1000]:
        for sh in range(4):
            cases.append((sh, bytes(rnd.choice(b"ACGTN" if n % 3 else b"ACGT") for _ in range(n))))
    rows = packer(cases)
    assert len(rows) == len(cases)
    for (sh, seq), r in zip(cases, rows):
        assert r["bad_pos"] == 0xFFFFFFFF
        assert r["words"] == expected_words(seq), (sh, seq)
        assert r["n"] == (len(seq) + 31) // 32 + (len(seq) + 63) // 64


def test_alignment_does_not_change_key_or_hash(packer):
    rnd = random.Random(2)
    seq = bytes(rnd.choice(b"ACGTN") for _ in range(150))
    rows = packer([(sh, seq) for sh in range(4)])
    assert all(r["words"] == rows[0]["words"] and r["hash"] == rows[0]["hash"] for r in rows)
    assert len(rows[0]["words"]) == 8          # 150 bp -> 64 B


def test_first_bad_byte_is_reported(packer):
    rows = packer([(0, b"ACGTxACGTy"), (1, b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTAC\r"), (2, b"n"), (3, b"ACG\x00"),
                   (0, b"ACGU"), (0, b"ACG "), (0, b"acgt")])
    assert (rows[0]["bad_pos"], rows[0]["bad_byte"]) == (4, ord("x"))
    assert (rows[1]["bad_pos"], rows[1]["bad_byte"]) == (38, ord("\r"))
    assert (rows[2]["bad_pos"], rows[2]["bad_byte"]) == (0, ord("n"))
    assert (rows[3]["bad_pos"], rows[3]["bad_byte"]) == (3, 0)
    assert (rows[4]["bad_pos"], rows[4]["bad_byte"]) == (3, ord("U"))
    assert (rows[5]["bad_pos"], rows[5]["bad_byte"]) == (3, ord(" "))
    assert (rows[6]["bad_pos"], rows[6]["bad_byte"]) == (0, ord("a"))


def test_every_byte_value_classified_like_the_reference(packer, oracle, capfd):
    # one sequence per byte value, at each of the 4 positions of a dword
    cases = [(0, b"A" * k + bytes([b]) + b"A" * (3 - k)) for b in range(256) for k in range(4)]
    rows = packer(cases)
    for (_, seq), r in zip(cases, rows):
        ref = oracle.pack_sequence(seq)
        if isinstance(ref, tuple):
            bad = next(i for i, c in enumerate(seq) if c not in b"ACGTN")
            assert (r["bad_pos"], r["bad_byte"]) == (bad, seq[bad]), seq
        else:
            assert r["bad_pos"] == 0xFFFFFFFF, seq
    capfd.readouterr()


def test_device_keys_equal_iff_oracle_keys_equal(packer, oracle):
    rnd = random.Random(3)
    seqs = [b"", b"A", b"C", b"AA", b"ACG", b"ACGA", b"AACG", b"ACGN", b"ACGG", b"N" * 32, b"G" * 32, b"N" * 33, b"G" * 33]
    base = bytes(rnd.choice(b"ACGT") for _ in range(150))
    seqs.append(base)
    for pos in (0, 31, 32, 63, 64, 127, 128, 149):          # single-base edits at word edges
        for c in b"ACGTN":
            s = bytearray(base); s[pos] = c; seqs.append(bytes(s))
    seqs += [base[:149], base + b"A", base[1:]]
    rows = packer([(rnd.randrange(4), s) for s in seqs])
    dev = [(len(s), tuple(r["words"])) for s, r in zip(seqs, rows)]
    ora = [(len(s), tuple(oracle.pack_sequence(s))) for s in seqs]
    for i in range(len(seqs)):
        for j in range(len(seqs)):
            assert (dev[i] == dev[j]) == (ora[i] == ora[j]) == (seqs[i] == seqs[j])
