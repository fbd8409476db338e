#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.json from the REFERENCE'S OWN object code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
It builds oracle/_ref/libfqd_ref.so (the reference's seq_utils.cpp, fastqview.cpp,
fastaview.cpp compiled where they lie) and records its outputs on the inputs
below.  The JSON holds inputs and expected outputs only — data, no reference text.
"""
import json
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.binding import FASTA, FASTQ, build, load_ref  # noqa: E402


def seqs():
    rnd = random.Random(20261003)
    out = [b"", b"A", b"C", b"G", b"T", b"N", b"ACG", b"ACGA", b"AACG", b"ACGN", b"NNNN"]
    for n in (16, 17, 18, 33, 34, 35, 51, 100, 149, 150, 151, 250, 301):
        out.append(bytes(rnd.choice(b"ACGT") for _ in range(n)))
        out.append(bytes(rnd.choice(b"ACGTN") for _ in range(n)))
        out.append(b"T" * n)          # largest digits: 5^17-1 territory for T..., N below
        out.append(b"N" * n)
        out.append(b"A" * n)
    # unknown bases: lowercase, CR, IUPAC, NUL-adjacent, high bit
    for bad in (b"acgt", b"ACGTa", b"ACGT\r", b"ACGR", b"AC-GT", b"ACG\xc1", b"ACGT" * 10 + b"n", b"U", b" ACGT"):
        out.append(bad)
    return out


FASTQ_BLOCKS = [
    b"@r1\nACGT\n+\nIIII\n@r2 desc\nNNAC\n+r2\n!!!!\n",
    b"@r1\nACGT\n+\nIIII\n@r2\nAC",                       # partial record at block end
    b"@r1\n\n+\n\n@r2\nA\n+\nI\n",                          # empty sequence line
    b"@r1\nACGT\n+\nIIII",                                  # no trailing newline: dropped
    b"",                                                    # empty block
    b"@only id\n",
    b"r1\nACGT\n+\nIIII\n",                                 # bad start char -> throws
    b"@r1\nACGT\n+\nIII\n",                                 # qual shorter -> throws
    b"@r1\nACGT\n+\nIIII\n>r2\nAC\n+\nII\n",                # second record bad start -> throws
    b"@r1\r\nACGT\r\n+\r\nIIII\r\n",                        # CRLF parses (bases fail later)
    b"@a\nACGT\nXYZ\nIIII\n",                               # third line is not checked
]
FASTA_BLOCKS = [
    b">0001\nATGCTAGCTA\n>0002\nCGTACGTAGC\n",
    b">0001\nATGC\n>0002",
    b">x\n\n>y\nN\n",
    b"@r\nACGT\n",                                          # bad start -> throws
    b">a\nAC\nGT\n",                                        # multi-line: 2nd record bad start -> throws
    b"",
    b">a\nACGT",
]

ID_LINES = [
    b"r000000001", b"r000000002", b"r000000010",
    b"SRR1.9 x", b"SRR1.10 y", b"SRR1.10 z", b"SRR1.100",
    b"A00123:45:HXX:1:1101:1234:5678 1:N:0:ACGT", b"A00123:45:HXX:1:1101:1234:5678 2:N:0:ACGT",
    b"A00123:45:HXX:1:1101:1234:5679 1:N:0:ACGT",
    b"read/1", b"read/2", b"read", b"read ", b"read comment.with.dot", b"x.y.z w", b".", b". ", b"",
    b"0001", b"0002", b"0010", b"001", b"00010",
]


def main():
    build(ref=True)
    ref = load_ref()
    assert ref is not None, "needs /root/reference"
    vec = {"seq2hash": [], "walk": [], "cmp_ids": []}
    for s in seqs():
        vec["seq2hash"].append({"seq": s.decode("latin-1"), "words": ref.seq2hash(s)})
    for fmt, blocks in ((FASTQ, FASTQ_BLOCKS), (FASTA, FASTA_BLOCKS)):
        for b in blocks:
            rows, consumed = ref.walk(b, fmt)
            vec["walk"].append({"fmt": fmt, "block": b.decode("latin-1"),
                                "records": None if rows is None else rows.tolist(),
                                "consumed": consumed})
    for fmt, lead, tail in ((FASTQ, b"@", b"\nAC\n+\nII\n"), (FASTA, b">", b"\nAC\n")):
        for a in ID_LINES:
            for b in ID_LINES:
                ra, rb = lead + a + tail, lead + b + tail
                vec["cmp_ids"].append({"fmt": fmt, "a": ra.decode("latin-1"), "b": rb.decode("latin-1"),
                                       "sign": ref.cmp_ids(ra, rb, fmt)})
    out = Path(__file__).with_name("ref_vectors.json")
    out.write_text(json.dumps(vec, indent=0, separators=(",", ":")))
    print(f"wrote {out}: {len(vec['seq2hash'])} seq2hash, {len(vec['walk'])} walk, {len(vec['cmp_ids'])} cmp_ids")


if __name__ == "__main__":
    main()
