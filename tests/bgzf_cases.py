"""Inputs shared by the CPU and GPU tests of the device BGZF coder."""
import numpy as np


def fastq_text(n, seed=0, read_len=150, style="illumina"):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = read_len if style != "ragged" else int(rng.integers(1, read_len + 1))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L, p=[.245, .245, .245, .245, .02]).tobytes()
        qual = rng.choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8), size=L).tobytes()
        if style == "short":
            head = b"@r%d" % i
        else:
            head = b"@A00123:45:HXXXXXXX:%d:%d:%d:%d 1:N:0:ACGTACGT" % (1 + i % 4, 1101 + i // 5000, int(rng.integers(1000, 33000)), int(rng.integers(1000, 40000)))
        out.append(head + b"\n" + seq + b"\n+\n" + qual + b"\n")
    return b"".join(out)


def fasta_text(n, seed=0):
    rng = np.random.default_rng(seed)
    return b"".join(b">contig_%d len=100\n" % i + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=100).tobytes() + b"\n" for i in range(n))


def cases():
    rng = np.random.default_rng(7)
    big = fastq_text(3000, 1)
    yield "empty", b"", 4
    yield "one_byte", b"@", 4
    yield "fastq_small", fastq_text(10, 2), 4
    yield "fastq_3000", big, 4
    yield "fastq_ragged", fastq_text(2500, 3, style="ragged"), 4
    yield "fastq_short_ids", fastq_text(2500, 4, read_len=36, style="short"), 4       # > 4096 lines per member
    yield "fasta", fasta_text(4000, 5), 2
    yield "member_minus_1", big[:65279], 4
    yield "member_exact", big[:65280], 4
    yield "member_plus_1", big[:65281], 4
    yield "two_members_exact", big[:2 * 65280], 4
    yield "random_bytes", rng.integers(0, 256, size=200_001, dtype=np.uint8).tobytes(), 4      # incompressible: stored
    yield "one_symbol", b"I" * 150_000, 4
    yield "only_newlines", b"\n" * 100_000, 4
    yield "two_symbols", rng.choice(np.frombuffer(b"ab", dtype=np.uint8), size=70_000).tobytes(), 2
    yield "binary_with_nul", bytes(range(256)) * 300 + b"\0" * 5000, 4
    yield "long_lines", (b"x" * 40_000 + b"\n") * 5, 4                                        # column candidate beyond 32768
    yield "skewed", rng.choice(np.arange(256, dtype=np.uint8), size=300_000, p=(lambda w: w / w.sum())(1.0 / np.arange(1, 257) ** 3)).tobytes(), 4
