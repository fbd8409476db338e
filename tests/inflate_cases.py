"""BGZF files for the tests of the GPU reader: members made by zlib under every strategy and level, so that
stored, fixed-Huffman and dynamic blocks, several blocks per member and all match shapes occur."""
import struct
import zlib

import numpy as np

from bgzf_cases import fasta_text, fastq_text

EOF_MARK = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def member(raw: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, flush_every=0) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    if flush_every:                                         # several blocks per member (empty stored blocks between them)
        body = b"".join(c.compress(raw[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(raw), flush_every)) + c.flush()
    else:
        body = c.compress(raw) + c.flush()
    total = 18 + len(body) + 8
    assert total <= 65536
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", total - 1) + body +
            struct.pack("<II", zlib.crc32(raw), len(raw)))


def bgzf(data: bytes, size=65280, **kw) -> bytes:
    return b"".join(member(data[i:i + size], **kw) for i in range(0, len(data), size)) + EOF_MARK


def cases():
    rng = np.random.default_rng(9)
    fq = fastq_text(2500, 31)
    yield "level6", fq, bgzf(fq)
    yield "level1", fq, bgzf(fq, level=1)
    yield "level9_small_window_blocks", fq, bgzf(fq, level=9, mem_level=1)           # many blocks per member
    yield "stored", fq[:200_000], bgzf(fq[:200_000], level=0)
    yield "fixed_huffman", fq, bgzf(fq, strategy=zlib.Z_FIXED)
    yield "huffman_only", fq, bgzf(fq, strategy=zlib.Z_HUFFMAN_ONLY)
    yield "rle", fq, bgzf(fq, strategy=zlib.Z_RLE)
    yield "flushes_inside_members", fq, bgzf(fq, flush_every=5000)
    yield "members_of_64k", fq, bgzf(fq, size=65536, level=9)
    yield "tiny_members", fq[:30_000], bgzf(fq[:30_000], size=97)
    yield "fasta", (fa := fasta_text(3000, 8)), bgzf(fa)
    rnd = rng.integers(0, 256, size=150_000, dtype=np.uint8).tobytes()
    yield "random_bytes", rnd, bgzf(rnd, size=60000)
    yield "long_runs", b"A" * 100_000 + b"CG" * 50_000, bgzf(b"A" * 100_000 + b"CG" * 50_000)
    periods = b"".join(bytes(range(65, 65 + k)) * (n // k) + b"\n" for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 31, 32, 33) for n in (3, 7, 8, 9, 15, 16, 17, 40, 64, 150, 258, 259, 1000))
    yield "short_periods", periods, bgzf(periods)
    yield "short_periods_level1", periods, bgzf(periods, level=1)
    import random
    rnd3 = random.Random(3)
    words = [bytes([97 + i, 97 + j, 97 + k]) for i in range(3) for j in range(3) for k in range(3)][:8]
    dense = b"".join(rnd3.choice(words) for _ in range(120_000))            # ten thousand three-byte matches per member:
    yield "more_matches_than_a_window_holds", dense, bgzf(dense, level=1)    # windows of the wave decoder are cut short
    yield "one_byte", b"x", bgzf(b"x")
    yield "empty_members_between", fq[:70_000], member(fq[:30_000]) + member(b"") + member(fq[30_000:70_000]) + EOF_MARK


def random_cases(seed: int, trials: int):
    """(data, BGZF bytes) pairs: texts of every texture under every zlib level, strategy and memory level, with flushes
    in odd places, in members of odd sizes."""
    import random
    rnd = random.Random(seed)

    def text(n):
        out = bytearray()
        lines = []
        while len(out) < n:
            kind = rnd.randrange(6)
            if kind == 0: piece = bytes([rnd.randrange(256)]) * rnd.randrange(1, 400)
            elif kind == 1: piece = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9))) * rnd.randrange(1, 60)
            elif kind == 2 and lines: piece = rnd.choice(lines)
            elif kind == 3: piece = bytes(rnd.choice(b"ACGT") for _ in range(rnd.randrange(1, 200))) + b"\n"
            elif kind == 4: piece = bytes(rnd.choice(b"FFFFFFF:,#") for _ in range(rnd.randrange(1, 200))) + b"\n"
            else: piece = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 50)))
            lines.append(piece)
            out += piece
        return bytes(out[:n])

    for _ in range(trials):
        data = text(rnd.randrange(1, 200_000))
        raw, at = b"", 0
        while at < len(data):
            size = rnd.choice([97, 1000, 20_000, 65_280, 65_536])
            kw = dict(level=rnd.randrange(0, 10), strategy=rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]),
                      mem_level=rnd.randrange(1, 10), flush_every=rnd.choice([0, 0, 777, 5000]))
            chunk = data[at:at + size]
            try:
                raw += member(chunk, **kw)
            except AssertionError:                               # (did not fit a BGZF member: noise at level 0)
                raw += member(chunk[: len(chunk) // 2], level=1) + member(chunk[len(chunk) // 2:], level=1)
            at += size
        yield data, raw + EOF_MARK
