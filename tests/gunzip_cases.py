"""Ordinary gzip files (one deflate stream per member, no BGZF fields) for the tests of the GPU gunzip reader: what gzip,
pigz and zlib write at several levels and strategies, with stored and fixed blocks in between, header fields, text that
packs a little and a lot."""
import gzip
import random
import struct
import zlib

from bgzf_cases import fasta_text, fastq_text


def member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem_level=8, header=b"", flush_every=0) -> bytes:
    """One gzip member around zlib's raw deflate of data; flush_every: a Z_FULL_FLUSH / Z_SYNC_FLUSH now and then (empty
    stored blocks in the stream, as pigz writes them)."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    body = b""
    if flush_every:
        for k, at in enumerate(range(0, len(data), flush_every)):
            body += co.compress(data[at:at + flush_every]) + co.flush(zlib.Z_FULL_FLUSH if k & 1 else zlib.Z_SYNC_FLUSH)
    else:
        body = co.compress(data)
    body += co.flush()
    head = header or b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03"
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def header_with_fields() -> bytes:
    """FEXTRA + FNAME + FCOMMENT + FHCRC."""
    h = bytearray(b"\x1f\x8b\x08\x1e\0\0\0\0\x00\x03")
    h += struct.pack("<H", 5) + b"ab\x01\x00z" + b"reads.fq\0" + b"a comment\0"
    h += struct.pack("<H", zlib.crc32(bytes(h)) & 0xFFFF)
    return bytes(h)


def cases():
    rnd = random.Random(11)
    fq = fastq_text(60000, 3)                                             # ~21 MB of FASTQ text with mixed qualities
    fa = fasta_text(3000, 5)
    noise = bytes(rnd.randrange(256) for _ in range(300_000))             # does not pack: stored blocks
    runs = (b"A" * 5000 + b"CGT" * 3000 + b"\n") * 60                       # packs hundredfold
    yield "fastq_level6", fq, member(fq, 6)
    yield "fastq_level1", fq, member(fq, 1)
    yield "fastq_level9", fq, member(fq, 9)
    yield "fastq_python_gzip", fq[:5_000_000], gzip.compress(fq[:5_000_000], 6)
    yield "fastq_header_fields", fq[:3_000_000], member(fq[:3_000_000], 6, header=header_with_fields())
    yield "fastq_with_flushes", fq[:8_000_000], member(fq[:8_000_000], 6, flush_every=700_001)
    yield "fastq_fixed_blocks", fq[:2_000_000], member(fq[:2_000_000], 6, zlib.Z_FIXED)
    yield "fastq_huffman_only", fq[:4_000_000], member(fq[:4_000_000], 6, zlib.Z_HUFFMAN_ONLY)
    yield "fastq_rle", fq[:4_000_000], member(fq[:4_000_000], 6, zlib.Z_RLE)
    yield "fastq_small_blocks", fq[:6_000_000], member(fq[:6_000_000], 6, mem_level=1)
    yield "fasta_level6", fa, member(fa, 6)
    yield "noise_then_text", noise + fq[:2_000_000] + noise, member(noise + fq[:2_000_000] + noise, 6)
    yield "long_runs", runs, member(runs, 6)
    yield "tiny", b"@r\nACGT\n+\nIIII\n", member(b"@r\nACGT\n+\nIIII\n", 6)
    yield "empty", b"", member(b"", 6)
