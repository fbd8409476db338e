"""ctypes bindings for the CPU checker (oracle/liboracle.so, oracle/_ref/libfqd_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and
by __graft_entry__.smoke().  The shipped engine never imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")

FASTQ, FASTA = 0, 1

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)


def build(ref: bool = True) -> None:
    """Compile the restatement, and the reference objects when /root/reference exists."""
    subprocess.run(["make", "-s", "-C", str(HERE), "all"], check=True)
    if ref and (REFERENCE / "src" / "seq_utils.cpp").exists():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.fqo_pack_sequence.restype = C.c_int64
        L.fqo_pack_sequence.argtypes = [C.c_char_p, C.c_int64, _u64p, C.c_int64]
        L.fqo_parse_block.restype = C.c_int64
        L.fqo_parse_block.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, _i64p, C.c_int64,
                                      _i64p, C.c_char_p, C.c_int64]
        L.fqo_compare_tags.restype = C.c_int
        L.fqo_compare_tags.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]
        L.fqo_dedup_single.restype = C.c_int64
        L.fqo_dedup_single.argtypes = [_u8p, _u64p, _u32p, C.c_uint64, _u8p, _u64p]
        L.fqo_dedup_paired.restype = C.c_int64
        L.fqo_dedup_paired.argtypes = [_u8p, _u64p, _u32p, _u8p, _u64p, _u32p, C.c_uint64, _u8p, _u64p]
        L.fqo_join_tags.restype = C.c_int64
        L.fqo_join_tags.argtypes = [_u8p, _u64p, _u32p, C.c_uint64, _u8p, _u64p, _u32p, C.c_uint64,
                                    C.c_int, _u64p, _u64p, _u64p]
        L.fqo_filter_single.restype = C.c_int
        L.fqo_filter_single.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int64,
                                        _u64p, _u64p, C.c_char_p, C.c_int64]
        L.fqo_filter_paired.restype = C.c_int
        L.fqo_filter_paired.argtypes = [C.c_char_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                                           _u64p, _u64p, _u64p, C.c_char_p, C.c_int64]

    # -- a1-a3 ---------------------------------------------------------------
    def pack_sequence(self, seq: bytes):
        """-> list of base-5 chunk words, or ('bad', byte) on an unknown base."""
        cap = len(seq) // 17 + 2
        out = (C.c_uint64 * cap)()
        n = self.lib.fqo_pack_sequence(seq, len(seq), out, cap)
        if n < 0:
            return ("bad", -n - 1)
        return [int(out[i]) for i in range(n)]

    # -- a13/a14 ---------------------------------------------------------------
    def parse_block(self, buf: bytes, fmt: int, want_tag: bool = False):
        cap = buf.count(b"\n") + 1
        out = np.zeros((cap, 7), dtype=np.int64)
        consumed = C.c_int64(0)
        err = C.create_string_buffer(256)
        n = self.lib.fqo_parse_block(buf, len(buf), fmt, int(want_tag), _ptr(out, _i64p), cap,
                                     C.byref(consumed), err, 256)
        if n < 0:
            raise ValueError(err.value.decode())
        return out[:n].copy(), consumed.value

    def compare_tags(self, a: bytes, b: bytes) -> int:
        c = self.lib.fqo_compare_tags(a, len(a), b, len(b))
        return (c > 0) - (c < 0)

    # -- array-level dedup -----------------------------------------------------
    def dedup_single(self, data: np.ndarray, off: np.ndarray, length: np.ndarray):
        """keep flags (uint8) for packed ASCII reads; raises ValueError(byte, index) on a bad base."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        n = len(off)
        keep = np.zeros(n, dtype=np.uint8)
        bad = C.c_uint64(0)
        r = self.lib.fqo_dedup_single(_ptr(data, _u8p), _ptr(off, _u64p), _ptr(length, _u32p), n,
                                      _ptr(keep, _u8p), C.byref(bad))
        if r < 0:
            raise ValueError(-r - 1, bad.value)
        return keep

    def dedup_paired(self, d1, off1, len1, d2, off2, len2):
        d1 = np.ascontiguousarray(d1, dtype=np.uint8); d2 = np.ascontiguousarray(d2, dtype=np.uint8)
        off1 = np.ascontiguousarray(off1, dtype=np.uint64); off2 = np.ascontiguousarray(off2, dtype=np.uint64)
        len1 = np.ascontiguousarray(len1, dtype=np.uint32); len2 = np.ascontiguousarray(len2, dtype=np.uint32)
        n = len(off1)
        keep = np.zeros(n, dtype=np.uint8)
        bad = C.c_uint64(0)
        r = self.lib.fqo_dedup_paired(_ptr(d1, _u8p), _ptr(off1, _u64p), _ptr(len1, _u32p),
                                      _ptr(d2, _u8p), _ptr(off2, _u64p), _ptr(len2, _u32p), n,
                                      _ptr(keep, _u8p), C.byref(bad))
        if r < 0:
            raise ValueError(-r - 1, bad.value)
        return keep

    def join_tags(self, t1, off1, len1, t2, off2, len2, tail_rule: bool):
        t1 = np.ascontiguousarray(t1, dtype=np.uint8); t2 = np.ascontiguousarray(t2, dtype=np.uint8)
        off1 = np.ascontiguousarray(off1, dtype=np.uint64); off2 = np.ascontiguousarray(off2, dtype=np.uint64)
        len1 = np.ascontiguousarray(len1, dtype=np.uint32); len2 = np.ascontiguousarray(len2, dtype=np.uint32)
        cap = max(1, min(len(off1), len(off2)))
        i1 = np.zeros(cap, dtype=np.uint64); i2 = np.zeros(cap, dtype=np.uint64)
        un = C.c_uint64(0)
        n = self.lib.fqo_join_tags(_ptr(t1, _u8p), _ptr(off1, _u64p), _ptr(len1, _u32p), len(off1),
                                   _ptr(t2, _u8p), _ptr(off2, _u64p), _ptr(len2, _u32p), len(off2),
                                   int(tail_rule), _ptr(i1, _u64p), _ptr(i2, _u64p), C.byref(un))
        return i1[:n].copy(), i2[:n].copy(), un.value

    # -- file drivers ------------------------------------------------------------
    def filter_single(self, src, dst, fmt=FASTQ, verbose=False, block_bytes=0):
        tot, dup = C.c_uint64(0), C.c_uint64(0)
        err = C.create_string_buffer(512)
        rc = self.lib.fqo_filter_single(str(src).encode(), str(dst).encode(), fmt, int(verbose), block_bytes,
                                        C.byref(tot), C.byref(dup), err, 512)
        if rc:
            raise RuntimeError(err.value.decode())
        return tot.value, dup.value

    def filter_paired(self, in1, in2, out1, out2, fmt=FASTQ, unordered=False, tail_rule=True,
                      verbose=False, block_bytes=0):
        tot, dup, un = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        err = C.create_string_buffer(512)
        rc = self.lib.fqo_filter_paired(str(in1).encode(), str(in2).encode(), str(out1).encode(),
                                        str(out2).encode(), fmt, int(unordered), int(tail_rule), int(verbose),
                                        block_bytes, C.byref(tot), C.byref(dup), C.byref(un), err, 512)
        if rc:
            raise RuntimeError(err.value.decode())
        return tot.value, dup.value, un.value


class RefLib:
    """The reference's own seq_utils / fastqview / fastaview object code (oracle/_ref)."""

    def __init__(self, lib):
        self.lib = lib
        lib.ref_seq2hash.restype = C.c_int64
        lib.ref_seq2hash.argtypes = [C.c_char_p, C.c_int64, _u64p, C.c_int64]
        lib.ref_pattern2number.restype = C.c_uint64
        lib.ref_pattern2number.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int)]
        for f in (lib.ref_walk_fastq, lib.ref_walk_fasta):
            f.restype = C.c_int64
            f.argtypes = [C.c_char_p, C.c_int64, _i64p, C.c_int64, _i64p]
        for f in (lib.ref_cmp_fastq_ids, lib.ref_cmp_fasta_ids):
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]

    def seq2hash(self, seq: bytes):
        cap = len(seq) // 17 + 2
        out = (C.c_uint64 * cap)()
        n = self.lib.ref_seq2hash(seq, len(seq), out, cap)
        if n < 0:
            return None
        return [int(out[i]) for i in range(n)]

    def walk(self, buf: bytes, fmt: int):
        cap = buf.count(b"\n") + 1
        out = np.zeros((cap, 5), dtype=np.int64)
        consumed = C.c_int64(0)
        b = C.create_string_buffer(buf, len(buf))
        f = self.lib.ref_walk_fastq if fmt == FASTQ else self.lib.ref_walk_fasta
        n = f(b, len(buf), _ptr(out, _i64p), cap, C.byref(consumed))
        if n < 0:
            return None, 0
        return out[:n].copy(), consumed.value

    def cmp_ids(self, rec_a: bytes, rec_b: bytes, fmt: int) -> int:
        a = C.create_string_buffer(rec_a, len(rec_a))
        b = C.create_string_buffer(rec_b, len(rec_b))
        f = self.lib.ref_cmp_fastq_ids if fmt == FASTQ else self.lib.ref_cmp_fasta_ids
        return f(a, len(rec_a), b, len(rec_b))


def load_oracle() -> Oracle:
    so = HERE / "liboracle.so"
    src_m = max((HERE / "fqd_oracle.cpp").stat().st_mtime, (HERE / "fqd_oracle.hpp").stat().st_mtime)
    if not so.exists() or so.stat().st_mtime < src_m:
        build(ref=False)
    return Oracle(C.CDLL(str(so)))


def load_ref():
    so = HERE / "_ref" / "libfqd_ref.so"
    if not so.exists():
        if (REFERENCE / "src" / "seq_utils.cpp").exists():
            build(ref=True)
        else:
            return None
    return RefLib(C.CDLL(str(so)))
