"""ctypes wrapper over the C ABI (include/fqdupaway.h).  Plumbing only: all work
happens in lib/libfqdupaway.so on the GPU; nothing here computes a result."""
import ctypes as C
from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import FqdError, load_library


@dataclass
class Reads:
    """One mate's sequences for a batch (mirrors struct fqd_reads).

    bases/offsets/lengths are numpy arrays (host space), torch CUDA tensors or raw
    device pointers as int (device space).  Leave offsets and lengths None for a
    uniform batch: read i = uniform_len bytes at bases + i*uniform_stride."""
    bases: Any
    offsets: Any = None
    lengths: Any = None
    uniform_len: int = 0
    uniform_stride: int = 0


def _is_host(x) -> bool:
    return isinstance(x, np.ndarray)


def _addr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, int):
        return x
    return x.data_ptr()          # torch tensor


def _torch_stream(x) -> int:
    """hipStream_t (as an integer) of torch's current stream on the tensor's device."""
    import torch
    return int(torch.cuda.current_stream(x.device).cuda_stream)


class Engine:
    """One HBM-resident exact sequence set on one MI355X (struct fqd_engine)."""

    def __init__(self, segments: int = 1, device: int = 0, capacity_reads: int = 0, capacity_bases: int = 0,
                 stream: Optional[int] = None, profile: bool = False, no_stage: bool = False, weak_hash: bool = False):
        self._L = load_library()
        cfg = _lib.Config(device=device, segments=segments, capacity_reads=capacity_reads,
                          capacity_bases=capacity_bases, stream=stream,
                          flags=(_lib.FLAG_PROFILE if profile else 0) | (_lib.FLAG_NO_STAGE if no_stage else 0)
                          | (_lib.FLAG_WEAK_HASH if weak_hash else 0))
        h = C.c_void_p()
        rc = self._L.fqd_engine_create(C.byref(cfg), C.byref(h))
        if rc != _lib.OK:
            raise FqdError(rc, (self._L.fqd_last_error(None) or b"").decode())
        self._h = h
        self._ordered = False
        self.segments = segments

    # -- lifecycle ----------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.fqd_engine_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ordering against the caller's stream ------------------------------------------
    # The engine runs on its own non-blocking stream and waits for nobody (include/fqdupaway.h, ORDERING RULE): whatever
    # torch queued for a tensor on ITS stream (zeros, fill_, copy_) must be ordered before the engine's work explicitly.
    # The first torch tensor among a call's arguments makes the engine wait for torch's current stream; arguments are
    # evaluated before the C function runs, so the wait is queued ahead of the call's kernels.
    def _p(self, x) -> Optional[int]:
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            return x.ctypes.data
        if isinstance(x, int):
            return x
        if not self._ordered:
            self._ordered = True
            rc = self._L.fqd_engine_wait_stream(self._h, _torch_stream(x))
            if rc != _lib.OK:
                self._ordered = False
                raise FqdError(rc, (self._L.fqd_last_error(self._h) or b"").decode())
        return x.data_ptr()

    def _desc(self, segs: Sequence[Reads]):
        arr = (_lib.ReadsDesc * 2)()
        for k, s in enumerate(segs):
            arr[k].bases = self._p(s.bases)
            arr[k].offsets = self._p(s.offsets)
            arr[k].lengths = self._p(s.lengths)
            arr[k].uniform_len = s.uniform_len
            arr[k].uniform_stride = s.uniform_stride
        return arr

    def wait_stream(self, stream: Optional[int] = None):
        """The engine starts after everything queued so far on `stream` (default: torch's current stream)."""
        if stream is None:
            import torch
            stream = int(torch.cuda.current_stream().cuda_stream)
        self._check(self._L.fqd_engine_wait_stream(self._h, stream))

    def release_to(self, stream: Optional[int] = None):
        """`stream` (default: torch's current stream) continues after everything the engine has queued so far."""
        if stream is None:
            import torch
            stream = int(torch.cuda.current_stream().cuda_stream)
        self._check(self._L.fqd_stream_wait_engine(self._h, stream))

    def _check(self, rc: int):
        self._ordered = False
        if rc != _lib.OK:
            raise FqdError(rc, (self._L.fqd_last_error(self._h) or b"").decode())

    def reset(self):
        self._check(self._L.fqd_engine_reset(self._h))

    def sync(self):
        self._check(self._L.fqd_engine_sync(self._h))

    def stream_handle(self) -> int:
        """The engine's hipStream_t as an integer (wrap with torch.cuda.ExternalStream)."""
        return int(self._L.fqd_engine_stream(self._h) or 0)

    # -- the hot path ---------------------------------------------------------------
    def submit(self, segs: Sequence[Reads], n: int, keep=None, final: bool = False):
        """Dedups n more records; returns their keep flags (numpy for host input,
        the given device buffer otherwise).  final: this is the run's last batch (fqd_submit_final)."""
        if len(segs) != self.segments:
            raise ValueError(f"engine has {self.segments} mate(s) per record, got {len(segs)}")
        host = _is_host(segs[0].bases)
        for s in segs:
            if _is_host(s.bases) != host:
                raise ValueError("all mates must live in the same memory space")
            if _is_host(s.bases):
                for a, dt in ((s.bases, np.uint8), (s.offsets, np.uint64), (s.lengths, np.uint32)):
                    if a is not None and (a.dtype != dt or not a.flags["C_CONTIGUOUS"]):
                        raise ValueError(f"host arrays must be C-contiguous {dt}")
        if host:
            keep = np.empty(n, dtype=np.uint8) if keep is None else keep
        elif keep is None:
            raise ValueError("device submits need a device keep buffer")
        fn = self._L.fqd_submit_final if final else self._L.fqd_submit
        rc = fn(self._h, self._desc(segs), n, _lib.MEM_HOST if host else _lib.MEM_DEVICE, self._p(keep))
        self._check(rc)
        return keep

    def bad_base(self):
        rec, seg, pos, byte = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint8()
        self._check(self._L.fqd_bad_base(self._h, C.byref(rec), C.byref(seg), C.byref(pos), C.byref(byte)))
        return rec.value, seg.value, pos.value, byte.value

    def stats(self):
        st = _lib.Stats()
        self._check(self._L.fqd_get_stats(self._h, C.byref(st)))
        return {"records": st.records, "duplicates": st.duplicates, "table_slots": st.table_slots, "key_bytes": st.key_bytes}

    def profile(self):
        p = _lib.Profile()
        self._check(self._L.fqd_get_profile(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in _lib.Profile._fields_}

    def reset_profile(self):
        self._check(self._L.fqd_reset_profile(self._h))

    # -- sharding halves ---------------------------------------------------------------
    def key_words(self, len0: int, len1: int = 0) -> int:
        return int(self._L.fqd_key_words(len0, len1))

    def encode_uniform(self, segs: Sequence[Reads], n: int, records):
        self._check(self._L.fqd_encode_uniform(self._h, self._desc(segs), n, self._p(records)))

    def partition_keys(self, records, n: int, key_words: int, n_parts: int, out_keys, counts, origin):
        self._check(self._L.fqd_partition_keys(self._h, self._p(records), n, key_words, n_parts,
                                               self._p(out_keys), self._p(counts), self._p(origin)))

    def reserve_keys(self, n: int, len0: int, len1: int) -> int:
        """Device address of room for n keys at the tail of the key store (receive in place)."""
        slot = C.c_void_p()
        self._check(self._L.fqd_reserve_keys(self._h, n, len0, len1, C.byref(slot)))
        return slot.value

    def insert_keys(self, keys, n: int, len0: int, len1: int, keep):
        self._check(self._L.fqd_insert_keys(self._h, self._p(keys), n, len0, len1, self._p(keep)))

    def padded_key_words(self, max_len0: int, max_len1: int = 0) -> int:
        return int(self._L.fqd_padded_key_words(max_len0, max_len1))

    def encode_padded(self, segs: Sequence[Reads], n: int, max_len0: int, max_len1: int, records):
        self._check(self._L.fqd_encode_padded(self._h, self._desc(segs), n, max_len0, max_len1, self._p(records)))

    def widen_keys(self, new_words: int):
        self._check(self._L.fqd_widen_keys(self._h, new_words))

    def partition_slabs(self, records, n: int, key_words: int, n_parts: int, slab_cap: int, out_keys, counts, origin):
        self._check(self._L.fqd_partition_slabs(self._h, self._p(records), n, key_words, n_parts, slab_cap,
                                                self._p(out_keys), self._p(counts), self._p(origin)))

    def encode_slabs(self, segs: Sequence[Reads], n: int, n_parts: int, chunk_reads: int, n_chunks: int, sub_cap: int, out_keys,
                     chunk_counts, totals, origin, exact: bool = False):
        self._check(self._L.fqd_encode_slabs(self._h, self._desc(segs), n, n_parts, chunk_reads, n_chunks, sub_cap, self._p(out_keys),
                                             self._p(chunk_counts), self._p(totals), self._p(origin), 1 if exact else 0))

    def encode_slabs_hashed(self, segs: Sequence[Reads], n: int, n_parts: int, chunk_reads: int, n_chunks: int, sub_cap: int, out_keys, out_hashes,
                            chunk_counts, totals, origin, exact: bool = False):
        self._check(self._L.fqd_encode_slabs_hashed(self._h, self._desc(segs), n, n_parts, chunk_reads, n_chunks, sub_cap, self._p(out_keys), self._p(out_hashes),
                                                    self._p(chunk_counts), self._p(totals), self._p(origin), 1 if exact else 0))

    def insert_slabs_hashed(self, keys, hashes, n_slabs: int, slab_cap: int, slab_count, len0: int, len1: int, keep):
        self._check(self._L.fqd_insert_slabs_hashed(self._h, self._p(keys), self._p(hashes), n_slabs, slab_cap, self._p(slab_count), len0, len1, self._p(keep)))

    def insert_slabs(self, keys, n_slabs: int, slab_cap: int, slab_count, len0: int, len1: int, keep):
        self._check(self._L.fqd_insert_slabs(self._h, self._p(keys), n_slabs, slab_cap, self._p(slab_count), len0, len1, self._p(keep)))

    # -- --unordered ID join ---------------------------------------------------------------
    def _tags(self, bytes_, offsets, lengths, n):
        return _lib.TagsDesc(bytes=self._p(bytes_), offsets=self._p(offsets), lengths=self._p(lengths), n=n)

    def sort_tags(self, bytes_, offsets, lengths, n: int, perm):
        t = self._tags(bytes_, offsets, lengths, n)
        self._check(self._L.fqd_sort_tags(self._h, C.byref(t), self._p(perm)))

    def extract_tags(self, text, id_start, id_len, n: int, tag_off, tag_len):
        self._check(self._L.fqd_extract_tags(self._h, self._p(text), self._p(id_start), self._p(id_len), n, self._p(tag_off), self._p(tag_len)))

    def join_tags(self, a, b, perm_a, perm_b, match_a, match_b, pair_a, pair_b) -> int:
        """a, b = (bytes, offsets, lengths, n); returns the number of pairs (the call drains the stream)."""
        ta, tb = self._tags(*a), self._tags(*b)
        n_pairs = C.c_uint64(0)
        out = _lib.JoinDesc(perm_a=self._p(perm_a), perm_b=self._p(perm_b), match_a=self._p(match_a), match_b=self._p(match_b),
                            pair_a=self._p(pair_a), pair_b=self._p(pair_b), n_pairs=C.pointer(n_pairs))
        self._check(self._L.fqd_join_tags(self._h, C.byref(ta), C.byref(tb), C.byref(out)))
        return int(n_pairs.value)

    def gather_seqs(self, idx, n: int, off_table, len_table, off_out, len_out):
        self._check(self._L.fqd_gather_seqs(self._h, self._p(idx), n, self._p(off_table), self._p(len_table), self._p(off_out), self._p(len_out)))

    def copy_spans(self, src, src_off, lens, n: int, dst, dst_off):
        self._check(self._L.fqd_copy_spans(self._h, self._p(src), self._p(src_off), self._p(lens), n, self._p(dst), self._p(dst_off)))

    def bgzf_bound(self, n: int) -> int:
        return int(self._L.fqd_bgzf_bound(n))

    def bgzf_deflate(self, src, n: int, dst, lines_per_record: int = 4) -> int:
        """BGZF members for the n bytes at src (device) written to dst (device, >= bgzf_bound(n) bytes);
        returns their total size.  The end-of-file marker is the caller's to append."""
        total = C.c_uint64(0)
        self._check(self._L.fqd_bgzf_deflate(self._h, self._p(src), n, lines_per_record, self._p(dst), dst.numel(), C.byref(total)))
        return int(total.value)

    def bgzf_inflate(self, comp, comp_off, comp_len, out_off, out_len, crc, n_members: int, text) -> int:
        """Members (device arrays describing them) of the BGZF bytes at comp inflated into text; returns the number
        of members that failed (damaged stream, wrong size or CRC)."""
        bad = C.c_uint64(0)
        self._check(self._L.fqd_bgzf_inflate(self._h, self._p(comp), self._p(comp_off), self._p(comp_len), self._p(out_off), self._p(out_len),
                                             self._p(crc), n_members, self._p(text), C.byref(bad)))
        return int(bad.value)

    def gunzip(self, deflate, avail: int, text, arrived=None):
        """An ordinary gzip member's deflate stream (device bytes from its first byte on) inflated into text (device); returns
        (ok, text_bytes, deflate_bytes, crc32).  arrived: a ctypes.c_uint64 another thread raises while it copies the file into
        `deflate` (fqd_gunzip_arriving: the call works on what is there and waits for the rest)."""
        tb, db, crc, ok = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0), C.c_int32(0)
        if arrived is None:
            self._check(self._L.fqd_gunzip(self._h, self._p(deflate), avail, self._p(text), text.numel(), C.byref(tb), C.byref(db), C.byref(crc), C.byref(ok)))
        else:
            self._check(self._L.fqd_gunzip_arriving(self._h, self._p(deflate), avail, C.byref(arrived), self._p(text), text.numel(),
                                                    C.byref(tb), C.byref(db), C.byref(crc), C.byref(ok)))
        return bool(ok.value), int(tb.value), int(db.value), int(crc.value)

    def count_lines(self, text, n: int) -> int:
        lines = C.c_uint64(0)
        self._check(self._L.fqd_count_lines(self._h, self._p(text), n, C.byref(lines)))
        return int(lines.value)

    def scan_records(self, text, n: int, lines_per_record: int, n_records: int, start, seq_off, id_len, seq_len, size) -> bool:
        ok = C.c_int(0)
        self._check(self._L.fqd_scan_records(self._h, self._p(text), n, lines_per_record, n_records, self._p(start), self._p(seq_off),
                                             self._p(id_len), self._p(seq_len), self._p(size), C.byref(ok)))
        return bool(ok.value)

    def count_tags_le(self, t, other, other_index: int) -> int:
        """Records of t = (bytes, offsets, lengths, n) whose tag is <= the tag of record other_index of `other`."""
        tt, to = self._tags(*t), self._tags(*other)
        count = C.c_uint64(0)
        self._check(self._L.fqd_count_tags_le(self._h, C.byref(tt), C.byref(to), other_index, C.byref(count)))
        return int(count.value)

    def output_offsets(self, keep, idx, n: int, sizes, dest) -> int:
        total = C.c_uint64(0)
        self._check(self._L.fqd_output_offsets(self._h, self._p(keep), self._p(idx), n, self._p(sizes), self._p(dest), C.byref(total)))
        return int(total.value)

    def output_plan(self, keep, idx, n: int, starts, sizes, src_off, lens, dst_off) -> int:
        total = C.c_uint64(0)
        self._check(self._L.fqd_output_plan(self._h, self._p(keep), self._p(idx), n, self._p(starts), self._p(sizes),
                                            self._p(src_off), self._p(lens), self._p(dst_off), C.byref(total)))
        return int(total.value)

    def scatter_flags(self, flags, origin, n: int, keep_out):
        self._check(self._L.fqd_scatter_flags(self._h, self._p(flags), self._p(origin), n, self._p(keep_out)))

    def synth_reads(self, seed: int, first: int, n: int, length: int, dup_permille: int, mate: int, bases, expect_keep=None):
        self._check(self._L.fqd_synth_reads(self._h, seed, first, n, length, dup_permille, mate,
                                            self._p(bases), self._p(expect_keep)))
