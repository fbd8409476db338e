"""ctypes wrapper over the shard group of include/fqdupaway.h (fqd_shard_*): one dedup job over several GPUs,
hash-prefix sharded, fixed-size all-to-all slabs over RCCL (or peer copies between ranks of one process).
Plumbing only — the exchange, its pipeline and the overflow handling live in csrc/fqd_shard.hip."""
import ctypes as C
from typing import Optional, Sequence

from . import _lib
from ._lib import FqdError, load_library
from .engine import Engine, Reads


def unique_id() -> bytes:
    """The id rank 0 of a multi-process group makes and every process passes to ShardGroup."""
    L = load_library()
    buf = (C.c_uint8 * _lib.SHARD_ID_BYTES)()
    rc = L.fqd_shard_unique_id(buf)
    if rc != _lib.OK:
        raise FqdError(rc, (L.fqd_shard_last_error(None) or b"").decode())
    return bytes(buf)


class ShardGroup:
    """The ranks of a shard group that live in this process.  engines: one Engine per local rank."""

    def __init__(self, engines: Sequence[Engine], world: int, first_rank: int, round_reads: int, len0: int, len1: int = 0,
                 transport: str = "rccl", uid: Optional[bytes] = None, slack_permille: int = 0, slab_records: int = 0,
                 padded: bool = False, send_hash: bool = False):
        """padded: len0/len1 are the longest reads allowed and batches may hold reads of any lengths up to them.
        send_hash: every key's placement hash travels with it (FQD_SHARD_SEND_HASH; ignored with padded)."""
        self._L = load_library()
        self.engines = list(engines)
        self.world, self.first_rank, self.S = world, first_rank, (2 if len1 else 1)
        self._uid = (C.c_uint8 * _lib.SHARD_ID_BYTES).from_buffer_copy(uid) if uid is not None else None
        cfg = _lib.ShardConfig(world=world, n_local=len(self.engines), first_rank=first_rank,
                               transport=_lib.SHARD_RCCL if transport == "rccl" else _lib.SHARD_COPY,
                               round_reads=round_reads, len0=len0, len1=len1, slack_permille=slack_permille,
                               flags=(_lib.SHARD_PADDED if padded else 0) | (_lib.SHARD_SEND_HASH if send_hash else 0), slab_records=slab_records,
                               unique_id=C.cast(self._uid, C.c_void_p) if self._uid is not None else None)
        handles = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        h = C.c_void_p()
        rc = self._L.fqd_shard_create(handles, C.byref(cfg), C.byref(h))
        if rc != _lib.OK:
            raise FqdError(rc, (self._L.fqd_shard_last_error(None) or b"").decode())
        self._h = h
        self.rounds = 0

    def _check(self, rc):
        if rc != _lib.OK:
            raise FqdError(rc, (self._L.fqd_shard_last_error(self._h) or b"").decode())

    def round(self, segs: Sequence[Sequence[Reads]], n: Sequence[int], keep: Sequence) -> int:
        """segs[r] = the mate descriptors of local rank r's batch (uniform, device memory), n[r] its reads,
        keep[r] a device uint8 array.  Asynchronous; returns the round's number."""
        nl = len(self.engines)
        arr = (_lib.ReadsDesc * (nl * self.S))()
        for r in range(nl):
            p = self.engines[r]._p                   # orders rank r's engine after torch's stream (ORDERING RULE of the header)
            for m in range(self.S):
                s = segs[r][m]
                d = arr[r * self.S + m]
                d.bases, d.offsets, d.lengths = p(s.bases), p(s.offsets), p(s.lengths)
                d.uniform_len, d.uniform_stride = s.uniform_len, s.uniform_stride
        ns = (C.c_uint64 * nl)(*[int(x) for x in n])
        ks = (C.c_void_p * nl)(*[self.engines[r]._p(k) for r, k in enumerate(keep)])
        for e in self.engines:
            e._ordered = False
        self._check(self._L.fqd_shard_round(self._h, arr, ns, ks))
        self.rounds += 1
        return self.rounds - 1

    def flush(self):
        self._check(self._L.fqd_shard_flush(self._h))

    def wait(self, round_index: int):
        self._check(self._L.fqd_shard_wait(self._h, round_index))

    def stats(self, local_rank: int = 0) -> dict:
        st = _lib.ShardStats()
        self._check(self._L.fqd_shard_get_stats(self._h, local_rank, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self._L.fqd_shard_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
