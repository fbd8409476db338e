"""Locates, builds and loads lib/libfqdupaway.so and declares the C ABI to ctypes."""
import ctypes as C
import re
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
import os as _os
# FQD_LIBRARY: another build of the same library (diagnostics: the phase-stamped one of `make STAMPS=1`)
LIB_PATH = Path(_os.environ["FQD_LIBRARY"]) if _os.environ.get("FQD_LIBRARY") else PKG_DIR / "lib" / "libfqdupaway.so"
CLI_PATH = PKG_DIR / "bin" / "fastq-dupaway"
HEADER = REPO_ROOT / "include" / "fqdupaway.h"

OK, ERR_ARG, ERR_HIP, ERR_BAD_BASE, ERR_CAPACITY, ERR_NO_DEVICE = range(6)
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_PROFILE, FLAG_NO_STAGE, FLAG_WEAK_HASH = 1, 2, 4


class FqdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"fqdupaway error {code}: {message}")
        self.code = code
        self.message = message


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("segments", C.c_int32), ("capacity_reads", C.c_uint64),
                ("capacity_bases", C.c_uint64), ("stream", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class ReadsDesc(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("lengths", C.c_void_p),
                ("uniform_len", C.c_uint32), ("uniform_stride", C.c_uint32)]


class TagsDesc(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("offsets", C.c_void_p), ("lengths", C.c_void_p), ("n", C.c_uint64)]


class JoinDesc(C.Structure):
    _fields_ = [("perm_a", C.c_void_p), ("perm_b", C.c_void_p), ("match_a", C.c_void_p), ("match_b", C.c_void_p),
                ("pair_a", C.c_void_p), ("pair_b", C.c_void_p), ("n_pairs", C.POINTER(C.c_uint64))]


class Stats(C.Structure):
    _fields_ = [("records", C.c_uint64), ("duplicates", C.c_uint64), ("table_slots", C.c_uint64), ("key_bytes", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("encode_ms", C.c_double), ("encode_launches", C.c_uint64), ("encode_reads", C.c_uint64),
                ("insert_ms", C.c_double), ("insert_launches", C.c_uint64), ("insert_reads", C.c_uint64),
                ("partition_ms", C.c_double), ("partition_launches", C.c_uint64), ("partition_reads", C.c_uint64),
                ("dedup_ms", C.c_double), ("dedup_launches", C.c_uint64), ("dedup_reads", C.c_uint64),
                ("other_ms", C.c_double), ("other_launches", C.c_uint64)]


class ShardConfig(C.Structure):
    _fields_ = [("world", C.c_int32), ("n_local", C.c_int32), ("first_rank", C.c_int32), ("transport", C.c_int32),
                ("round_reads", C.c_uint64), ("len0", C.c_uint32), ("len1", C.c_uint32), ("slack_permille", C.c_uint32),
                ("flags", C.c_uint32), ("slab_records", C.c_uint64), ("unique_id", C.c_void_p)]


class ShardStats(C.Structure):
    _fields_ = [("rounds", C.c_uint64), ("overflow_rounds", C.c_uint64), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64),
                ("slab_records", C.c_uint64), ("exchange_ms", C.c_double), ("transport", C.c_int32), ("ranks_in_comm", C.c_int32)]


SHARD_RCCL, SHARD_COPY, SHARD_ID_BYTES, SHARD_PADDED, SHARD_SEND_HASH, OPAQUE_KEYS = 0, 1, 128, 1, 2, 0xFFFFFFFF


def build_native(target: str = "all") -> None:
    """hipcc --offload-arch=gfx950 build of the library (and CLI); cross-compiles without a GPU."""
    subprocess.run(["make", "-s", "-C", str(PKG_DIR), target], check=True)


def declared_symbols():
    """Every function name include/fqdupaway.h declares."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(fqd_[a-z_0-9]+)\s*\(", text)))


_lib = None


def load_library():
    """Loads the HIP library or raises: there is deliberately no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FqdError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() "
                                      f"(make -C {PKG_DIR}); this engine has no CPU path")
    # torch bundles its own libamdhip64 (same soname as /opt/rocm's).  One process must hold ONE
    # HIP runtime, so when torch is installed it is imported first and this library binds to the
    # runtime torch already loaded; otherwise torch would find ours and see no devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(LIB_PATH))
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.fqd_abi_version.restype = i32
    L.fqd_device_count.argtypes = [C.POINTER(i32)]
    L.fqd_engine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.fqd_engine_destroy.argtypes = [vp]
    L.fqd_engine_reset.argtypes = [vp]
    L.fqd_submit.argtypes = [vp, C.POINTER(ReadsDesc), u64, i32, vp]
    L.fqd_submit_final.argtypes = [vp, C.POINTER(ReadsDesc), u64, i32, vp]
    L.fqd_engine_sync.argtypes = [vp]
    L.fqd_engine_wait_stream.argtypes = [vp, vp]
    L.fqd_stream_wait_engine.argtypes = [vp, vp]
    L.fqd_engine_stream.argtypes = [vp]
    L.fqd_engine_stream.restype = vp
    L.fqd_bad_base.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint8)]
    L.fqd_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.fqd_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.fqd_reset_profile.argtypes = [vp]
    L.fqd_last_error.argtypes = [vp]
    L.fqd_last_error.restype = C.c_char_p
    L.fqd_key_words.argtypes = [u32, u32]
    L.fqd_key_words.restype = u32
    L.fqd_encode_uniform.argtypes = [vp, C.POINTER(ReadsDesc), u64, vp]
    L.fqd_sort_tags.argtypes = [vp, C.POINTER(TagsDesc), vp]
    L.fqd_extract_tags.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    L.fqd_join_tags.argtypes = [vp, C.POINTER(TagsDesc), C.POINTER(TagsDesc), C.POINTER(JoinDesc)]
    L.fqd_gather_seqs.argtypes = [vp, vp, u64, vp, vp, vp, vp]
    L.fqd_copy_spans.argtypes = [vp, vp, vp, vp, u64, vp, vp]
    L.fqd_count_tags_le.argtypes = [vp, C.POINTER(TagsDesc), C.POINTER(TagsDesc), u64, C.POINTER(u64)]
    L.fqd_output_offsets.argtypes = [vp, vp, vp, u64, vp, vp, C.POINTER(u64)]
    L.fqd_output_plan.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.fqd_scatter_flags.argtypes = [vp, vp, vp, u64, vp]
    L.fqd_bgzf_bound.argtypes = [u64]
    L.fqd_bgzf_bound.restype = u64
    L.fqd_bgzf_deflate.argtypes = [vp, vp, u64, u32, vp, u64, C.POINTER(u64)]
    L.fqd_bgzf_inflate.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, vp, C.POINTER(u64)]
    L.fqd_bgzf_inflate_async.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, vp, vp]
    L.fqd_gunzip.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), C.POINTER(C.c_int32)]
    L.fqd_gunzip_arriving.argtypes = [vp, vp, u64, C.POINTER(u64), vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32), C.POINTER(C.c_int32)]
    L.fqd_count_lines.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.fqd_scan_records.argtypes = [vp, vp, u64, u32, u64, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.fqd_partition_keys.argtypes = [vp, vp, u64, u32, u32, vp, vp, vp]
    L.fqd_reserve_keys.argtypes = [vp, u64, u32, u32, C.POINTER(vp)]
    L.fqd_insert_keys.argtypes = [vp, vp, u64, u32, u32, vp]
    L.fqd_widen_keys.argtypes = [vp, u32]
    L.fqd_synth_reads.argtypes = [vp, u64, u64, u64, u32, u32, i32, vp, vp]
    L.fqd_padded_key_words.argtypes = [u32, u32]
    L.fqd_padded_key_words.restype = u32
    L.fqd_encode_padded.argtypes = [vp, C.POINTER(ReadsDesc), u64, u32, u32, vp]
    L.fqd_sample_tags.argtypes = [vp, C.POINTER(TagsDesc), u32, u32, vp, vp]
    L.fqd_classify_tags.argtypes = [vp, C.POINTER(TagsDesc), vp, u32, vp, u32, vp]
    L.fqd_range_keep.argtypes = [vp, vp, u64, u32, vp, C.POINTER(u64)]
    L.fqd_max_u32.argtypes = [vp, vp, u64, C.POINTER(u32)]
    L.fqd_partition_slabs.argtypes = [vp, vp, u64, u32, u32, u64, vp, vp, vp]
    L.fqd_encode_slabs.argtypes = [vp, C.POINTER(ReadsDesc), u64, u32, u64, u32, u64, vp, vp, vp, vp, u32]
    L.fqd_insert_slabs.argtypes = [vp, vp, u32, u64, vp, u32, u32, vp]
    L.fqd_insert_slabs_hashed.argtypes = [vp, vp, vp, u32, u64, vp, u32, u32, vp]
    L.fqd_encode_slabs_hashed.argtypes = [vp, C.POINTER(ReadsDesc), u64, u32, u64, u32, u64, vp, vp, vp, vp, vp, u32]
    L.fqd_shard_unique_id.argtypes = [vp]
    L.fqd_shard_slab_capacity.argtypes = [u64, C.c_int32, u32]
    L.fqd_shard_slab_capacity.restype = u64
    L.fqd_shard_create.argtypes = [C.POINTER(vp), C.POINTER(ShardConfig), C.POINTER(vp)]
    L.fqd_shard_destroy.argtypes = [vp]
    L.fqd_shard_round.argtypes = [vp, C.POINTER(ReadsDesc), C.POINTER(u64), C.POINTER(vp)]
    L.fqd_shard_flush.argtypes = [vp]
    L.fqd_shard_wait.argtypes = [vp, u64]
    L.fqd_shard_bad_base.argtypes = [vp, u64, C.POINTER(C.c_int32), C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint8)]
    L.fqd_shard_get_stats.argtypes = [vp, C.c_int32, C.POINTER(ShardStats)]
    L.fqd_shard_last_error.argtypes = [vp]
    L.fqd_shard_last_error.restype = C.c_char_p
    for name in declared_symbols():
        fn = getattr(L, name)          # AttributeError here = header and library disagree
        if name not in ("fqd_last_error", "fqd_key_words", "fqd_engine_stream", "fqd_bgzf_bound", "fqd_shard_last_error", "fqd_shard_slab_capacity", "fqd_padded_key_words"):
            fn.restype = i32
    _lib = L
    return L
