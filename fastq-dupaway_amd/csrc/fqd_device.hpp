// fqd_device.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Key format (replaces the reference's base-5 17-mer words, seq_utils.cpp:23-49,
// with an equally lossless but denser, shift-only packing):
//   per mate, per block of 64 bases:  [codes(bases 0..31)] [codes(bases 32..63), if any] [N-mask(64 bits)]
//   codes: 2 bits per base, base k of the group at bits 2k..2k+1, A=0 C=1 T=2 G=3, N=3 (+ its mask bit)
//   words(L) = ceil(L/32) + ceil(L/64);  L = 150 -> 8 words = 64 B (one HBM line)
// Unused high bits are zero, so for equal lengths  key words equal <=> sequences equal.
// Lengths are compared separately (uniform engines: implied; ragged: header word).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fqd {

constexpr uint64_t kEmptySlot = 0xFFFFFFFFFFFFFFFFull;
constexpr uint64_t kNoError   = 0xFFFFFFFFFFFFFFFFull;

__host__ __device__ inline uint32_t seg_words(uint32_t len) { return (len + 31u) / 32u + (len + 63u) / 64u; }

// ---- hash ------------------------------------------------------------------
// Only places keys in the table (as boost::hash_combine does for the reference,
// hash_dup_remover.hpp:43-68); equality is always decided on the key words.
constexpr uint64_t kHashSeed = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kHashMul  = 0x9FB21C651E98DF25ull;

__host__ __device__ inline uint64_t hash_begin(uint32_t len0, uint32_t len1)
{
    uint64_t h = kHashSeed ^ (uint64_t(len0) | (uint64_t(len1) << 32));
    h *= kHashMul;
    return h ^ (h >> 32);
}
__host__ __device__ inline uint64_t hash_word(uint64_t h, uint64_t w)
{
    h = (h ^ w) * kHashMul;
    return h ^ (h >> 32);
}
__host__ __device__ inline uint64_t hash_end(uint64_t h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 33);
}

// ---- error word ---------------------------------------------------------------
// (record:32 | segment:1 | position:23 | byte:8); atomicMin keeps the first in
// input order, which is the one the reference would have hit (seq_utils.cpp:17-19).
__host__ __device__ inline uint64_t make_error(uint64_t record, uint32_t seg, uint32_t pos, uint32_t byte)
{
    if (pos > 0x7FFFFFu) pos = 0x7FFFFFu;
    return (record << 32) | (uint64_t(seg & 1u) << 31) | (uint64_t(pos) << 8) | (byte & 0xFFu);
}

// ---- the two byte-shuffling instructions the packer leans on -------------------
// Device code uses v_perm_b32 / v_alignbyte_b32 directly; the host bodies exist
// only so tests/host_pack_check.cpp can exercise the same packer without a GPU.
__host__ __device__ __forceinline__ uint32_t perm_bytes(uint32_t s0, uint32_t s1, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);   // selector 0..3 -> bytes of s1, 4..7 -> bytes of s0
#else
    const uint64_t pool = (uint64_t(s0) << 32) | s1;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t c = (sel >> (8 * k)) & 0xFFu;
        const uint32_t b = c < 8u ? uint32_t((pool >> (8u * c)) & 0xFFu) : (c >= 13u ? 0xFFu : 0u);
        r |= b << (8 * k);
    }
    return r;
#endif
}
// Bytes of an aligned dword pair seen through a byte shift.
__host__ __device__ __forceinline__ uint32_t shifted_dword(uint32_t lo, uint32_t hi, uint32_t shift_bytes)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, shift_bytes);
#else
    return uint32_t(((uint64_t(hi) << 32) | lo) >> (8u * (shift_bytes & 3u)));
#endif
}
__host__ __device__ __forceinline__ uint32_t first_set_bit(uint32_t x)   // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(__ffs(int(x)) - 1);
#else
    return uint32_t(__builtin_ctz(x));
#endif
}

// ---- 4 bases at a time -----------------------------------------------------------
// ASCII      A=0x41 C=0x43 G=0x47 T=0x54 N=0x4E
// (c>>1)&7   A=0    C=1    G=3    T=2    N=7      -> low 2 bits = code, bit 2 = "is N"
// A v_perm_b32 lookup of that 3-bit value rebuilds the byte the code stands for;
// any input byte that does not round-trip is outside {A,C,G,T,N}.
struct Quad {
    uint32_t codes;   // 8 bits: base k at bits 2k..2k+1
    uint32_t nmask;   // 4 bits
    uint32_t diff;    // nonzero byte k <=> byte k invalid
};

__host__ __device__ __forceinline__ Quad pack_quad(uint32_t w)
{
    const uint32_t c3 = (w >> 1) & 0x07070707u;
    const uint32_t back = perm_bytes(0x4E000000u, 0x47544341u, c3);
    Quad q;
    q.diff = back ^ w;
    const uint32_t c2 = c3 & 0x03030303u;
    const uint32_t t = c2 | (c2 >> 6);
    q.codes = (t | (t >> 12)) & 0xFFu;
    const uint32_t m = (c3 >> 2) & 0x01010101u;
    const uint32_t u = m | (m >> 7);
    q.nmask = (u | (u >> 14)) & 0xFu;
    return q;
}

// Replaces bytes at index >= n (n in 1..3) of w by 'A' so they pack to zeros.
__host__ __device__ __forceinline__ uint32_t pad_tail(uint32_t w, uint32_t n)
{
    const uint32_t keep = (1u << (8u * n)) - 1u;
    return (w & keep) | (0x41414141u & ~keep);
}

// ---- streaming packer ----------------------------------------------------------
// Feeds consecutive dwords of one mate's sequence; emits key words in layout
// order through `sink(word)` and reports the first invalid byte.
struct Packer {
    uint64_t codes = 0;     // current 32-base group
    uint64_t mask = 0;      // current 64-base block
    uint32_t bases = 0;     // bases consumed so far
    uint32_t bad_pos = 0xFFFFFFFFu;
    uint32_t bad_byte = 0;

    template <class Sink>
    __host__ __device__ __forceinline__ void push(uint32_t w, uint32_t nvalid /*1..4*/, Sink&& sink)
    {
        if (nvalid < 4u) w = pad_tail(w, nvalid);
        const Quad q = pack_quad(w);
        if (q.diff != 0u && bad_pos == 0xFFFFFFFFu) {
            const uint32_t k = first_set_bit(q.diff) >> 3;
            bad_pos = bases + k;
            bad_byte = (w >> (8u * k)) & 0xFFu;
        }
        const uint32_t in_group = bases & 31u;
        const uint32_t in_block = bases & 63u;
        codes |= uint64_t(q.codes) << (2u * in_group);
        mask  |= uint64_t(q.nmask) << in_block;
        bases += nvalid;
        if ((bases & 31u) == 0u && nvalid == 4u) {          // a group filled exactly
            sink(codes); codes = 0;
            if ((bases & 63u) == 0u) { sink(mask); mask = 0; }
        }
    }
    // Flushes the partial group/block after the last push.
    template <class Sink>
    __host__ __device__ __forceinline__ void finish(Sink&& sink)
    {
        if ((bases & 31u) != 0u) { sink(codes); codes = 0; }
        if ((bases & 63u) != 0u) { sink(mask); mask = 0; }
    }
};

} // namespace fqd
