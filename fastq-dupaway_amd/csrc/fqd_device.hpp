// fqd_device.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Key format (replaces the reference's base-5 17-mer words, seq_utils.cpp:23-49,
// with an equally lossless but denser, shift-only packing):
//   per mate, per block of 64 bases:  [codes(bases 0..31)] [codes(bases 32..63), if any] [N-mask(64 bits)]
//   words(L) = ceil(L/32) + ceil(L/64);  L = 150 -> 8 words = 64 B (one HBM line)
//   2-bit code: A=0 C=1 T=2 G=3, N=3 plus its mask bit.
// Bit positions follow what four-bases-per-dword SWAR produces without any
// transposition (any fixed bijection serves equality):  a 32-base group is 8 dwords
// d0..d7 of 4 ASCII bytes; base 4k+j (dword k, byte j) of the group goes to
//   codes word: half = k/4 (low / high 32 bits), bits 8j + 2(k%4) .. +1
//   mask  word: 32 bits per group (low: first group of the block, high: second),
//               bit 8j + k
// Bases past the end count as 'A' (zeros), so for equal lengths
//   key words equal <=> sequences equal.
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
// One key word into the chain.  The chain only has to keep different keys apart (every step is a bijection of the
// state for a given word and of the word for a given state, so keys that differ in one word never meet, and keys that
// differ in more meet with the odds of a 64-bit equation); the spreading over table positions, tags and partition
// digits is hash_end's.  So the step is kept cheap — the staged encoder issues instructions 80 % of the time (SQ
// counters) and a 64-bit multiply is three quarter-rate instructions: round 3's step
// (h ^ w) * K; h ^= h >> 32 cost 16 issue slots per word, this one 6: two Feistel halves over the 32-bit halves of
// h ^ w, a 24-bit multiply (full rate) for the non-linear one, an add of a rotation for the other.
// FQD_OLD_HASH: round 3's step (A/B builds).
__host__ __device__ inline uint64_t hash_word(uint64_t h, uint64_t w)
{
#ifdef FQD_OLD_HASH
    h = (h ^ w) * kHashMul;
    return h ^ (h >> 32);
#else
    const uint64_t x = h ^ w;
    uint32_t lo = uint32_t(x), hi = uint32_t(x >> 32);
    hi ^= (lo & 0xFFFFFFu) * 0x9E3779u;                       // v_mul_u32_u24: low 32 bits of a 24 x 24 bit product
    lo += (hi << 15) | (hi >> 17);                            // v_alignbit + v_add
    return (uint64_t(hi) << 32) | lo;
#endif
}
// A record's hash = hash_end of its mate-1 chain (single-end), or of mate-1's chain fed with
// mate-2's chain (paired): each mate is hashed on its own from hash_begin(len, 0), so the two
// mates of a pair can be encoded by two lanes.
__host__ __device__ inline uint64_t hash_pair(uint64_t mate0_chain, uint64_t mate1_chain);
// kSkipHash in a batch's hash array means "no record at this position" (slab slots the sharded exchange left
// empty): every insert path passes over it.  No record's own hash ever takes that value.
constexpr uint64_t kSkipHash = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kOpaqueKeys = 0xFFFFFFFFu;            // FQD_OPAQUE_KEYS of the ABI: len1 of an engine whose keys are len0 opaque words
__host__ __device__ inline uint64_t hash_end(uint64_t h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h == kSkipHash ? kSkipHash - 1 : h;
}

__host__ __device__ inline uint64_t hash_pair(uint64_t mate0_chain, uint64_t mate1_chain)
{
    return hash_end(hash_word(mate0_chain, mate1_chain));
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
// a | (b ^ c)
__host__ __device__ __forceinline__ uint32_t or_xor(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xF6);       // truth table over (a, b, c) = 0xF0 | (0xCC ^ 0xAA)
#else
    return a | (b ^ c);
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

// ---- 32 bases at a time -----------------------------------------------------------
// ASCII      A=0x41 C=0x43 G=0x47 T=0x54 N=0x4E
// (c>>1)&7   A=0    C=1    G=3    T=2    N=7      -> low 2 bits = code, bit 2 = "is N"
// A v_perm_b32 lookup of that 3-bit value rebuilds the byte the code stands for;
// any input byte that does not round-trip is outside {A,C,G,T,N}.
struct Group {
    uint64_t codes;
    uint32_t nmask;
    uint32_t diff;     // nonzero <=> some byte of the group is invalid
};

// w[0..7]: the group's dwords, bytes past the sequence end already replaced by 'A'.
//
// Three byte-table lookups per dword, all off ONE selector — the byte's low three bits, A=1 C=3 T=4 N=6 G=7 (distinct;
// 0, 2, 5: no base) — do what round 3 did with shifts and masks:
//   * the byte the selector stands for (0xFF for 0, 2, 5: no byte with those low bits equals it): any input byte that
//     does not round-trip is outside {A,C,G,T,N};  diff |= lookup ^ w is one three-input bit operation on gfx950;
//   * the 2-bit code ALREADY SHIFTED to where dword k of the group keeps it (bits 2(k%4).. of the byte);
//   * the N bit already at bit k of the byte.
// So a dword costs AND + 3 x v_perm + 1 bit-op, and the words are plain ORs of the lookups (v_or3): 6 instructions per
// four bases against 8 (the staged encoder issues instructions 80 % of the time: DESIGN §7).  Same key bits as before.
// FQD_PACK_V1: round 3's form (A/B builds).
__host__ __device__ __forceinline__ Group pack_group(const uint32_t (&w)[8])
{
    Group g;
#ifdef FQD_PACK_V1
    uint32_t c[8], diff = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c[k] = (w[k] >> 1) & 0x07070707u;
        diff |= perm_bytes(0x4E000000u, 0x47544341u, c[k]) ^ w[k];
    }
    const uint32_t M = 0x03030303u, N = 0x04040404u;
    const uint32_t lo = (c[0] & M) | ((c[1] & M) << 2) | ((c[2] & M) << 4) | ((c[3] & M) << 6);
    const uint32_t hi = (c[4] & M) | ((c[5] & M) << 2) | ((c[6] & M) << 4) | ((c[7] & M) << 6);
    const uint32_t y0 = ((c[0] & N) >> 2) | ((c[1] & N) >> 1) | (c[2] & N) | ((c[3] & N) << 1);
    const uint32_t y1 = ((c[4] & N) << 2) | ((c[5] & N) << 3) | ((c[6] & N) << 4) | ((c[7] & N) << 5);
    g.codes = uint64_t(lo) | (uint64_t(hi) << 32);
    g.nmask = y0 | y1;
    g.diff = diff;
#else
    // table bytes by selector: s1 = selectors 0..3 = (-, A, -, C), s0 = selectors 4..7 = (T, -, N, G)
    uint32_t cod[8], nb[8], diff = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t sel = w[k] & 0x07070707u;
        diff = or_xor(diff, perm_bytes(0x474EFF54u, 0x43FF41FFu, sel), w[k]);
        const uint32_t sh = 2u * (uint32_t(k) & 3u);
        cod[k] = perm_bytes((2u << sh) | (3u << (16 + sh)) | (3u << (24 + sh)), 1u << (24 + sh), sel);     // A0 C1 T2 N3 G3
        nb[k]  = perm_bytes(1u << (16 + k), 0u, sel);                                                       // N only
    }
    const uint32_t lo = cod[0] | cod[1] | cod[2] | cod[3], hi = cod[4] | cod[5] | cod[6] | cod[7];
    g.codes = uint64_t(lo) | (uint64_t(hi) << 32);
    g.nmask = (nb[0] | nb[1] | nb[2] | nb[3]) | (nb[4] | nb[5] | nb[6] | nb[7]);
    g.diff = diff;
#endif
    return g;
}

// Replaces bytes at index >= n (n in 0..4) of w by 'A' so they pack to zeros.
__host__ __device__ __forceinline__ uint32_t pad_tail(uint32_t w, uint32_t n)
{
    const uint32_t keep = n >= 4u ? 0xFFFFFFFFu : ((1u << (8u * n)) - 1u);
    return (w & keep) | (0x41414141u & ~keep);
}

// First byte of [p, p+len) outside {A,C,G,T,N}: the slow path behind a nonzero diff.
__host__ __device__ inline uint32_t first_bad_base(const uint8_t* p, uint32_t len, uint32_t* byte)
{
    for (uint32_t k = 0; k < len; ++k) {
        const uint8_t c = p[k];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N') { *byte = c; return k; }
    }
    *byte = 0;
    return 0xFFFFFFFFu;
}

// ---- one mate, group by group -------------------------------------------------------
// q: dword-aligned window, the sequence starts `sh` bytes (0..3) into q[0].  Emits key
// words in layout order through sink(word); returns the OR of the groups' diffs.
// Never reads a dword that holds no byte of the sequence.
template <class Sink>
__host__ __device__ __forceinline__ uint32_t pack_mate(const uint32_t* __restrict__ q, uint32_t sh, uint32_t len, Sink&& sink)
{
    const uint32_t n_src = (sh + len + 3u) >> 2;          // aligned dwords that hold sequence bytes
    const uint32_t n_full = len >> 5;                     // groups of exactly 32 bases
    uint32_t diff = 0;
    uint32_t mask_lo = 0;
    uint32_t carry = n_src ? q[0] : 0u;
    uint32_t g = 0;
    for (; g < n_full; ++g) {                             // every dword of these groups is in range
        uint32_t d[9], w[8];
        d[0] = carry;
#pragma unroll
        for (int k = 1; k < 8; ++k) d[k] = q[8u * g + k];
        d[8] = (8u * g + 8u < n_src) ? q[8u * g + 8u] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = shifted_dword(d[k], d[k + 1], sh);
        carry = d[8];
        const Group r = pack_group(w);
        diff |= r.diff;
        sink(r.codes);
        if (g & 1u) { sink(uint64_t(mask_lo) | (uint64_t(r.nmask) << 32)); mask_lo = 0; }
        else        mask_lo = r.nmask;
    }
    const uint32_t rem = len & 31u;
    if (rem) {                                            // last, partial group
        uint32_t d[9], w[8];
        d[0] = carry;
#pragma unroll
        for (int k = 1; k < 9; ++k) d[k] = (8u * g + k < n_src) ? q[8u * g + k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t have = rem > 4u * k ? rem - 4u * k : 0u;
            w[k] = pad_tail(shifted_dword(d[k], d[k + 1], sh), have);
        }
        const Group r = pack_group(w);
        diff |= r.diff;
        sink(r.codes);
        if (g & 1u) { sink(uint64_t(mask_lo) | (uint64_t(r.nmask) << 32)); mask_lo = 0; g = 0; }
        else        { sink(uint64_t(r.nmask)); g = 0; }
    } else if (n_full & 1u) {
        sink(uint64_t(mask_lo));                          // odd number of full groups: flush the half block
    }
    return diff;
}

} // namespace fqd
