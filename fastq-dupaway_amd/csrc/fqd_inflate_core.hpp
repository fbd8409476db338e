// fqd_inflate_core.hpp — one raw deflate stream (RFC 1951) decoded by ONE thread: the part of the GPU
// BGZF reader that does not care where it runs.
//
// The reference reads `.gz` inputs through Boost's gzip_decompressor, one thread per file
// (file_utils.hpp:58-69).  A BGZF file is a sequence of independent members of at most 64 KiB, so
// here every member gets its own GPU thread (fqd_inflate.hip: 64 members per wave, tables in LDS) and
// a file of a million members is inflated in a handful of rounds — the host only reads the
// compressed bytes and walks the member headers.
//
// The decoder works on the canonical form of a Huffman code: the symbols in code order (symbol[]) and,
// per code length, where its codes end when written left-justified — fifteen numbers that stay in
// registers.  The length of the next code is then fifteen comparisons away, its symbol one LDS read
// more.  That needs 420 bytes per thread (PackedTables below) — six waves' worth fit a CU's LDS — where a
// lookup-table decoder would need kilobytes per thread.  The main loop is a state machine that does
// a bounded amount of work per turn (one symbol decoded, or a few bytes of a match copied), so the
// 64 members of a wave advance side by side instead of waiting out each other's long copies.
//
// `Tables` is where the per-thread arrays live: PackedTables<64> in LDS on the GPU, PackedTables<1> over a
// plain array in tests/native/inflate_core_check.cpp, which runs this very code on the CPU against zlib.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define FQD_HD __host__ __device__ __forceinline__
#else
#define FQD_HD inline
#endif

namespace fqd {
namespace inflate {

constexpr uint32_t kMaxBits = 15;
constexpr uint32_t kBurst = 32;                  // literals decoded back to back before the other states get their turn
constexpr uint32_t kLitSymbols = 288, kDistSymbols = 30;
// entries of a thread's table space (uint16 each): lit count[16], lit symbol[288], dist count[16], dist symbol[32]
constexpr uint32_t kLitCount = 0, kLitSymbol = 16, kDistCount = 16 + 288, kDistSymbol = 16 + 288 + 16;
constexpr uint32_t kTableEntries = 16 + 288 + 16 + 32;

// Where a thread's tables live: 32 sixteen-bit "base" entries (lit 0..15, dist 16..31), the literal/length
// symbols as a byte plane plus a ninth-bit plane, the distance symbols as bytes — 420 bytes per thread, so
// that six waves of 64 threads fit the 160 KB of a CU.  Consecutive entries of one thread lie `Stride`
// elements apart (64 on the GPU: the lanes of a wave side by side in every row; 1 in the CPU harness).
constexpr uint32_t kPackedBytes = 32 * 2 + 288 + 36 + 32;
template <uint32_t Stride>
struct PackedTables {
    uint16_t* base16;                 // 32 rows
    uint8_t* lit_lo;                  // 288 rows
    uint8_t* lit_hi;                  // 36 rows: bit (j & 7) of row (j >> 3) = ninth bit of literal/length symbol j
    uint8_t* dist;                    // 32 rows
    // `block` = the wave's (or thread's) table memory, `lane` = this thread's column
    FQD_HD PackedTables(uint8_t* block, uint32_t lane)
        : base16(reinterpret_cast<uint16_t*>(block) + lane), lit_lo(block + 64u * Stride + lane),
          lit_hi(block + (64u + 288u) * Stride + lane), dist(block + (64u + 288u + 36u) * Stride + lane) {}
    // wave vote (the CPU harness is a wave of one)
    FQD_HD bool most_lanes(bool mine) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return 2 * __popcll(__ballot(mine)) >= __popcll(__ballot(true));
#else
        return mine;
#endif
    }
    FQD_HD static uint32_t base_row(uint32_t i) { return i < kDistCount ? i : 16u + (i - kDistCount); }
    FQD_HD uint16_t base_get(uint32_t i) const { return base16[base_row(i) * Stride]; }
    FQD_HD void base_set(uint32_t i, uint16_t v) { base16[base_row(i) * Stride] = v; }
    FQD_HD uint32_t lit_sym(uint32_t j) const         // j-th literal/length symbol in code order
    {
        return uint32_t(lit_lo[j * Stride]) | (((uint32_t(lit_hi[(j >> 3) * Stride]) >> (j & 7u)) & 1u) << 8);
    }
    FQD_HD uint32_t dist_sym(uint32_t j) const { return dist[j * Stride]; }
    FQD_HD uint32_t sym_get(uint32_t i) const { return i >= kDistSymbol ? dist_sym(i - kDistSymbol) : lit_sym(i - kLitSymbol); }
    FQD_HD void sym_set(uint32_t i, uint32_t v)
    {
        if (i >= kDistSymbol) { dist[(i - kDistSymbol) * Stride] = uint8_t(v); return; }
        const uint32_t j = i - kLitSymbol;
        lit_lo[j * Stride] = uint8_t(v);
        uint8_t& h = lit_hi[(j >> 3) * Stride];
        h = uint8_t((h & ~(1u << (j & 7u))) | (((v >> 8) & 1u) << (j & 7u)));
    }
};

enum Status : uint32_t { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadLengths = 3, kBadCode = 4, kBadDistance = 5,
                         kOutputOverrun = 6, kInputOverrun = 7, kShortOutput = 8 };

// Bits of the stream, least significant first, fetched as aligned 32-bit words, one word ahead of
// the one being consumed (its load is in flight while the bits before it are decoded).  Reading past
// the end of the member is counted (and reported as kInputOverrun) instead of performed.
struct BitReader {
    const uint32_t* words;           // aligned base
    uint64_t buf = 0;
    uint32_t cnt = 0;                // valid bits in buf
    uint32_t next = 0;               // index of the word held in `ahead`
    uint32_t end_word;               // words available (whole words covering the member)
    uint32_t ahead = 0;
    uint32_t overrun = 0;

    FQD_HD BitReader(const uint8_t* p, uint32_t nbytes)
    {
        // (stepping back from p, not rebuilding the pointer from an integer: the compiler then still knows that it
        //  points to global memory and its loads do not share a wait counter with the LDS reads)
        const uint32_t skip = uint32_t(reinterpret_cast<uintptr_t>(p) & 3u);
        words = reinterpret_cast<const uint32_t*>(p - skip);
        end_word = (skip + nbytes + 3u) / 4u;
        ahead = end_word ? words[0] : 0u;
        fill();
        buf >>= 8u * skip; cnt -= 8u * skip;
    }
    FQD_HD void fill()
    {
        while (cnt <= 32u) {
            if (next >= end_word) ++overrun;
            buf |= uint64_t(ahead) << cnt;
            cnt += 32u;
            ++next;
            ahead = next < end_word ? words[next] : 0u;
        }
    }
    FQD_HD uint32_t bits(uint32_t n)                  // n <= 16
    {
        if (cnt < n) fill();
        const uint32_t v = uint32_t(buf) & ((1u << n) - 1u);
        buf >>= n; cnt -= n;
        return v;
    }
    FQD_HD void align_to_byte() { const uint32_t r = cnt & 7u; buf >>= r; cnt -= r; }
};

FQD_HD uint32_t reverse_bits32(uint32_t v)
{
#if defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
#endif
}

// A canonical code seen from the decoder: written left-justified in 15 bits, every code of length l
// is smaller than every longer one, so lim[l-1] = end of the codes of length <= l finds the length of
// the code at the head of the stream with fifteen comparisons — in registers, no table walk — and
// base[l] (with the thread's tables, entry base_at + l) turns the code into its place in symbol[].
struct Code { uint32_t lim[kMaxBits]; };

template <class Tables>
FQD_HD uint32_t decode_symbol(BitReader& in, const Tables& t, const Code& c, uint32_t base_at, uint32_t symbol_at)
{
    if (in.cnt < kMaxBits) in.fill();
    const uint32_t w = reverse_bits32(uint32_t(in.buf)) >> 17;        // 15 bits, the first bit of the stream on top
    uint32_t len = 1;
#pragma unroll
    for (uint32_t l = 0; l + 1 < kMaxBits; ++l) len += w >= c.lim[l] ? 1u : 0u;
    if (w >= c.lim[kMaxBits - 1]) return 0xFFFFu;                     // no code starts like this
    const uint32_t index = uint32_t(int32_t(int16_t(t.base_get(base_at + len))) + int32_t(w >> (kMaxBits - len)));
    in.buf >>= len; in.cnt -= len;
    return symbol_at == kLitSymbol ? t.lit_sym(index) : t.dist_sym(index);   // (a constant at every call site)
}

// Tables and comparison limits from code lengths len(0..n-1) (0 = symbol unused).  Returns false for an
// over-subscribed set; an incomplete one passes only where zlib lets it pass: a distance code with no
// code at all or with a single one-bit code.
template <class Tables, class Len>
FQD_HD bool build_table(Tables& t, uint32_t base_at, uint32_t symbol_at, uint32_t n, const Len& len, Code& c, bool may_be_single = false)
{
    for (uint32_t l = 0; l <= kMaxBits; ++l) t.base_set(base_at + l, 0);
    for (uint32_t s = 0; s < n; ++s) t.base_set(base_at + len(s), uint16_t(t.base_get(base_at + len(s)) + 1u));
    int32_t left = 1;
    uint32_t codes = 0;
    for (uint32_t l = 1; l <= kMaxBits; ++l) {
        left <<= 1;
        left -= int32_t(t.base_get(base_at + l));
        codes += t.base_get(base_at + l);
        if (left < 0) return false;
    }
    if (left > 0 && !(may_be_single && (codes == 0u || (codes == 1u && t.base_get(base_at + 1u) == 1u)))) return false;
    // per length: first code, place of its first symbol; the count makes room for the running place
    uint32_t code = 0, offset = 0;
    int32_t base[kMaxBits];
#pragma unroll
    for (uint32_t l = 1; l <= kMaxBits; ++l) {
        const uint32_t count = t.base_get(base_at + l);
        base[l - 1] = int32_t(offset) - int32_t(code);
        c.lim[l - 1] = (code + count) << (kMaxBits - l);
        t.base_set(base_at + l, uint16_t(offset));
        offset += count;
        code = (code + count) << 1;
    }
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = len(s);
        if (l) { const uint32_t at = t.base_get(base_at + l); t.sym_set(symbol_at + at, s); t.base_set(base_at + l, uint16_t(at + 1u)); }
    }
#pragma unroll
    for (uint32_t l = 1; l <= kMaxBits; ++l) t.base_set(base_at + l, uint16_t(base[l - 1]));
    return true;
}

FQD_HD uint32_t length_base(uint32_t lsym, uint32_t& ebits)           // lsym = symbol - 257, 0..28
{
    if (lsym < 8u) { ebits = 0; return 3u + lsym; }
    if (lsym == 28u) { ebits = 0; return 258u; }
    const uint32_t hb = lsym / 4u + 1u;
    ebits = hb - 2u;
    return 3u + ((1u << hb) | ((lsym & 3u) << ebits));
}

FQD_HD uint32_t dist_base(uint32_t dsym, uint32_t& ebits)             // 0..29
{
    if (dsym < 4u) { ebits = 0; return dsym + 1u; }
    const uint32_t hb = dsym / 2u;
    ebits = hb - 1u;
    return 1u + ((1u << hb) | ((dsym & 1u) << ebits));
}

// The whole member.  `lens` is scratch for the code lengths of one dynamic block (kLitSymbols +
// kDistSymbols + 2 bytes, any memory); returns kOk iff exactly out_len bytes came out.
template <class Tables, class Lens>
FQD_HD uint32_t inflate_member(const uint8_t* in_bytes, uint32_t in_len, uint8_t* out, uint32_t out_len, Tables& t, Lens& lens)
{
    BitReader in(in_bytes, in_len);
    Code lit, dst;
    for (uint32_t l = 0; l < kMaxBits; ++l) lit.lim[l] = dst.lim[l] = 0;
    uint32_t pos = 0;
    enum { kHeader, kStoredCopy, kSymbols, kMatchCopy, kDone } state = kHeader;
    uint32_t last = 0, left = 0, dist = 0, status = kOk;
    constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    while (state != kDone) {
        if (state == kSymbols) {
            // Literals are most of what a FASTQ stream holds, and a turn of this loop costs every lane of the wave
            // every branch some lane takes: a run of up to kBurst literals is therefore decoded in a tight loop of
            // its own, kept up as long as most lanes of the wave are still in it (a stream full of matches would
            // otherwise wait out the few lanes that are).
            uint32_t sym = decode_symbol(in, t, lit, kLitCount, kLitSymbol);
            for (uint32_t burst = 1; burst < kBurst; ++burst) {
                const bool more = sym < 256u && pos + 1u < out_len;
                if (!t.most_lanes(more)) break;                   // the run goes on while most of the wave is in it
                if (more) {
                    out[pos++] = uint8_t(sym);
                    sym = decode_symbol(in, t, lit, kLitCount, kLitSymbol);
                }
            }
            if (sym < 256u) {
                if (pos >= out_len) { status = kOutputOverrun; break; }
                out[pos++] = uint8_t(sym);
            } else if (sym == 256u) {
                state = last ? kDone : kHeader;
            } else {
                if (sym > 285u) { status = kBadCode; break; }
                uint32_t eb;
                left = length_base(sym - 257u, eb);
                left += in.bits(eb);
                const uint32_t dsym = decode_symbol(in, t, dst, kDistCount, kDistSymbol);
                if (dsym >= kDistSymbols) { status = kBadCode; break; }
                dist = dist_base(dsym, eb);
                dist += in.bits(eb);
                if (dist > pos) { status = kBadDistance; break; }
                if (pos + left > out_len) { status = kOutputOverrun; break; }
                state = kMatchCopy;
            }
        } else if (state == kMatchCopy) {
            const uint32_t step = left < 8u ? left : 8u;
            for (uint32_t i = 0; i < step; ++i, ++pos) out[pos] = out[pos - dist];
            left -= step;
            if (!left) state = kSymbols;
        } else if (state == kStoredCopy) {
            const uint32_t step = left < 8u ? left : 8u;
            for (uint32_t i = 0; i < step; ++i) out[pos++] = uint8_t(in.bits(8));
            left -= step;
            if (!left) state = last ? kDone : kHeader;
        } else {                                                      // kHeader
            last = in.bits(1);
            const uint32_t type = in.bits(2);
            if (type == 0u) {
                in.align_to_byte();
                const uint32_t n = in.bits(16), nn = in.bits(16);
                if ((n ^ nn) != 0xFFFFu) { status = kBadStored; break; }
                if (pos + n > out_len) { status = kOutputOverrun; break; }
                left = n;
                state = n ? kStoredCopy : (last ? kDone : kHeader);
            } else if (type == 1u) {
                auto fixed_lit = [](uint32_t s) -> uint32_t { return s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u; };
                auto fixed_dist = [](uint32_t) -> uint32_t { return 5u; };
                build_table(t, kLitCount, kLitSymbol, kLitSymbols, fixed_lit, lit);
                build_table(t, kDistCount, kDistSymbol, 32u, fixed_dist, dst);     // 32 five-bit codes; 30 and 31 never occur in valid data
                state = kSymbols;
            } else if (type == 2u) {
                const uint32_t nlen = in.bits(5) + 257u, ndist = in.bits(5) + 1u, ncode = in.bits(4) + 4u;
                if (nlen > 286u || ndist > kDistSymbols) { status = kBadLengths; break; }
                for (uint32_t i = 0; i < 19u; ++i) lens.set(i, 0);
                for (uint32_t i = 0; i < ncode; ++i) lens.set(order[i], uint8_t(in.bits(3)));
                auto cl = [&](uint32_t s) -> uint32_t { return lens.get(s); };
                // the code-length code borrows the distance tables until the real ones are built
                if (!build_table(t, kDistCount, kDistSymbol, 19u, cl, dst)) { status = kBadLengths; break; }
                uint32_t i = 0;
                bool bad = false;
                while (i < nlen + ndist) {
                    const uint32_t sym = decode_symbol(in, t, dst, kDistCount, kDistSymbol);
                    if (sym < 16u) { lens.set(i++, uint8_t(sym)); continue; }
                    uint32_t prev = 0, rep;
                    if (sym == 16u) { if (i == 0) { bad = true; break; } prev = lens.get(i - 1u); rep = 3u + in.bits(2); }
                    else if (sym == 17u) rep = 3u + in.bits(3);
                    else if (sym == 18u) rep = 11u + in.bits(7);
                    else { bad = true; break; }
                    if (i + rep > nlen + ndist) { bad = true; break; }
                    while (rep--) lens.set(i++, uint8_t(prev));
                }
                if (bad || lens.get(256) == 0) { status = kBadLengths; break; }
                auto ll = [&](uint32_t s) -> uint32_t { return lens.get(s); };
                auto dl = [&](uint32_t s) -> uint32_t { return lens.get(nlen + s); };
                if (!build_table(t, kLitCount, kLitSymbol, nlen, ll, lit) || !build_table(t, kDistCount, kDistSymbol, ndist, dl, dst, true)) { status = kBadLengths; break; }
                state = kSymbols;
            } else { status = kBadBlockType; break; }
        }
        if (in.overrun > 2u) { status = kInputOverrun; break; }       // the look-ahead may touch one word beyond the end
    }
    if (status == kOk && pos != out_len) status = kShortOutput;
    return status;
}

} // namespace inflate
} // namespace fqd
