// fqd_gunzip_core.hpp — an ORDINARY gzip member (one long deflate stream, no member sizes: what gzip, pigz and sequencer
// software write) inflated by MANY decoders at once: the part that does not care where it runs.
//
// The reference reads `.gz` inputs through Boost's gzip_decompressor on its one thread (file_utils.cpp:59-66).  BGZF files
// carry their members' sizes and are inflated one wave per member (fqd_inflate_wave.hpp); an ordinary stream hides two
// things from a second decoder: where a block starts, and the 32 KiB of text before it that its matches may reach into.
// Both are dealt with as pugz / rapidgzip and this build's host reader (host/pgzip.hpp) do, here for the GPU:
//
//   1. the compressed bytes are cut into UNITS of equal size; for every unit the first bit offset at or after its start
//      where a dynamic, non-final block header parses — length codes decode, both codes complete — is LOOKED FOR
//      (find_block_start: 64 offsets at a time on a wave): a guess, right but for one time in many millions;
//   2. every unit is decoded from its guessed start to the first block boundary at or after the next unit's nominal start
//      where such a block begins (decode_unit: one decoder per unit, serial, tables in LDS) into 16-bit SYMBOLS: a byte, or
//      "the byte this far back in the 32 KiB before my first byte" wherever a match reaches there — copies of such symbols
//      stay symbols, so nothing about the text before the unit has to be known;
//   3. the chain is checked: a unit counts only if it starts at the very bit the unit before it ended at (a wrong guess, a
//      missed boundary or damage breaks the chain: the caller then reads the file the host way, whose diagnostics are the
//      reference's); output offsets are the running sum of the units' symbol counts;
//   4. windows: unit u's last 32 KiB of TEXT from its symbols and unit u-1's window, unit after unit (short: 32 KiB each);
//   5. every symbol becomes a byte, all units at once; CRC-32 and ISIZE of the whole against the member's trailer.
//
// Written as `FQD_HD` functions over plain pointers: fqd_gunzip.hip runs them on the GPU (a unit's decoder is lane 0 of a
// wave whose other lanes move its output), tests/native/gunzip_core_check.cpp on the CPU against zlib, under the sanitizers.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define FQD_HD __host__ __device__ __forceinline__
#define FQD_HD_CALL __host__ __device__ __attribute__((noinline))     /* the long ones: called, not inlined — a kernel that inlines the whole decoder is 6700 instructions of one control-flow graph */
#else
#define FQD_HD inline
#define FQD_HD_CALL inline
#endif

namespace fqd {
namespace gunz {

constexpr uint32_t kWindow = 32768;
constexpr uint32_t kLitBits = 10, kDistBits = 9;
constexpr uint32_t kNoStart = 0xFFFFFFFFu;

enum Status : uint32_t { kOk = 0, kBoundary = 1 /* stopped where the next unit begins */, kFinal = 2 /* through the final block */,
                         kBadData = 3, kOutputFull = 4, kInputEnd = 5 };

// Bits of the stream, least significant first, read as aligned 64-bit words (the buffer has 32 readable bytes behind it: the decoder loads a word ahead).
struct BitIn {
    const uint64_t* words = nullptr;     // 8-byte aligned
    uint64_t lead = 0;                   // bits of words[0] before the stream's first bit
    uint64_t nbits = 0;                  // bits of the stream
    FQD_HD uint64_t peek(uint64_t pos) const                    // >= 57 bits from stream bit `pos` on (what lies beyond the stream is whatever the buffer holds)
    {
        const uint64_t p = pos + lead, i = p >> 6;
        const uint32_t s = uint32_t(p & 63u);
        const uint64_t lo = words[i], hi = words[i + 1];
        return s ? (lo >> s) | (hi << (64u - s)) : lo;
    }
};

// One canonical Huffman code: first-level table + what the slow way needs for codes longer than its index.
// Table entries (32 bits), as host/pgzip.hpp has them: bits 0-3 code length (0: longer than the index, or no code), bits
// 4-5 kind (0 literal, 1 length, 2 end of block, 3 distance), bits 8-12 extra bits, bits 16-31 the literal / the base.
struct Code {
    uint32_t lim[16];                    // [l]: end of the codes of length <= l, left-justified in 15 bits
    int32_t  base[16];
    uint16_t sorted[288];
};
struct Tables {
    uint32_t lit[1u << kLitBits];
    uint32_t dist[1u << kDistBits];
    Code     lc, dc;
};

FQD_HD uint32_t lit_entry(uint32_t sym, uint32_t len)
{
    // length bases and extra bits (RFC 1951 §3.2.5) by arithmetic: symbols 257..264 -> 3..10 (0 extra); then groups of four
    if (sym < 256u) return len | (sym << 16);
    if (sym == 256u) return len | (2u << 4);
    if (sym > 285u) return 0u;
    if (sym == 285u) return len | (1u << 4) | (258u << 16);
    const uint32_t k = sym - 257u;
    if (k < 8u) return len | (1u << 4) | ((3u + k) << 16);
    const uint32_t ex = (k >> 2) - 1u, b = 3u + ((4u + (k & 3u)) << ex);
    return len | (1u << 4) | (ex << 8) | (b << 16);
}
FQD_HD uint32_t dist_entry(uint32_t ds, uint32_t len)
{
    if (ds >= 30u) return 0u;
    if (ds < 4u) return len | (3u << 4) | ((1u + ds) << 16);
    const uint32_t ex = (ds >> 1) - 1u, b = 1u + ((2u + (ds & 1u)) << ex);
    return len | (3u << 4) | (ex << 8) | (b << 16);
}

FQD_HD uint32_t reverse15(uint32_t v)                        // the low 15 bits, first bit on top
{
#if defined(__clang__)
    return __builtin_bitreverse32(v) >> 17;
#else
    uint32_t r = 0; for (int b = 0; b < 15; ++b) r |= ((v >> b) & 1u) << (14 - b); return r;
#endif
}

// Builds one code from its lengths.  false: over-subscribed, or incomplete where zlib does not let that pass.
// which: 0 literal/length (table lit, kLitBits), 1 distance (table dist, kDistBits).
FQD_HD_CALL bool build_code(const uint8_t* lens, uint32_t n, bool may_be_single, uint32_t* table, uint32_t index_bits, Code& c, bool is_dist)
{
    uint32_t count[16];
    for (uint32_t l = 0; l < 16u; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n; ++s) ++count[lens[s]];
    int32_t left = 1; uint32_t codes = 0;
    for (uint32_t l = 1; l <= 15u; ++l) { left = (left << 1) - int32_t(count[l]); if (left < 0) return false; codes += count[l]; }
    if (left > 0 && !(may_be_single && (codes == 0u || (codes == 1u && count[1] == 1u)))) return false;
    uint32_t code = 0, offset = 0, offs[16], next_code[16];
    offs[0] = 0; next_code[0] = 0; c.lim[0] = 0; c.base[0] = 0;
    for (uint32_t l = 1; l <= 15u; ++l) {
        c.base[l] = int32_t(offset) - int32_t(code);
        c.lim[l] = (code + count[l]) << (15u - l);
        offs[l] = offset; next_code[l] = code;
        offset += count[l];
        code = (code + count[l]) << 1;
    }
    const uint32_t size = 1u << index_bits;
    for (uint32_t i = 0; i < size; ++i) table[i] = 0u;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens[s];
        if (!l) continue;
        c.sorted[offs[l]++] = uint16_t(s);
        const uint32_t cd = next_code[l]++;
        if (l <= index_bits) {
            const uint32_t rev = reverse15(cd << (15u - l));                // the code's bits in stream order, at the bottom
            const uint32_t e = is_dist ? dist_entry(s, l) : lit_entry(s, l);
            for (uint32_t idx = rev; idx < size; idx += 1u << l) table[idx] = e;
        }
    }
    return true;
}
// The symbol at the head of w the slow way; len = 0: no code starts so.
FQD_HD uint32_t slow_decode(const Code& c, uint64_t w, uint32_t& len)
{
    const uint32_t r = reverse15(uint32_t(w) & 0x7FFFu);
    uint32_t l = 1;
    for (uint32_t k = 1; k < 15u; ++k) l += r >= c.lim[k] ? 1u : 0u;
    if (r >= c.lim[15]) { len = 0; return 0; }
    len = l;
    return c.sorted[uint32_t(c.base[l] + int32_t(r >> (15u - l)))];
}

// The code lengths of a dynamic block whose three header bits have been read: pos -> first code of the block, lens[0..nlen)
// literal/length, lens[nlen..nlen+ndist) distance.  false: no such header (counts out of range, a code-length code that is
// not complete, lengths that overrun, no end-of-block code).  lens: 320 bytes.
FQD_HD_CALL bool read_code_lengths(const BitIn& in, uint64_t& pos, uint8_t* lens, uint32_t& nlen, uint32_t& ndist)
{
    uint64_t w = in.peek(pos);
    nlen = uint32_t(w & 31u) + 257u; ndist = uint32_t((w >> 5) & 31u) + 1u;
    const uint32_t ncode = uint32_t((w >> 10) & 15u) + 4u;
    pos += 14;
    if (nlen > 286u || ndist > 30u) return false;
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19];
    for (uint32_t i = 0; i < 19u; ++i) cl[i] = 0;
    w = in.peek(pos);
    for (uint32_t i = 0; i < ncode; ++i) {
        if (i == 16u) w = in.peek(pos);
        cl[order[i]] = uint8_t(w & 7u); w >>= 3; pos += 3;
    }
    // the code-length code: complete (zlib: type CODES), at most 7 bits: canonical decode without a table
    uint32_t count[8], first[8], offs[8]; uint8_t sorted[19];
    for (uint32_t l = 0; l < 8u; ++l) count[l] = 0;
    for (uint32_t i = 0; i < 19u; ++i) ++count[cl[i]];
    {
        int32_t left = 1;
        for (uint32_t l = 1; l <= 7u; ++l) { left = (left << 1) - int32_t(count[l]); if (left < 0) return false; }
        if (left != 0) return false;
    }
    {
        uint32_t code = 0, offset = 0;
        first[0] = 0; offs[0] = 0;
        for (uint32_t l = 1; l <= 7u; ++l) { first[l] = code; offs[l] = offset; offset += count[l]; code = (code + count[l]) << 1; }
        uint32_t at[8];
        for (uint32_t l = 0; l < 8u; ++l) at[l] = offs[l];
        for (uint32_t s = 0; s < 19u; ++s) if (cl[s]) sorted[at[cl[s]]++] = uint8_t(s);
    }
    const uint32_t total = nlen + ndist;
    uint32_t i = 0;
    while (i < total) {
        if (pos > in.nbits) return false;                                  // (garbage read as a header: no read beyond the 16 bytes behind the stream)
        w = in.peek(pos);
        uint32_t code = 0, l = 0, sym = 0xFFu;
        for (l = 1; l <= 7u; ++l) {                                       // a bit at a time, first bit the most significant of the code
            code = (code << 1) | uint32_t((w >> (l - 1u)) & 1u);
            if (count[l] && code >= first[l] && code - first[l] < count[l]) { sym = sorted[offs[l] + (code - first[l])]; break; }
        }
        if (sym == 0xFFu) return false;
        pos += l; w >>= l;
        if (sym < 16u) { lens[i++] = uint8_t(sym); continue; }
        uint32_t prev = 0, rep;
        if (sym == 16u) { if (i == 0u) return false; prev = lens[i - 1u]; rep = 3u + uint32_t(w & 3u); pos += 2; }
        else if (sym == 17u) { rep = 3u + uint32_t(w & 7u); pos += 3; }
        else { rep = 11u + uint32_t(w & 127u); pos += 7; }
        if (i + rep > total) return false;
        while (rep--) lens[i++] = uint8_t(prev);
    }
    return pos <= in.nbits && lens[256] != 0;
}
// Do these lengths make a complete code (or, where zlib lets it pass, none or a lone one-bit code)?
FQD_HD bool code_is_complete(const uint8_t* lens, uint32_t n, bool may_be_single)
{
    uint32_t count[16];
    for (uint32_t l = 0; l < 16u; ++l) count[l] = 0;
    for (uint32_t s = 0; s < n; ++s) ++count[lens[s]];
    int32_t left = 1; uint32_t codes = 0;
    for (uint32_t l = 1; l <= 15u; ++l) { left = (left << 1) - int32_t(count[l]); if (left < 0) return false; codes += count[l]; }
    return left == 0 || (may_be_single && (codes == 0u || (codes == 1u && count[1] == 1u)));
}
FQD_HD bool parse_dynamic_header(const BitIn& in, uint64_t& pos, Tables& t, uint8_t* lens)
{
    uint32_t nlen, ndist;
    if (!read_code_lengths(in, pos, lens, nlen, ndist)) return false;
    return build_code(lens, nlen, true, t.lit, kLitBits, t.lc, false) && build_code(lens + nlen, ndist, true, t.dist, kDistBits, t.dc, true);
}

FQD_HD_CALL void fixed_codes(Tables& t, uint8_t* lens)
{
    for (uint32_t s = 0; s < 288u; ++s) lens[s] = uint8_t(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u);
    (void)build_code(lens, 288u, false, t.lit, kLitBits, t.lc, false);
    for (uint32_t s = 0; s < 32u; ++s) lens[s] = 5;
    (void)build_code(lens, 32u, false, t.dist, kDistBits, t.dc, true);
}

// Is there a dynamic, non-final block header at stream bit `pos` whose codes are complete?  (What a unit's start is guessed by.)
// In two steps, because of 64 offsets tried side by side nearly all fail the first — a few instructions, nothing but registers —
// and the second wants 320 bytes of scratch: the three header bits, the counts, and a complete code-length code ...
// `bits(d)`: at least 57 bits of the stream from bit pos + d on (d = 0, 17, 65: the kernel answers out of five words it holds
// in registers and slides along, so that a turn of its loop waits for no load).
// (The very first part of it, on 13 bits: the kernel sifts every offset with this alone and looks further at what passes.)
FQD_HD bool block_start_header_bits(uint32_t w13)
{
    return (w13 & 7u) == 4u && ((w13 >> 3) & 31u) <= 29u && ((w13 >> 8) & 31u) <= 29u;     // BFINAL 0, BTYPE 2, HLIT <= 29 (286 codes), HDIST <= 29
}
template <class Bits>
FQD_HD bool block_start_first_look_bits(Bits&& bits)
{
    uint64_t w = bits(0u);
    if (!block_start_header_bits(uint32_t(w) & 0x1FFFu)) return false;
    const uint32_t ncode = uint32_t((w >> 13) & 15u) + 4u;
    // the code-length code is complete iff its lengths l > 0 add up to one in units of 2^-l (more: over-subscribed, less:
    // incomplete) — a sum, no table of counts: on the GPU an array indexed by data lives in scratch memory
    uint32_t kraft = 0;
    w = bits(17u);
    const uint64_t w2 = bits(17u + 48u);
    for (uint32_t i = 0; i < 19u; ++i) {                                  // (all nineteen, the ones beyond ncode not counted: no loop whose length is data)
        if (i == 16u) w = w2;
        const uint32_t l = uint32_t(w & 7u);
        kraft += (i < ncode && l) ? 128u >> l : 0u;
        w >>= 3;
    }
    return kraft == 128u;
}
FQD_HD bool block_start_first_look(const BitIn& in, uint64_t pos)
{
    if (pos + 300 > in.nbits) return false;                               // (a dynamic block's header alone is longer: 17 bits, >= 4 x 3, >= 257 lengths)
    return block_start_first_look_bits([&](uint32_t d) { return in.peek(pos + d); });
}
// ... then the lengths themselves and both codes.
FQD_HD bool block_start_second_look(const BitIn& in, uint64_t pos, uint8_t* lens)
{
    uint64_t p = pos + 3;
    uint32_t nlen, ndist;
    if (!read_code_lengths(in, p, lens, nlen, ndist)) return false;
    return code_is_complete(lens, nlen, true) && code_is_complete(lens + nlen, ndist, true);
}
FQD_HD bool plausible_block_start(const BitIn& in, uint64_t pos, uint8_t* lens)
{
    return block_start_first_look(in, pos) && block_start_second_look(in, pos, lens);
}

// ---- one unit decoded into symbols ------------------------------------------------------------------------------------
// The sink a decoder writes through (the GPU's is an LDS ring flushed by the whole wave, the CPU harness's an array):
//   bool room(uint32_t n)      at least n more symbols fit (258 are asked for before every code)
//   void put(uint16_t s)
//   void copy(uint32_t d, uint32_t len)   a match: len symbols, symbol k = the one at place count() - d + (k mod d) before the
//                              copy, or, where that place is negative, the window symbol 256 + kWindow + place (the GPU's
//                              sink lets every lane of the wave copy its share: the places read all exist beforehand)
//   uint64_t count()
// A decoder's state between calls (the GPU decodes a stretch, lets the wave flush, and goes on).
struct State {
    uint64_t pos = 0;                    // next bit to read
    uint64_t start_bit = 0;
    uint32_t in_block = 0;               // 1: inside a compressed block (tables valid), 2: inside a stored block, `stored_left` bytes to go
    uint32_t last = 0;                   // the block being decoded is the final one
    uint32_t stored_left = 0;
    uint32_t status = kOk;               // once != kOk the unit is done
    uint32_t deepest = 0;                // how far back into the window before the unit a match reached
};

// Decodes on from st until about `budget` symbols have been produced, the unit is done (st.status != kOk), or the sink is
// full.  stop_bit: the next unit's nominal start — the unit ends at the first block boundary at or after it where a
// dynamic, non-final block begins (kBoundary; st.pos = that boundary), or with the end of the final block (kFinal).
template <class Sink>
FQD_HD void decode_some(const BitIn& in, Tables& t, uint8_t* lens, State& st, uint64_t stop_bit, Sink& out, uint64_t budget)
{
    const uint64_t until = out.count() + budget;
    // every turn of the loops below produces a symbol, ends a block or ends the unit; `fuel` holds them to that whatever the data
    uint64_t fuel = 2u * budget + 4096u;
    while (st.status == kOk && out.count() < until) {
        if (fuel-- == 0u) { st.status = kBadData; return; }
        if (st.in_block == 0u) {
            if (st.pos + 3 > in.nbits) { st.status = kInputEnd; return; }
            const uint64_t w = in.peek(st.pos);
            const uint32_t last = uint32_t(w & 1u), type = uint32_t((w >> 1) & 3u);
            if (st.pos != st.start_bit && !last && type == 2u && st.pos >= stop_bit) { st.status = kBoundary; return; }
            st.pos += 3; st.last = last;
            if (type == 3u) { st.status = kBadData; return; }
            if (type == 0u) {
                st.pos = (st.pos + 7u) & ~uint64_t(7);
                if (st.pos + 32 > in.nbits) { st.status = kInputEnd; return; }
                const uint64_t h = in.peek(st.pos);
                const uint32_t n = uint32_t(h & 0xFFFFu), nn = uint32_t((h >> 16) & 0xFFFFu);
                if ((n ^ nn) != 0xFFFFu) { st.status = kBadData; return; }
                st.pos += 32;
                if (st.pos + uint64_t(n) * 8u > in.nbits) { st.status = kInputEnd; return; }
                st.stored_left = n; st.in_block = 2u;
            } else {
                if (type == 1u) fixed_codes(t, lens);
                else if (!parse_dynamic_header(in, st.pos, t, lens)) { st.status = kBadData; return; }
                st.in_block = 1u;
            }
        }
        if (st.in_block == 2u) {
            while (st.stored_left && out.count() < until) {
                if (fuel-- == 0u) { st.status = kBadData; return; }
                if (!out.room(8)) { st.status = kOutputFull; return; }
                uint64_t w = in.peek(st.pos);
                const uint32_t n = st.stored_left < 7u ? st.stored_left : 7u;      // (peek gives 57 bits)
                for (uint32_t k = 0; k < n; ++k) { out.put(uint16_t(w & 0xFFu)); w >>= 8; }
                st.pos += 8u * n; st.stored_left -= n;
            }
            if (st.stored_left == 0u) { st.in_block = 0u; if (st.last) { st.status = kFinal; return; } }
            continue;
        }
        // ---- a compressed block ------------------------------------------------------------------------------------------------
        // The bits in a register (`bb`, the low `bc` of them valid), topped up 32 at a time; the 64-bit word they come from is
        // loaded ONE WORD AHEAD of its use: on the GPU a load that the next code has to wait for costs a trip to L2 or HBM —
        // some thousand clocks against the hundred or two a code takes — and the first version of this loop (57 fresh bits
        // from memory whenever fewer than 48 were left, i.e. every code or two) ran at 2300 clocks per code.
        uint64_t pos = st.pos;
        const uint64_t p0 = pos + in.lead;
        uint64_t wi = p0 >> 6;                                            // word being drained
        uint64_t cur = in.words[wi], ahead = in.words[wi + 1];
        uint32_t half = uint32_t((p0 >> 5) & 1u);                         // next half of `cur` to take
        uint64_t bb; uint32_t bc;
        {
            const uint32_t s5 = uint32_t(p0 & 31u);                       // bits of the current half already consumed
            bb = (half ? cur >> 32 : cur & 0xFFFFFFFFull) >> s5; bc = 32u - s5;
            if (half) { cur = ahead; ++wi; ahead = in.words[wi + 1]; }
            half ^= 1u;
        }
        auto refill = [&]() {                                             // bc <= 32 -> bc + 32
            bb |= (half ? cur >> 32 : cur & 0xFFFFFFFFull) << bc; bc += 32u;
            if (half) { cur = ahead; ++wi; ahead = in.words[wi + 1]; }
            half ^= 1u;
        };
        bool end_of_block = false;
        uint32_t bad = kOk;
        while (out.count() < until) {
            if (fuel-- == 0u) { bad = kBadData; break; }
            if (bc <= 32u) {
                if (pos > in.nbits) { bad = kInputEnd; break; }           // (damaged or cut data: no read beyond the 16 bytes behind the stream)
                refill();
            }
            if (!out.room(258)) { bad = kOutputFull; break; }
            uint32_t e = t.lit[bb & ((1u << kLitBits) - 1u)];             // >= 33 bits are there: a code and its extra bits take 20 at most
            if ((e & 15u) == 0u) {
                uint32_t l;
                const uint32_t sy = slow_decode(t.lc, bb, l);
                if (l == 0u || sy > 285u) { bad = kBadData; break; }
                e = lit_entry(sy, l);
            }
            const uint32_t kind = (e >> 4) & 3u, cl = e & 15u;
            if (kind == 0u) { out.put(uint16_t(e >> 16)); bb >>= cl; bc -= cl; pos += cl; continue; }
            if (kind == 2u) { pos += cl; end_of_block = true; break; }
            const uint32_t ex = (e >> 8) & 31u;
            const uint32_t length = (e >> 16) + uint32_t((bb >> cl) & ((1u << ex) - 1u));
            bb >>= cl + ex; bc -= cl + ex; pos += cl + ex;
            if (bc <= 32u) refill();                                      // the distance: 15 + 13 bits at most
            uint32_t f = t.dist[bb & ((1u << kDistBits) - 1u)];
            if ((f & 15u) == 0u) {
                uint32_t l;
                const uint32_t ds = slow_decode(t.dc, bb, l);
                if (l == 0u || ds > 29u) { bad = kBadData; break; }
                f = dist_entry(ds, l);
            }
            const uint32_t dl = f & 15u, dx = (f >> 8) & 31u;
            const uint32_t d = (f >> 16) + uint32_t((bb >> dl) & ((1u << dx) - 1u));
            bb >>= dl + dx; bc -= dl + dx; pos += dl + dx;
            const uint64_t have = out.count();
            if (d > have) {
                const uint32_t before = uint32_t(d - have);               // the match starts this far before the unit's first byte
                if (before > kWindow) { bad = kBadData; break; }
                if (before > st.deepest) st.deepest = before;
            }
            out.copy(d, length);
        }
        st.pos = pos;
        if (bad != kOk) { st.status = bad; return; }
        if (pos > in.nbits) { st.status = kInputEnd; return; }
        if (end_of_block) { st.in_block = 0u; if (st.last) { st.status = kFinal; return; } }
    }
}

// A unit's window: the last kWindow bytes of the text up to and including the unit (a shorter text: its bytes at the END of
// the window, what lies before them is never read by a valid stream), from its symbols and the window before it.
//   prev: kWindow bytes; sym: the unit's n symbols; next: kWindow bytes written.  One element per call (k = 0 .. kWindow-1).
FQD_HD uint8_t window_byte(const uint8_t* prev, const uint16_t* sym, uint64_t n, uint32_t k)
{
    // position k of the new window is text position (n - kWindow + k) of the unit, or, before the unit, position (k + n) of the old window
    if (n >= kWindow || k + n >= kWindow) {
        const uint16_t s = sym[n - kWindow + k];                          // (n - kWindow + k >= 0 in both cases, as unsigned arithmetic has it)
        return s < 256u ? uint8_t(s) : prev[s - 256u];
    }
    return prev[k + n];
}

} // namespace gunz
} // namespace fqd
