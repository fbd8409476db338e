// pgzip.hpp — an ORDINARY gzip file (one long deflate stream, no member sizes: what gzip, pigz and sequencer software
// write) inflated by several threads.  Header-only: file_io.cpp includes it, nothing else does.
//
// The reference reads `.gz` inputs through Boost's gzip_decompressor on its one thread (file_utils.hpp:58-69), and so
// did this reader (zlib's gzread: 0.45-0.5 GB/s of text per file, 13 s of a 13.8 s run on 2 x 6.4 GB).  A deflate
// stream hides two things from a second thread: where a block starts, and the 32 KiB of text before it that its
// matches may reach into.  Both are dealt with the way pugz and rapidgzip do:
//
//   * the compressed bytes are cut into chunks; the worker of a chunk LOOKS for the first bit offset at or after its
//     chunk's start where a dynamic block header parses and both of its codes are complete — a guess, right but for
//     one time in many millions — and decodes from there to the first such block boundary at or after the next
//     chunk's start;
//   * what it decodes are 16-bit symbols: a byte, or "the byte that lies this far back in the 32 KiB before my
//     start" wherever a match reaches there;
//   * the reader walks the chunks in order: a chunk counts only if it starts exactly where the chunk before it
//     ended (a chain of block boundaries that begins at the stream's true start — a wrong guess breaks the chain
//     and the stretch is decoded again from the right place, on the spot); then the window it needed is known, its
//     symbols become bytes (a worker's job again, with the CRC-32 of the piece), and the pieces' CRCs are combined
//     and held against the member's trailer, ISIZE too, as zlib would.
//
// Anything this cannot do (no regular file, a header field it does not know) leaves the file to gzread as before.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <zlib.h>

namespace fqdhost {
namespace pgz {

// ---- bits ----------------------------------------------------------------------------------------------------------
struct BitIn {
    const uint8_t* base = nullptr;
    uint64_t nbytes = 0;
    // at least 56 valid bits of the stream from bit `pos` on (zeros beyond its end)
    uint64_t window(uint64_t pos) const
    {
        const uint64_t byte = pos >> 3;
        uint64_t v = 0;
        if (byte + 8 <= nbytes) std::memcpy(&v, base + byte, 8);
        else for (uint64_t k = 0; byte + k < nbytes && k < 8; ++k) v |= uint64_t(base[byte + k]) << (8 * k);
        return v >> (pos & 7);
    }
    uint64_t bits() const { return nbytes * 8; }
};

// ---- one canonical Huffman code ------------------------------------------------------------------------------------
struct Code {
    static constexpr int kLut = 11;
    uint16_t lut[1 << kLut];            // symbol << 4 | length; 0: a longer code, or none
    uint32_t lim[16];                   // [l]: end of the codes of length <= l, left-justified in 15 bits
    int32_t base[16];
    uint16_t sorted[288];

    // false: over-subscribed, or incomplete where zlib does not let that pass
    bool build(const uint8_t* lens, int n, bool may_be_single)
    {
        int count[16] = {0};
        for (int s = 0; s < n; ++s) ++count[lens[s]];
        int left = 1, codes = 0;
        for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; codes += count[l]; }
        if (left > 0 && !(may_be_single && (codes == 0 || (codes == 1 && count[1] == 1)))) return false;
        uint32_t code = 0, offset = 0, offs[16] = {0};
        for (int l = 1; l <= 15; ++l) {
            base[l] = int32_t(offset) - int32_t(code);
            lim[l] = (code + uint32_t(count[l])) << (15 - l);
            offs[l] = offset;
            offset += uint32_t(count[l]);
            code = (code + uint32_t(count[l])) << 1;
        }
        std::memset(lut, 0, sizeof lut);
        uint32_t next_code[16];
        { uint32_t c = 0; for (int l = 1; l <= 15; ++l) { next_code[l] = c; c = (c + uint32_t(count[l])) << 1; } }
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            sorted[offs[l]++] = uint16_t(s);
            const uint32_t c = next_code[l]++;
            if (l <= kLut) {
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((c >> b) & 1u) << (l - 1 - b);
                for (uint32_t idx = rev; idx < (1u << kLut); idx += 1u << l) lut[idx] = uint16_t((s << 4) | l);
            }
        }
        return true;
    }
    // the symbol at the head of w (stream order, least significant bit first); -1: no code starts so
    inline int decode(uint64_t w, int& len) const
    {
        const uint32_t e = lut[w & ((1u << kLut) - 1)];
        if (e) { len = int(e & 15); return int(e >> 4); }
        uint32_t v = uint32_t(w) & 0x7FFFu, r = 0;                  // 15 bits, reversed: the first bit on top
        for (int b = 0; b < 15; ++b) r |= ((v >> b) & 1u) << (14 - b);
        int l = 1;
        for (int k = 1; k < 15; ++k) l += r >= lim[k] ? 1 : 0;
        if (r >= lim[15]) return -1;
        len = l;
        return sorted[uint32_t(base[l] + int32_t(r >> (15 - l)))];
    }
};

inline bool parse_dynamic_header(const BitIn& in, uint64_t& pos, Code& lit, Code& dist)
{
    uint64_t w = in.window(pos);
    const int nlen = int(w & 31) + 257, ndist = int((w >> 5) & 31) + 1, ncode = int((w >> 10) & 15) + 4;
    pos += 14;
    if (nlen > 286 || ndist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    w = in.window(pos);                                             // 19 x 3 = 57 bits at most: two windows
    for (int i = 0; i < ncode; ++i) {
        if (i == 16) w = in.window(pos);
        cl[order[i]] = uint8_t(w & 7); w >>= 3; pos += 3;
    }
    {   // the code-length code must be complete (zlib: type CODES), checked before a table is made of it
        int left = 1, count[8] = {0};
        for (int i = 0; i < 19; ++i) ++count[cl[i]];
        for (int l = 1; l <= 7; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
        if (left != 0) return false;
    }
    Code clc;
    if (!clc.build(cl, 19, false)) return false;
    uint8_t lens[320];
    int i = 0;
    const int total = nlen + ndist;
    while (i < total) {
        w = in.window(pos);
        int l;
        const int sym = clc.decode(w, l);
        if (sym < 0) return false;
        pos += uint64_t(l); w >>= l;
        if (sym < 16) { lens[i++] = uint8_t(sym); continue; }
        int prev = 0, rep;
        if (sym == 16) { if (i == 0) return false; prev = lens[i - 1]; rep = 3 + int(w & 3); pos += 2; }
        else if (sym == 17) { rep = 3 + int(w & 7); pos += 3; }
        else { rep = 11 + int(w & 127); pos += 7; }
        if (i + rep > total) return false;
        while (rep--) lens[i++] = uint8_t(prev);
    }
    if (pos > in.bits() || lens[256] == 0) return false;
    return lit.build(lens, nlen, true) && dist.build(lens + nlen, ndist, true);     // (zlib lets a lone one-bit code pass in either)
}

// What the decode loop reads per code: everything about it in one 32-bit word.
//   bits 0-3 code length (0: longer than the table's index, or no code — the slow way decides), bits 4-5 kind
//   (0 literal, 1 length, 2 end of block, 3 distance), bits 8-12 extra bits, bits 16-31 the literal / the base.
inline uint32_t lit_entry(uint32_t sym, uint32_t len)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (sym < 256u) return len | (sym << 16);
    if (sym == 256u) return len | (2u << 4);
    if (sym <= 285u) return len | (1u << 4) | (uint32_t(lext[sym - 257u]) << 8) | (uint32_t(lbase[sym - 257u]) << 16);
    return 0;                                                        // (286, 287: no entry — the slow way reports them)
}
inline uint32_t dist_entry(uint32_t ds, uint32_t len)
{
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    return ds < 30u ? len | (3u << 4) | (uint32_t(dext[ds]) << 8) | (uint32_t(dbase[ds]) << 16) : 0u;
}
struct Fast {
    uint32_t lit[1 << Code::kLut], dist[1 << Code::kLut];
    void build(const Code& l, const Code& d)
    {
        for (uint32_t i = 0; i < (1u << Code::kLut); ++i) {
            const uint32_t e = l.lut[i], f = d.lut[i];
            lit[i] = (e & 15u) ? lit_entry(e >> 4, e & 15u) : 0u;
            dist[i] = (f & 15u) ? dist_entry(f >> 4, f & 15u) : 0u;
        }
    }
};

inline void fixed_codes(Code& lit, Code& dist)
{
    uint8_t l[288], d[32];
    for (int s = 0; s < 288; ++s) l[s] = uint8_t(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
    for (int s = 0; s < 32; ++s) d[s] = 5;
    lit.build(l, 288, false); dist.build(d, 32, false);
}

// ---- a stretch of the stream decoded into 16-bit symbols -------------------------------------------------------------
constexpr uint32_t kWindow = 32768;
constexpr size_t kMaxPiece = size_t(24) << 20;          // symbols of a piece before it ends at the next block boundary
enum class Stop { Boundary, FinalBlock, Error };

struct Piece {
    uint64_t start_bit = UINT64_MAX, end_bit = 0;   // start: UINT64_MAX = no block start found in the chunk
    Stop stop = Stop::Error;
    std::vector<uint16_t> sym;                      // < 256: a byte; else 256 + place in the 32 KiB before the piece's first byte
    uint32_t deepest = 0;                           // how far back into that window the piece reaches (0: not at all)
};

// Decodes blocks from `pos` (a block boundary) to the first boundary at or after `stop_bit` where a dynamic, non-final
// block begins — the kind a chunk's worker looks for — or through the end of a final block.
inline void decode_piece(const BitIn& in, uint64_t pos, uint64_t stop_bit, Piece& out)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.start_bit = pos; out.stop = Stop::Error; out.deepest = 0;
    std::vector<uint16_t>& sym = out.sym;
    const uint64_t total = in.bits();
    auto codes = std::make_unique<std::pair<Code, Code>>();
    Code& lit = codes->first; Code& dist = codes->second;
    auto fast = std::make_unique<Fast>();
    for (;;) {
        if (pos + 3 > total) return;
        uint64_t w = in.window(pos);
        const int last = int(w & 1), type = int((w >> 1) & 3);
        // (a piece also ends where it has grown large — text that packs a thousandfold would otherwise be held whole, twice)
        //  — at whatever kind of block: a file of stored blocks has no other)
        if (pos != out.start_bit && ((!last && type == 2 && pos >= stop_bit) || sym.size() >= kMaxPiece)) { out.end_bit = pos; out.stop = Stop::Boundary; return; }
        pos += 3;
        if (type == 3) return;
        if (type == 0) {
            pos = (pos + 7) & ~uint64_t(7);
            if (pos + 32 > total) return;
            const uint64_t h = in.window(pos);
            const uint32_t n = uint32_t(h & 0xFFFF), nn = uint32_t((h >> 16) & 0xFFFF);
            if ((n ^ nn) != 0xFFFF) return;
            pos += 32;
            if (pos + uint64_t(n) * 8 > total) return;
            const uint8_t* p = in.base + (pos >> 3);
            const size_t at = sym.size();
            sym.resize(at + n);
            for (uint32_t k = 0; k < n; ++k) sym[at + k] = p[k];
            pos += uint64_t(n) * 8;
        } else {
            if (type == 1) fixed_codes(lit, dist);
            else if (!parse_dynamic_header(in, pos, lit, dist)) return;
            // (the symbols go through a bare pointer into the vector's spare room: a push_back per literal is what this
            //  loop would otherwise mostly do)
            size_t n = sym.size();
            auto room = [&](size_t need) { if (sym.size() < n + need) sym.resize(n + need + 65536); };     // (a step at a time: the vector is cut back to n at every block's end, and growing fills with zeros)
            bool at_end_of_block = false;
            if ((pos >> 3) + 64 < in.nbytes) {
                // The fast way, while the stream's end is far: the bits in a register, refilled eight bytes at a load;
                // everything about a code in one table word.  What it cannot settle (a code longer than the table's
                // index, the last bytes of the stream) is left to the loop below, from the very bit it stopped at.
                fast->build(lit, dist);
                const uint8_t* p = in.base + (pos >> 3);
                const uint8_t* const safe = in.base + in.nbytes - 16;
                uint64_t bb = 0;
                uint32_t bc = 0;
                // bc (56..63 after a refill) bits of bb count; the whole bytes among them lie before p, and what bb holds
                // above them is a copy of what the next refill ORs in again
                auto refill = [&]() { uint64_t v; std::memcpy(&v, p, 8); bb |= v << bc; p += (63u - bc) >> 3; bc |= 56u; };
                refill();
                bb >>= (pos & 7); bc -= uint32_t(pos & 7);
                constexpr uint32_t kMask = (1u << Code::kLut) - 1u;
                for (;;) {
                    if (p >= safe) break;
                    room(300);
                    uint16_t* const q0 = sym.data();
                    refill();
                    uint32_t e = fast->lit[bb & kMask];
                    if ((e & 15u) == 0u) {                            // a code longer than the table's index: the slow way, for this one code
                        int l;
                        const int sl = lit.decode(bb, l);
                        if (sl < 0 || sl > 285) break;                // (nothing consumed: the loop below reports it)
                        e = lit_entry(uint32_t(sl), uint32_t(l));
                    }
                    if ((e & 0x30u) == 0u) {                          // literals, up to three from one fill
                        bb >>= (e & 15u); bc -= (e & 15u); q0[n++] = uint16_t(e >> 16);
                        e = fast->lit[bb & kMask];
                        if ((e & 15u) != 0u && (e & 0x30u) == 0u) {
                            bb >>= (e & 15u); bc -= (e & 15u); q0[n++] = uint16_t(e >> 16);
                            e = fast->lit[bb & kMask];
                            if ((e & 15u) != 0u && (e & 0x30u) == 0u) { bb >>= (e & 15u); bc -= (e & 15u); q0[n++] = uint16_t(e >> 16); }
                        }
                        continue;
                    }
                    if ((e & 0x30u) == 0x20u) { bb >>= (e & 15u); bc -= (e & 15u); at_end_of_block = true; break; }
                    // a length: its code and extra bits, then the distance's (48 bits at most, 56 are there)
                    const uint32_t ex = (e >> 8) & 31u;
                    const uint64_t after = bb >> ((e & 15u) + ex);
                    uint32_t f = fast->dist[after & kMask];
                    if ((f & 15u) == 0u) {
                        int dl;
                        const int ds = dist.decode(after, dl);
                        if (ds < 0 || ds > 29) break;                 // (nothing consumed yet)
                        f = dist_entry(uint32_t(ds), uint32_t(dl));
                    }
                    bb >>= (e & 15u);
                    const uint32_t length = (e >> 16) + uint32_t(bb & ((1u << ex) - 1u));
                    bb = after >> (f & 15u);
                    const uint32_t dx = (f >> 8) & 31u;
                    const uint32_t d = (f >> 16) + uint32_t(bb & ((1u << dx) - 1u));
                    bb >>= dx;
                    bc -= (e & 15u) + ex + (f & 15u) + dx;
                    uint16_t* q = q0 + n;
                    if (d <= n) {
                        const uint16_t* from = q - d;
                        if (d >= 4u) {                                // four symbols a step (the copy may run up to three past the match: room(300) covers it,
                            for (uint32_t k = 0; k < length; k += 4u) std::memcpy(q + k, from + k, 8);       // and they are overwritten by what comes next)
                        } else for (uint32_t k = 0; k < length; ++k) q[k] = from[k];
                    } else {
                        const uint32_t before = uint32_t(d - n);
                        if (before > kWindow) { sym.resize(n); return; }
                        if (before > out.deepest) out.deepest = before;
                        for (uint32_t k = 0; k < length; ++k) {
                            if (k < before) q[k] = uint16_t(256 + (kWindow - before + k));
                            else q[k] = q[int64_t(k) - int64_t(d)];
                        }
                    }
                    n += length;
                }
                pos = uint64_t(p - in.base) * 8u - bc;
            }
            for (;;) {
                if (at_end_of_block) break;
                if (pos >= total) { sym.resize(n); return; }          // the stream ends inside a block
                room(300);
                uint16_t* const q0 = sym.data();
                w = in.window(pos);
                int l;
                int s = lit.decode(w, l);
                if (s < 0) { sym.resize(n); return; }
                pos += uint64_t(l); w >>= l;
                if (pos > total) { sym.resize(n); return; }           // (a code read out of the zeros beyond the stream's last byte)
                if (s < 256) {
                    q0[n++] = uint16_t(s);
                    // a second literal from the same window, more often than not
                    s = lit.decode(w, l);
                    if (s >= 0 && s < 256 && pos + uint64_t(l) <= total) { q0[n++] = uint16_t(s); pos += uint64_t(l); }
                    continue;
                }
                if (s == 256) break;
                if (s > 285) { sym.resize(n); return; }
                s -= 257;
                const uint32_t length = lbase[s] + uint32_t(w & ((1u << lext[s]) - 1u));
                pos += lext[s]; w >>= lext[s];
                int dl;
                const int ds = dist.decode(w, dl);
                if (ds < 0 || ds > 29) { sym.resize(n); return; }
                pos += uint64_t(dl); w >>= dl;
                const uint32_t d = dbase[ds] + uint32_t(w & ((1u << dext[ds]) - 1u));
                pos += dext[ds];
                uint16_t* q = q0 + n;
                if (d <= n) {
                    const uint16_t* from = q - d;
                    for (uint32_t k = 0; k < length; ++k) q[k] = from[k];
                } else {                                            // (part of) it lies before the piece: named, not known
                    const uint32_t before = uint32_t(d - n);        // bytes back from the piece's first byte, of the match's first byte
                    if (before > kWindow) { sym.resize(n); return; }
                    if (before > out.deepest) out.deepest = before;
                    for (uint32_t k = 0; k < length; ++k) {
                        if (k < before) q[k] = uint16_t(256 + (kWindow - before + k));
                        else q[k] = q[int64_t(k) - int64_t(d)];
                    }
                }
                n += length;
            }
            sym.resize(n);
        }
        if (last) { out.end_bit = pos; out.stop = Stop::FinalBlock; return; }
    }
}

// The first bit offset in [from, to) where a dynamic non-final block header parses with complete codes; UINT64_MAX: none.
inline uint64_t find_block(const BitIn& in, uint64_t from, uint64_t to)
{
    auto codes = std::make_unique<std::pair<Code, Code>>();
    const uint64_t total = in.bits();
    for (uint64_t o = from; o < to && o + 17 <= total; ++o) {
        const uint64_t w = in.window(o);
        if ((w & 7) != 4) continue;                                  // BFINAL = 0, BTYPE = 2
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        uint64_t pos = o + 3;
        if (parse_dynamic_header(in, pos, codes->first, codes->second)) return o;
    }
    return UINT64_MAX;
}

// ---- symbols to bytes -------------------------------------------------------------------------------------------------
// `window` = the (up to 32 KiB of) text before the piece's first byte.  false: the piece reaches further back than that.
inline bool resolve(const std::vector<uint16_t>& sym, size_t from, size_t to, const std::vector<uint8_t>& window, uint8_t* out)
{
    const size_t w = window.size();
    for (size_t k = from; k < to; ++k) {
        const uint16_t s = sym[k];
        if (s < 256) { out[k - from] = uint8_t(s); continue; }
        const uint32_t back = kWindow - (uint32_t(s) - 256u);       // bytes back from the piece's first byte
        if (back > w) return false;
        out[k - from] = window[w - back];
    }
    return true;
}

struct Ready { std::vector<uint8_t> bytes; size_t size = 0; uint32_t crc = 0; bool ok = true; };   // (bytes may be longer than size: a recycled buffer)

// Buffers go round: a piece's symbols and a run's bytes are megabytes, and fresh megabytes are pages the kernel hands
// out one fault at a time, behind one lock for all threads of the process.
struct Pool {
    std::mutex m;
    std::vector<std::vector<uint16_t>> syms;
    std::vector<std::vector<uint8_t>> bytes;
    std::vector<uint16_t> take_syms()
    {
        std::lock_guard<std::mutex> g(m);
        if (syms.empty()) { std::vector<uint16_t> v; v.reserve(size_t(6) << 20); return v; }
        std::vector<uint16_t> v = std::move(syms.back()); syms.pop_back(); v.clear(); return v;
    }
    void give(std::vector<uint16_t>&& v) { std::lock_guard<std::mutex> g(m); if (syms.size() < 32 && v.capacity() <= (size_t(16) << 20)) syms.push_back(std::move(v)); }
    std::vector<uint8_t> take_bytes(size_t n)
    {
        std::vector<uint8_t> v;
        { std::lock_guard<std::mutex> g(m); if (!bytes.empty()) { v = std::move(bytes.back()); bytes.pop_back(); } }
        if (v.size() < n) v.resize(n + n / 8);
        return v;
    }
    void give(std::vector<uint8_t>&& v) { std::lock_guard<std::mutex> g(m); if (bytes.size() < 32 && v.capacity() <= (size_t(16) << 20)) bytes.push_back(std::move(v)); }
};

// ---- the reader ---------------------------------------------------------------------------------------------------------
class Reader {
public:
    // throws std::invalid_argument where this reader does not apply (the caller reads the file with zlib then)
    // crc: CRC-32 of a buffer continued from a value, as zlib's crc32 (nullptr: zlib's; libdeflate's is several times faster)
    using CrcFn = uint32_t (*)(uint32_t, const void*, size_t);
    Reader(int fd, uint64_t size, unsigned threads, CrcFn crc = nullptr) : size_(size), threads_(std::min(most_threads(), std::max(2u, threads))), crc_(crc)
    {
        void* m = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) throw std::invalid_argument("mmap");
        in_.base = static_cast<const uint8_t*>(m); in_.nbytes = size;
#ifdef MADV_SEQUENTIAL
        (void)::madvise(m, size, MADV_SEQUENTIAL);
#endif
        uint64_t at = 0;
        if (!member_header(at)) { ::munmap(m, size); throw std::invalid_argument("header"); }
        cur_bit_ = at * 8;
        next_chunk_ = cur_bit_ / kChunkBits + 1;
        first_of_member_ = true;
    }
    ~Reader()
    {
        for (auto& f : ahead_) if (f.second.valid()) f.second.wait();
        for (auto& f : ready_) if (f.valid()) f.wait();
        ::munmap(const_cast<uint8_t*>(in_.base), size_);
    }
    // pieces decoded ahead (and as many being turned into bytes): FQD_PGZIP_THREADS, 8 unless told otherwise — two files read
    // side by side then fill the 16 cores a GPU box gives a job
    static unsigned most_threads()
    {
        static const unsigned n = [] { const char* v = std::getenv("FQD_PGZIP_THREADS"); const int x = v ? std::atoi(v) : 0; return x >= 2 ? unsigned(x) : 8u; }();
        return n;
    }
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;

    size_t read(char* dst, size_t n)
    {
        size_t got = 0;
        while (got < n) {
            if (have_pos_ < have_.size) {
                const size_t k = std::min(n - got, have_.size - have_pos_);
                copy_out(dst + got, have_.bytes.data() + have_pos_, k);
                have_pos_ += k; got += k;
                continue;
            }
            if (!next_ready()) break;
        }
        return got;
    }

private:
    // The text leaves through this one thread: megabytes at a time are copied by four (one core moves 8-10 GB/s, which
    // is what eight decoding threads make).
    static void copy_out(char* dst, const uint8_t* src, size_t n)
    {
        if (n < (size_t(2) << 20)) { std::memcpy(dst, src, n); return; }
        constexpr size_t T = 4;
        std::thread helper[T - 1];
        for (size_t t = 1; t < T; ++t) helper[t - 1] = std::thread([=] { const size_t a = n / T * t, b = t + 1 == T ? n : n / T * (t + 1); std::memcpy(dst + a, src + a, b - a); });
        std::memcpy(dst, src, n / T);
        for (std::thread& h : helper) h.join();
    }
    static constexpr uint64_t kChunkBits = uint64_t(1) << 23;       // 1 MiB of compressed bytes per chunk

    [[noreturn]] static void corrupt() { throw std::runtime_error("gzip input is corrupt or truncated"); }

    // the gzip member header at byte `at`; true: at = first byte of its deflate stream
    bool member_header(uint64_t& at) const
    {
        const uint8_t* p = in_.base;
        if (at + 18 > size_ || p[at] != 31 || p[at + 1] != 139 || p[at + 2] != 8) return false;
        const uint8_t flg = p[at + 3];
        if (flg & 0xE0) return false;
        uint64_t q = at + 10;
        if (flg & 4) { if (q + 2 > size_) return false; q += 2 + (uint64_t(p[q]) | (uint64_t(p[q + 1]) << 8)); }
        if (flg & 8) { while (q < size_ && p[q]) ++q; ++q; }
        if (flg & 16) { while (q < size_ && p[q]) ++q; ++q; }
        if (flg & 2) q += 2;
        if (q + 8 > size_) return false;
        at = q;
        return true;
    }

    std::future<Piece> launch(uint64_t chunk)
    {
        return std::async(std::launch::async, [this, chunk] {
            Piece p;
            const uint64_t lo = chunk * kChunkBits, hi = lo + kChunkBits;
            const uint64_t s = find_block(in_, lo, hi);
            if (s == UINT64_MAX) return p;
            p.sym = pool_.take_syms();
            decode_piece(in_, s, hi, p);
            return p;
        });
    }

    void top_up()
    {
        while (ahead_.size() < threads_ && next_chunk_ * kChunkBits < in_.bits()) {
            ahead_.emplace_back(next_chunk_, launch(next_chunk_));
            ++next_chunk_;
        }
    }

    // The piece that starts at cur_bit_: the worker's, if its guess was that very bit, else decoded here and now.
    Piece take_piece()
    {
        top_up();
        // chunks whose range lies behind cur_bit_ are of no use any more (a piece ran through them)
        while (!ahead_.empty() && (ahead_.front().first + 1) * kChunkBits <= cur_bit_) { pool_.give(std::move(ahead_.front().second.get().sym)); ahead_.pop_front(); top_up(); }
        if (!ahead_.empty() && ahead_.front().first * kChunkBits <= cur_bit_) {
            Piece p = ahead_.front().second.get();
            ahead_.pop_front();
            top_up();
            if (p.start_bit == cur_bit_ && p.stop != Stop::Error) return p;
            pool_.give(std::move(p.sym));
            // a wrong guess, a chunk without a block start, or a stretch the worker could not decode: from the right place
        }
        Piece p;
        p.sym = pool_.take_syms();
        const uint64_t stop = (cur_bit_ / kChunkBits + 1) * kChunkBits;
        decode_piece(in_, cur_bit_, stop, p);
        return p;
    }

    // Brings the next run of text into have_; false at the end of the file.
    bool next_ready()
    {
        // keep a few pieces being turned into bytes while the oldest is handed out
        while (!done_ && ready_.size() < threads_) step();
        if (ready_.empty()) { if (failed_) corrupt(); return false; }
        pool_.give(std::move(have_.bytes));
        have_ = ready_.front().get();
        ready_.pop_front();
        have_pos_ = 0;
        if (!have_.ok) corrupt();
        return true;
    }

    // One piece further along the chain: window known, bytes and CRC left to a worker, member ends checked.
    // (damage found here is reported by next_ready once the text before it has been handed out, as zlib would)
    void fail() { failed_ = true; done_ = true; }
    void step()
    {
        if (failed_) { done_ = true; return; }
        auto piece = std::make_shared<Piece>(take_piece());
        if (piece->deepest > window_.size()) { fail(); return; }    // a match that reaches before the member's first byte
        if (piece->stop == Stop::Error) {                           // what was decoded before the damage is still text: handed out first,
            failed_ = true;                                         // as zlib hands out what it has before it reports the error
            done_ = true;
            if (piece->sym.empty()) return;
        }
        auto window = std::make_shared<std::vector<uint8_t>>(window_);
        // the window after this piece: its last 32 KiB, resolved here (the next piece cannot start without it)
        {
            const size_t n = piece->sym.size();
            std::vector<uint8_t> tail(std::min<size_t>(n, kWindow));
            if (!resolve(piece->sym, n - tail.size(), n, window_, tail.data())) { fail(); return; }
            if (tail.size() < kWindow) {
                const size_t keep = std::min<size_t>(window_.size(), kWindow - tail.size());
                std::vector<uint8_t> w(window_.end() - static_cast<ptrdiff_t>(keep), window_.end());
                w.insert(w.end(), tail.begin(), tail.end());
                window_.swap(w);
            } else window_.swap(tail);
        }
        member_bytes_ += piece->sym.size();
        const bool ends_member = piece->stop == Stop::FinalBlock;
        if (failed_) {
            ready_.push_back(std::async(std::launch::async, [this, piece, window] {
                Ready r;
                r.size = piece->sym.size();
                r.bytes = pool_.take_bytes(r.size);
                r.ok = resolve(piece->sym, 0, piece->sym.size(), *window, r.bytes.data());
                pool_.give(std::move(piece->sym));
                return r;
            }));
            return;
        }
        uint32_t want_crc = 0;
        bool check = false;
        cur_bit_ = piece->end_bit;
        if (ends_member) {
            uint64_t at = (cur_bit_ + 7) / 8;
            if (at + 8 > size_) { fail(); return; }
            const uint8_t* t = in_.base + at;
            want_crc = uint32_t(t[0]) | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
            const uint32_t isize = uint32_t(t[4]) | (uint32_t(t[5]) << 8) | (uint32_t(t[6]) << 16) | (uint32_t(t[7]) << 24);
            if (isize != uint32_t(member_bytes_)) { fail(); return; }
            check = true;
            at += 8;
            // what follows: another member, zero padding, or the end
            while (at < size_ && in_.base[at] == 0) ++at;
            if (at >= size_ || !member_header(at)) done_ = true;      // (what follows a member and is no member: ignored, as zlib does)
            else {
                cur_bit_ = at * 8;
                window_.clear(); member_bytes_ = 0;
                // workers ahead of the new member's start guessed inside the old one or across the seam: the chain decides
            }
        } else if (cur_bit_ >= in_.bits()) { fail(); return; }       // the stream ends without a final block
        // bytes + CRC of the piece on a worker; the CRCs of a member's pieces are combined in order as they are handed out
        const bool first = first_of_member_;
        first_of_member_ = ends_member;
        auto crc_so_far = crc_chain_;
        auto mine = std::make_shared<std::promise<uint32_t>>();
        crc_chain_ = mine->get_future().share();
        const CrcFn crc_fn = crc_;
        ready_.push_back(std::async(std::launch::async, [this, piece, window, first, crc_so_far, mine, check, want_crc, crc_fn] {
            Ready r;
            r.size = piece->sym.size();
            r.bytes = pool_.take_bytes(r.size);
            r.ok = resolve(piece->sym, 0, piece->sym.size(), *window, r.bytes.data());
            pool_.give(std::move(piece->sym));
            uint32_t c = crc_fn ? crc_fn(0u, r.bytes.data(), r.size) : uint32_t(crc32_z(0L, r.bytes.data(), r.size));
            if (!first) c = uint32_t(crc32_combine(crc_so_far.get(), c, static_cast<z_off_t>(r.size)));
            mine->set_value(c);
            if (check && c != want_crc) r.ok = false;
            return r;
        }));
    }

    BitIn in_;
    Pool pool_;
    uint64_t size_;
    unsigned threads_;
    CrcFn crc_ = nullptr;
    uint64_t cur_bit_ = 0, next_chunk_ = 0, member_bytes_ = 0;
    bool done_ = false, first_of_member_ = true, failed_ = false;
    std::vector<uint8_t> window_;
    std::deque<std::pair<uint64_t, std::future<Piece>>> ahead_;
    std::deque<std::future<Ready>> ready_;
    std::shared_future<uint32_t> crc_chain_;
    Ready have_;
    size_t have_pos_ = 0;
};

} // namespace pgz
} // namespace fqdhost
