// fqd_inflate_wave.hpp — one raw deflate stream (RFC 1951) decoded by ONE WAVE: the part of the GPU BGZF reader
// that does not care where it runs.
//
// The reference reads `.gz` inputs through Boost's gzip_decompressor, one thread per file (file_utils.hpp:58-69).
// A BGZF file is a sequence of independent members of at most 64 KiB; here a member is the work of one 64-lane
// wave, and what a deflate block hides from a parallel reader — where its codes start — is found the way
// self-synchronising Huffman decoders find it:
//
//   1. the block's bits are cut into one subsequence per lane; every lane decodes from where it GUESSES a code
//      starts, to the first code boundary at or beyond its subsequence's end, and counts what it saw (bytes, matches);
//   2. a lane's end is the next lane's true start: lanes whose start moved decode again; prefix codes fall into
//      step within a few codes, so the ends hardly move and the second round is the last but for stragglers
//      (the chain is exact after at most as many rounds as there are lanes, whatever the data);
//   3. byte and match counts of the lanes up to the one that met the end-of-block code are summed up: every lane
//      now knows where its output goes, decodes once more and writes its literals; matches become tokens
//      (where, how long, how far back) in stream order;
//   4. the tokens are resolved 64 at a time: a match whose source holds no unresolved match of the same group is
//      copied at once, the others as soon as the ones they wait for are done.
//
// The code is written over a context `Ctx` that runs a phase for every lane and then makes its effects visible
// (`lanes`), and asks all lanes a yes/no question (`ballot`): on the GPU that is a wave and its LDS
// (fqd_inflate.hip); in tests/native/inflate_wave_check.cpp it is a loop over the lanes, which runs this very
// code on the CPU against zlib, under the sanitizers.  Everything a lane keeps from one phase to the next lives
// in `Shared` (LDS); values that steer the phases are the same in every lane (`ctx.same`).  Phases that touch
// nothing of each other's may run into one another (`lanes_open`, then one `sync`).  `ctx.mark(k)` is where a
// diagnostic build reads the clock (phase k ends here); it does nothing otherwise.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define FQD_HD __host__ __device__ __forceinline__
#else
#define FQD_HD inline
#endif

namespace fqd {
namespace winf {

constexpr uint32_t kMaxBits = 15;
constexpr uint32_t kLitBits = 10, kDistBits = 9;        // index bits of the two first-level tables
constexpr uint32_t kLitSymbols = 288, kDistSymbols = 30;
constexpr uint32_t kMinSub = 256, kMaxSub = 2048;       // bits of a lane's subsequence
constexpr uint32_t kTokenRoom = 12288;                  // matches of one window (a window that holds more is cut short)

enum Status : uint32_t { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadLengths = 3, kBadCode = 4, kBadDistance = 5,
                         kOutputOverrun = 6, kInputOverrun = 7, kShortOutput = 8 };
// What a lane met on its way: the end-of-block code stops it; bits that are no code do not — it steps over one bit and
// goes on, because a lane that started from a guess reads such bits as a matter of course and should still fall
// into step, so that the lane after it learns its true start in this round and not in a later one.  (In a lane whose
// start is exact, kBroken means a damaged member.)
enum : uint32_t { kEndOfBlock = 1u, kBroken = 2u, kOffTheEnd = 4u };

struct Token { uint32_t dst; uint32_t len_dist; };      // len_dist = length << 16 | distance

#if defined(FQD_WINF_STATS)                             // (the CPU harness counts what the phases did)
struct Stats { unsigned long long blocks, windows, rounds, lane_decodes, symbols_hint, groups, group_rounds, tokens, cut, dep_one_periodic, dep_one_partial, dep_many; };
inline Stats& stats() { static Stats s{}; return s; }
#define FQD_WINF_COUNT(field, n) (stats().field += (n))
#else
#define FQD_WINF_COUNT(field, n) ((void)0)
#endif

// What the next kLitBits bits of a block decode to, at one LDS read: up to three literals in a row (DNA and quality
// codes are two to five bits long), or one length / end-of-block symbol.
//   bits 0-3   bits used (0: the first code is longer than kLitBits, or no code: the slow way finds out)
//   bits 4-5   literals (0: bits 8-16 hold a symbol >= 256)
//   bits 8-31  the literals, first one lowest
FQD_HD uint32_t pack_symbol(uint32_t sym, uint32_t len) { return len | (sym < 256u ? 1u << 4 : 0u) | (sym << 8); }

template <uint32_t L>
struct PerLane {
    uint32_t start[L], end[L], nbytes[L], ntok[L], flags[L];
    uint32_t pre_bytes[L], pre_tok[L];
    uint32_t tok_dst[L], tok_len[L], tok_dist[L], dep_lo[L], dep_hi[L];
};

template <uint32_t L>
struct Shared {
    uint32_t lit_lut[1u << kLitBits];
    uint16_t dist_lut[1u << kDistBits];                 // symbol << 4 | code length; 0: a longer code, or none
    uint16_t sorted[2][kLitSymbols];                    // symbols in code order: [0] literal/length, [1] distance (and the code-length code)
    uint32_t lim[2][kMaxBits + 1];                      // [l]: end of the codes of length <= l, written left-justified in 15 bits
    int32_t base[2][kMaxBits + 1];                      // [l]: place in sorted[] of the first code of length l, minus that code
    uint32_t count[2][kMaxBits + 1];
    uint32_t offs[2][kMaxBits + 1];
    uint8_t lens[kLitSymbols + 32 + 4];
    uint8_t cl[20];
    union {
        PerLane<L> pl;                                  // what the lanes hand each other while a block's codes are decoded
        uint16_t single[1u << kLitBits];                // while the literal/length table is built: one symbol per entry
    };
    uint32_t same[10];                                  // what lane 0 found out for everybody
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

FQD_HD uint32_t popcount64(uint64_t v) { return uint32_t(__builtin_popcountll(v)); }
FQD_HD uint32_t lowest_bit64(uint64_t v) { return uint32_t(__builtin_ctzll(v)); }

// Bits of the member, least significant first, read as aligned 32-bit words from any bit position on.  The words
// come four at a time and AHEAD of their use: a memory instruction costs the CU's one address unit some 25 to 60
// clocks however few lanes ask (tools/vmem_probe.hip), and with 64 lanes each wanting a word every few codes that is
// one instruction per turn of the decode loop unless a lane asks seldom; and the load is in flight while the bits
// before it are decoded.  Words beyond the member read as zero: a lane that runs off the end decodes zeros until
// its position says so.
struct Bits {
    const uint32_t* words = nullptr;
    uint32_t end_word = 0, lead = 0;      // words that hold the member; bits of words[0] before its first byte
    uint64_t buf = 0;
    uint32_t cnt = 0, next = 0;           // valid bits of buf; index of the first word not yet in buf
    uint64_t q_lo = 0, q_hi = 0;          // words[next ...], `have` of them
    uint32_t have = 0;
    uint32_t at = 0;                      // bit position of the head of buf, from the member's first bit

    FQD_HD void open(const uint8_t* p, uint32_t nbytes)
    {
        const uint32_t skip = uint32_t(reinterpret_cast<uintptr_t>(p) & 3u);
        words = reinterpret_cast<const uint32_t*>(p - skip);
        end_word = (skip + nbytes + 3u) / 4u;
        lead = 8u * skip;
    }
    FQD_HD uint32_t word(uint32_t i) const { return i < end_word ? words[i] : 0u; }
    FQD_HD void fetch(uint32_t i)                         // q = words[i .. i + 4)
    {
        if (i + 4u <= end_word) {
            uint64_t v[2];
            __builtin_memcpy(v, words + i, 16);
            q_lo = v[0]; q_hi = v[1];
        } else {
            q_lo = uint64_t(word(i)) | (uint64_t(word(i + 1u)) << 32);
            q_hi = uint64_t(word(i + 2u)) | (uint64_t(word(i + 3u)) << 32);
        }
        have = 4u;
    }
    FQD_HD void seek(uint32_t bit)
    {
        const uint32_t a = bit + lead;
        next = a >> 5;
        const uint64_t lo = word(next), hi = word(next + 1u);
        next += 2u;
        fetch(next);
        buf = (lo | (hi << 32)) >> (a & 31u);
        cnt = 64u - (a & 31u);
        at = bit;
    }
    FQD_HD uint32_t pos() const { return at; }
    FQD_HD void ensure()                                  // at least 33 bits afterwards
    {
        if (cnt <= 32u) {
            buf |= (q_lo & 0xFFFFFFFFull) << cnt;
            cnt += 32u; ++next;
            q_lo = (q_lo >> 32) | (q_hi << 32); q_hi >>= 32;
            if (--have == 0u) fetch(next);
        }
    }
    // Called by all lanes of a wave at the same turns: a lane that is running low fetches NOW, with the others, instead
    // of alone a few turns later — one memory instruction for the wave every few turns instead of one per turn (some
    // lane of 64 always is about to run dry).  The words still queued are fetched again with the rest.
    FQD_HD void top_up() { if (have <= 2u) fetch(next); }
    FQD_HD uint32_t peek(uint32_t n) const { return uint32_t(buf) & ((1u << n) - 1u); }     // n <= 31
    FQD_HD void skip(uint32_t n) { buf >>= n; cnt -= n; at += n; }
    FQD_HD uint32_t take(uint32_t n) { const uint32_t v = peek(n); skip(n); return v; }
};

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

// The code at the head of w (15 bits of the stream, its first bit on top): a canonical code written left-justified
// is smaller than every longer one, so counting the limits w has passed gives its length.  0xFFFF: no code starts so.
template <class S>
FQD_HD uint32_t canonical(const S& sh, uint32_t which, uint32_t w, uint32_t& len)
{
    uint32_t l = 1;
#pragma unroll
    for (uint32_t k = 1; k < kMaxBits; ++k) l += w >= sh.lim[which][k] ? 1u : 0u;
    if (w >= sh.lim[which][kMaxBits]) return 0xFFFFu;
    len = l;
    return sh.sorted[which][uint32_t(sh.base[which][l] + int32_t(w >> (kMaxBits - l)))];
}

// Tables of one code from the lengths lens[0..n) (0 = unused symbol): sorted[], lim[], base[] and the first-level
// table `lut` of 2^bits entries.  false for an over-subscribed set; an incomplete one passes only where zlib lets
// it pass here: a distance code with no code at all or a single one-bit code.
template <class Ctx, class S>
FQD_HD bool build_code(Ctx& ctx, S& sh, uint32_t which, uint32_t n, const uint8_t* lens, uint16_t* lut, uint32_t bits, bool may_be_single)
{
    constexpr uint32_t L = Ctx::kLanes;
    ctx.lanes([&](uint32_t lane) { for (uint32_t l = lane; l <= kMaxBits; l += L) sh.count[which][l] = 0; });
    ctx.lanes([&](uint32_t lane) { for (uint32_t s = lane; s < n; s += L) ctx.add(&sh.count[which][lens[s]], 1u); });
    ctx.lanes([&](uint32_t lane) {
        if (lane != 0) return;
        int32_t left = 1;
        uint32_t codes = 0, code = 0, offset = 0, ok = 1;
        for (uint32_t l = 1; l <= kMaxBits; ++l) {
            const uint32_t c = sh.count[which][l];
            left = (left << 1) - int32_t(c);
            if (left < 0) { ok = 0; break; }
            codes += c;
            sh.base[which][l] = int32_t(offset) - int32_t(code);
            sh.lim[which][l] = (code + c) << (kMaxBits - l);
            sh.offs[which][l] = offset;
            offset += c;
            code = (code + c) << 1;
        }
        if (ok && left > 0 && !(may_be_single && (codes == 0u || (codes == 1u && sh.count[which][1] == 1u)))) ok = 0;
        sh.same[0] = ok;
    });
    if (!ctx.same(sh.same[0])) return false;
    // symbols in code order: the lengths share the lanes, each walks the symbols for its own
    ctx.lanes([&](uint32_t lane) {
        for (uint32_t l = 1u + lane; l <= kMaxBits; l += L) {
            uint32_t at = sh.offs[which][l];
            if (sh.count[which][l] == 0u) continue;
            for (uint32_t s = 0; s < n; ++s) if (lens[s] == l) sh.sorted[which][at++] = uint16_t(s);
        }
    });
    ctx.lanes([&](uint32_t lane) {
        for (uint32_t idx = lane; idx < (1u << bits); idx += L) {
            uint32_t len = 0;
            const uint32_t sym = canonical(sh, which, reverse_bits32(idx) >> 17, len);   // the bits beyond `bits` read as zero:
            lut[idx] = sym != 0xFFFFu && len <= bits ? uint16_t((sym << 4) | len) : uint16_t(0);   // a code that short is decided by its own bits
        }
    });
    return true;
}

// The literal/length table proper, from the one-symbol-per-entry table in sh.single: an entry whose first code is a
// literal takes in the literals that follow while their codes still end inside the index (a code that ends there is
// decided by the bits of the index alone).
template <class Ctx, class S>
FQD_HD void pack_literals(Ctx& ctx, S& sh)
{
    constexpr uint32_t L = Ctx::kLanes;
    ctx.lanes([&](uint32_t lane) {
        for (uint32_t idx = lane; idx < (1u << kLitBits); idx += L) {
            const uint32_t e = sh.single[idx], len = e & 15u, sym = e >> 4;
            uint32_t packed = 0;
            if (len != 0u) {
                packed = pack_symbol(sym, len);
                if (sym < 256u) {
                    uint32_t used = len, n = 1;
                    while (n < 3u) {
                        const uint32_t e2 = sh.single[idx >> used], l2 = e2 & 15u, s2 = e2 >> 4;
                        if (l2 == 0u || s2 >= 256u || used + l2 > kLitBits) break;
                        packed |= s2 << (8u + 8u * n);
                        used += l2; ++n;
                    }
                    packed = (packed & ~0x3Fu) | used | (n << 4);
                }
            }
            sh.lit_lut[idx] = packed;
        }
    });
}

// One lane over its subsequence: from the bit `from` (a code boundary, or a guess at one) to the first code boundary
// at or beyond `stop`, or to the end-of-block code, or to something that is no code.  kWrite: the literals go to
// out[] from `pos` on and the matches to tok[] from `tk` on (the counts of the run before said where).
FQD_HD uint64_t load8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }      // (any alignment)
FQD_HD void store8(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

FQD_HD void store_low(uint8_t* d, uint64_t v, uint32_t n)                       // the low n < 8 bytes of v
{
    if (n & 4u) { const uint32_t w = uint32_t(v); __builtin_memcpy(d, &w, 4); d += 4; v >>= 32; }
    if (n & 2u) { const uint16_t w = uint16_t(v); __builtin_memcpy(d, &w, 2); d += 2; v >>= 16; }
    if (n & 1u) *d = uint8_t(v);
}

// Do most of the lanes that are still decoding say yes?  (On the CPU a lane is asked alone.)  The answer only decides
// in which order the lanes of a wave get their turns, never what they decode.
FQD_HD bool most_lanes(bool mine)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 2 * __popcll(__ballot(mine)) >= __popcll(__ballot(true));
#else
    return mine;
#endif
}

constexpr uint32_t kBurst = 32;                        // literal turns in a row before a lane that waits with a match gets its own

template <bool kWrite, class S>
FQD_HD void decode_range(const S& sh, Bits& in, uint32_t from, uint32_t stop, uint32_t total_bits,
                         uint32_t& end, uint32_t& nbytes, uint32_t& ntok, uint32_t& flags,
                         uint8_t* out, uint32_t out_len, uint32_t pos, Token* tok, uint32_t tk, uint8_t* out2 = nullptr)
{
    in.seek(from);
    uint32_t bytes = 0, matches = 0, fl = 0, burst = 0;
    const uint32_t limit = stop < total_bits ? stop : total_bits;
    // kWrite: literals wait in a register until there are eight of them — one store instead of eight
    uint64_t acc = 0;
    uint32_t waiting = 0;                                             // bytes in acc: out[pos - waiting, pos)
    auto put = [&](uint32_t lits, uint32_t n) {                       // n <= 3 literals, the first one lowest
        acc |= uint64_t(lits) << (8u * waiting);
        waiting += n; pos += n;
        if (waiting >= 8u) {
            if (pos - waiting + 8u <= out_len) { store8(out + (pos - waiting), acc); if (out2) store8(out2 + (pos - waiting), acc); }
            waiting -= 8u;
            acc = waiting ? uint64_t(lits) >> (8u * (n - waiting)) : 0ull;
        }
    };
    auto flush = [&]() {
        if (waiting && pos <= out_len) { store_low(out + (pos - waiting), acc, waiting); if (out2) store_low(out2 + (pos - waiting), acc, waiting); }
        waiting = 0; acc = 0;
    };
    // A turn of this loop costs every lane of the wave every branch some lane takes, and literals are most of what
    // a FASTQ stream holds: while most lanes have a literal next, a turn is the literal's few instructions only and
    // the lanes that have a length code next wait (kBurst turns at most).
    uint32_t turn = 0;
    while (in.at < limit) {
        if ((turn++ & 3u) == 0u) in.top_up();
        in.ensure();
        uint32_t e = sh.lit_lut[in.peek(kLitBits)], len = e & 15u, sym = e >> 8;
        uint32_t n = (e >> 4) & 3u;
        if (in.at + kLitBits > limit) {                               // the last codes before the limit go one by one: where a
            len = 0; n = 0;                                           // lane stops must not depend on how literals were grouped
        }
        const bool literal = n != 0u;
        if (literal) {
            in.skip(len);
            if (kWrite) put(sym, n);
            bytes += n;
        }
        if (most_lanes(literal) && ++burst < kBurst) continue;
        burst = 0;
        if (literal) continue;
        if (len == 0u) {
            sym = canonical(sh, 0, reverse_bits32(uint32_t(in.buf)) >> 17, len);
            if (sym == 0xFFFFu) { fl |= kBroken; in.skip(1); continue; }
        }
        in.skip(len);
        if (sym < 256u) {                                             // (a literal with a code longer than the table's index)
            if (kWrite) put(sym, 1u);
            ++bytes;
            continue;
        }
        if (sym == 256u) { fl |= kEndOfBlock; break; }
        if (sym > 285u) { fl |= kBroken; continue; }
        uint32_t eb;
        uint32_t length = length_base(sym - 257u, eb);
        length += in.take(eb);
        in.ensure();
        e = sh.dist_lut[in.peek(kDistBits)]; len = e & 15u;
        uint32_t dsym = e >> 4;
        if (len == 0u) {
            dsym = canonical(sh, 1, reverse_bits32(uint32_t(in.buf)) >> 17, len);
            if (dsym == 0xFFFFu) { fl |= kBroken; in.skip(1); continue; }
        }
        in.skip(len);
        if (dsym >= kDistSymbols) { fl |= kBroken; continue; }
        uint32_t dist = dist_base(dsym, eb);
        dist += in.take(eb);
        if (kWrite) {
            flush();
            if (tk < kTokenRoom) { tok[tk].dst = pos; tok[tk].len_dist = (length << 16) | dist; }
            ++tk;
            pos += length;
        }
        bytes += length; ++matches;
    }
    if (kWrite) flush();
    if (!(fl & kEndOfBlock) && in.at < stop) fl |= kBroken | kOffTheEnd;       // the member ends inside a block
    end = in.at; nbytes = bytes; ntok = matches; flags = fl;
}

// The first `dist` < 8 bytes at s, repeated to fill a register.
FQD_HD uint64_t period_of(const uint8_t* s, uint32_t dist)
{
    uint64_t pat = 0;
    for (uint32_t i = 0; i < dist; ++i) pat |= uint64_t(s[i]) << (8u * i);
    for (uint32_t sh = 8u * dist; sh < 64u; sh *= 2u) pat |= pat << sh;
    return pat;
}

// What a copy costs is its memory instructions — every one of them is 64 separate requests to the CU's one address
// unit, whatever the lanes ask for — and the trips to memory that must wait for each other.  So a SHORT match (fewer
// than kLongMatch bytes) is one lane's work and reads all it may before it writes: three loads and three stores of
// eight bytes at most.  `per` = 0: a plain copy of len bytes from d - dist (dist >= len).  Otherwise the match
// repeats its first `per` bytes, which lie at d - dist (right before d as the stream had it: dist == per; further
// back when resolve_matches found where THOSE bytes were copied from).
constexpr uint32_t kLongMatch = 24;
FQD_HD void copy_plain_short(uint8_t* d, const uint8_t* s, uint32_t len)       // len < kLongMatch, nothing read is written here
{
    if (len < 8u) {
        uint64_t v = 0;
        if (s + 8 <= d) v = load8(s);                                           // (eight bytes from s on are the member's: they end before d)
        else for (uint32_t i = 0; i < len; ++i) v |= uint64_t(s[i]) << (8u * i);
        store_low(d, v, len);
        return;
    }
    const uint32_t o2 = len - 8u < 8u ? len - 8u : 8u, o3 = len - 8u;
    const uint64_t a = load8(s), b = load8(s + o2);
    uint64_t c = 0;
    if (len > 16u) c = load8(s + o3);
    store8(d, a);
    if (len > 8u) store8(d + o2, b);
    if (len > 16u) store8(d + o3, c);
}
FQD_HD void copy_short(uint8_t* d, uint32_t len, uint32_t dist, uint32_t per)
{
    const uint8_t* s = d - dist;
    if (per == 0u) { copy_plain_short(d, s, len); return; }
    if (per >= 8u) {                                                            // its first period from where that lies, the rest from itself
        copy_plain_short(d, s, per);
        uint32_t k = per;
        for (; k + 8u <= len; k += 8u) store8(d + k, load8(d + k - per));
        if (k < len) { uint64_t v = 0; for (uint32_t i = 0; k + i < len; ++i) v |= uint64_t(d[k + i - per]) << (8u * i); store_low(d + k, v, len - k); }
        return;
    }
    const uint64_t pat = period_of(s, per);
    const uint32_t step = (8u / per) * per;                                     // a store advances by whole periods
    uint32_t k = 0;
    for (; k + 8u <= len; k += step) store8(d + k, pat);
    if (k < len) store_low(d + k, pat, len - k);
}

// A LONG match is the work of many lanes, sixteen bytes each, side by side (chunk i = bytes [16 i, 16 i + 16), the
// last one moved back to end with the match): one trip to memory for the whole match, and requests that fall into
// the same few lines.
FQD_HD void copy_chunk(uint8_t* d, uint32_t len, uint32_t dist, uint32_t per, uint32_t i)
{
    uint32_t o = 16u * i;
    if (o >= len) return;
    if (o + 16u > len) o = len - 16u;
    const uint8_t* s = d - dist;
    uint64_t a, b;
    if (per == 0u || o + 16u <= per) { a = load8(s + o); b = load8(s + o + 8u); }      // bytes that were there before the match
    else if (per < 8u) {                                                        // a short period: from a register
        const uint64_t pat = period_of(s, per);
        uint32_t at = o % per;
        a = b = 0;
        for (uint32_t k = 0; k < 8u; ++k) { a |= ((pat >> (8u * at)) & 0xFFull) << (8u * k); at = at + 1u == per ? 0u : at + 1u; }
        for (uint32_t k = 0; k < 8u; ++k) { b |= ((pat >> (8u * at)) & 0xFFull) << (8u * k); at = at + 1u == per ? 0u : at + 1u; }
    } else {                                                                    // the match runs into itself: byte by byte from its first period
        uint32_t at = o % per;
        a = b = 0;
        for (uint32_t k = 0; k < 8u; ++k) { a |= uint64_t(s[at]) << (8u * k); at = at + 1u == per ? 0u : at + 1u; }
        for (uint32_t k = 0; k < 8u; ++k) { b |= uint64_t(s[at]) << (8u * k); at = at + 1u == per ? 0u : at + 1u; }
    }
    store8(d + o, a); store8(d + o + 8u, b);
}

// Matches of one window, in stream order, 64 at a time.  A match may be copied once no unresolved match of its group
// writes into the bytes it reads (everything before the group is final: literals are, and earlier groups are done).
template <class Ctx, class S>
FQD_HD bool resolve_matches(Ctx& ctx, S& sh, uint8_t* out, const Token* tok, uint32_t ntok, uint8_t* out2 = nullptr)
{   // out2: a SECOND text the same tokens are applied to (csrc/fqd_gunzip.hip: the two planes of a unit, which differ only in the
    // 32 KiB before the stretch): every copy is made in both, in the same round — the trips to memory that wait for each other are shared
    constexpr uint32_t L = Ctx::kLanes;
    bool fine = true;
    for (uint32_t g0 = 0; g0 < ntok; g0 += L) {
        const uint32_t n = ntok - g0 < L ? ntok - g0 : L;
        ctx.lanes([&](uint32_t j) {
            if (j >= n) return;
            const Token t = tok[g0 + j];
            const uint32_t len = t.len_dist >> 16, dist = t.len_dist & 0xFFFFu;
            sh.pl.tok_dst[j] = t.dst; sh.pl.tok_len[j] = len; sh.pl.tok_dist[j] = dist;
            sh.pl.pre_bytes[j] = dist < len ? dist : 0u;                // its period, if it repeats itself (pre_bytes is free by now)
        });
        const uint64_t broken = ctx.ballot([&](uint32_t j) { return j < n && sh.pl.tok_dist[j] > sh.pl.tok_dst[j]; });
        if (broken) { fine = false; break; }                      // a distance that reaches before the member's first byte
        // Which earlier matches of the group a match must wait for — and, first, whether it must wait at all: a match
        // that reads nothing but what ONE earlier match of the group writes, and that match a plain copy from
        // further back, may as well read where that one reads (a line of qualities copied from the record before,
        // which was copied from the record before it, ...: every one of them ends up reading the first; a run of
        // one quality that starts with the last byte of such a copy takes that byte from where the copy took it).
        // The chains halve with every turn.
        for (uint32_t turn = 0;; ++turn) {
            ctx.lanes([&](uint32_t j) {
                if (j >= n) return;
                const uint32_t dst = sh.pl.tok_dst[j], dist = sh.pl.tok_dist[j], src = dst - dist, len = sh.pl.tok_len[j], per = sh.pl.pre_bytes[j];
                const uint32_t rend = src + (per ? per : len);                    // the bytes read that are not its own: [src, rend)
                uint64_t dep = 0;
                uint32_t further = dist;
                if (j != 0u && rend > sh.pl.tok_dst[0]) {
                    // earlier matches of the group lie in stream order: [a, b] are those that write into [src, rend)
                    uint32_t lo = 0, hi = j;                                       // a = first i with dst_i + len_i > src
                    while (lo < hi) { const uint32_t mid = (lo + hi) / 2u; if (sh.pl.tok_dst[mid] + sh.pl.tok_len[mid] > src) hi = mid; else lo = mid + 1u; }
                    const uint32_t a = lo;
                    lo = a; hi = j;                                                // b + 1 = first i >= a with dst_i >= rend
                    while (lo < hi) { const uint32_t mid = (lo + hi) / 2u; if (sh.pl.tok_dst[mid] >= rend) hi = mid; else lo = mid + 1u; }
                    const uint32_t b1 = lo;
                    if (b1 > a) dep = (b1 - a >= 64u ? ~0ull : ((1ull << (b1 - a)) - 1ull)) << a;
                    if (b1 == a + 1u && turn < 7u) {
                        const uint32_t adst = sh.pl.tok_dst[a], alen = sh.pl.tok_len[a], adist = sh.pl.tok_dist[a];
                        if (sh.pl.pre_bytes[a] == 0u && adst <= src && rend <= adst + alen) further = dist + adist;
                    }
#if defined(FQD_WINF_STATS)                                                     // what a match still waits for, per turn
                    if (b1 == a + 1u && further == dist) {
                        const bool inside = sh.pl.tok_dst[a] <= src && rend <= sh.pl.tok_dst[a] + sh.pl.tok_len[a];
                        if (inside) FQD_WINF_COUNT(dep_one_periodic, 1); else FQD_WINF_COUNT(dep_one_partial, 1);
                    } else if (b1 > a + 1u) FQD_WINF_COUNT(dep_many, 1);
#endif
                }
                sh.pl.dep_lo[j] = uint32_t(dep); sh.pl.dep_hi[j] = uint32_t(dep >> 32);
                sh.pl.pre_tok[j] = further;
            });
            const uint64_t moved = ctx.ballot([&](uint32_t j) { return j < n && sh.pl.pre_tok[j] != sh.pl.tok_dist[j]; });
            if (!moved) break;
            ctx.lanes([&](uint32_t j) { if (j < n) sh.pl.tok_dist[j] = sh.pl.pre_tok[j]; });
        }
        const uint64_t all = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
        uint64_t done = 0;
        FQD_WINF_COUNT(groups, 1); FQD_WINF_COUNT(tokens, n);
        while (done != all) {
            FQD_WINF_COUNT(group_rounds, 1);
            const uint64_t go = ctx.ballot([&](uint32_t j) {
                if (j >= n || ((done >> j) & 1ull)) return false;
                const uint64_t dep = uint64_t(sh.pl.dep_lo[j]) | (uint64_t(sh.pl.dep_hi[j]) << 32);
                return (dep & ~done) == 0ull;
            });
            ctx.mark(5);
            const uint64_t big = go & ctx.ballot([&](uint32_t j) { return j < n && sh.pl.tok_len[j] >= kLongMatch; });
            ctx.lanes_open([&](uint32_t j) {
                if (!(((go & ~big) >> j) & 1ull)) return;
                copy_short(out + sh.pl.tok_dst[j], sh.pl.tok_len[j], sh.pl.tok_dist[j], sh.pl.pre_bytes[j]);
                if (out2) copy_short(out2 + sh.pl.tok_dst[j], sh.pl.tok_len[j], sh.pl.tok_dist[j], sh.pl.pre_bytes[j]);
            });
            // (matches that go in the same round neither read nor write each other's bytes: no need to wait in between)
            constexpr uint32_t kAtOnce = L >= 63u ? 3u : 1u, kPerMatch = L / kAtOnce;     // 17 chunks hold 258 bytes
            for (uint64_t rest = big; rest;) {
                uint32_t t0 = 64u, t1 = 64u, t2 = 64u;
                t0 = lowest_bit64(rest); rest &= rest - 1ull;
                if (kAtOnce > 1u && rest) { t1 = lowest_bit64(rest); rest &= rest - 1ull; }
                if (kAtOnce > 2u && rest) { t2 = lowest_bit64(rest); rest &= rest - 1ull; }
                ctx.lanes_open([&](uint32_t k) {
                    const uint32_t q = k / kPerMatch;
                    if (q >= kAtOnce) return;
                    const uint32_t j = q == 0u ? t0 : q == 1u ? t1 : t2;
                    if (j >= 64u) return;
                    for (uint32_t i = k % kPerMatch; i < 17u; i += kPerMatch) {
                        copy_chunk(out + sh.pl.tok_dst[j], sh.pl.tok_len[j], sh.pl.tok_dist[j], sh.pl.pre_bytes[j], i);
                        if (out2) copy_chunk(out2 + sh.pl.tok_dst[j], sh.pl.tok_len[j], sh.pl.tok_dist[j], sh.pl.pre_bytes[j], i);
                    }
                });
            }
            ctx.sync();
            ctx.mark(7);
            done |= go;
        }
    }
    return fine;
}

// The whole member (kUnit = false): `tok` is room for kTokenRoom tokens (any memory this wave has to itself); kOk iff
// exactly out_len bytes came out.
// A STRETCH of a longer deflate stream (kUnit = true; csrc/fqd_gunzip.hip: a unit of an ordinary gzip file): decoding starts
// at bit `first_bit` of comp — a block boundary — and ends at the first boundary at or after `stop_bit` where a dynamic,
// non-final block begins (what a unit's start is guessed by), or with the end of the final block; the text goes to
// out[out_start ...) (out_len: all the room there is, out_start included), and matches may reach back into out[0, out_start):
// the 32 KiB before the stretch, whatever the caller put there.  kOk: info[0] = the bit the stretch ended at, info[1] = where
// its text ended in out, info[2] = 1 (a boundary) or 2 (the final block's end).
constexpr uint32_t kStopHere = 100;
template <bool kUnit, class Ctx, class S>
FQD_HD uint32_t inflate_impl(Ctx& ctx, S& sh, const uint8_t* comp, uint32_t comp_len, uint32_t first_bit, uint32_t stop_bit,
                             uint8_t* out, uint32_t out_start, uint32_t out_len, Token* tok, uint32_t* info, uint8_t* out2 = nullptr)
{
    constexpr uint32_t L = Ctx::kLanes;
    const uint32_t total_bits = comp_len * 8u;
    uint32_t bitpos = first_bit, outpos = out_start;
    for (;;) {
        // ---- block header: lane 0 reads it, everybody learns what it said ------------------------------------------
        ctx.lanes([&](uint32_t lane) {
            if (lane != 0) return;
            Bits in; in.open(comp, comp_len); in.seek(bitpos); in.ensure();
            uint32_t status = kOk;
            const uint32_t last = in.take(1), type = in.take(2);
            uint32_t a = 0, b = 0, c = 0;
            if (kUnit && bitpos != first_bit && !last && type == 2u && bitpos >= stop_bit) status = kStopHere;
            else if (type == 0u) {
                in.seek((in.pos() + 7u) & ~7u); in.ensure();
                a = in.take(16); in.ensure();
                const uint32_t nn = in.take(16);
                if ((a ^ nn) != 0xFFFFu) status = kBadStored;
            } else if (type == 2u) {
                a = in.take(5) + 257u; b = in.take(5) + 1u; c = in.take(4) + 4u;
                if (a > 286u || b > kDistSymbols) status = kBadLengths;
                constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (uint32_t i = 0; i < 19u; ++i) sh.cl[i] = 0;
                for (uint32_t i = 0; i < c; ++i) { in.ensure(); sh.cl[order[i]] = uint8_t(in.take(3)); }
            } else if (type == 3u) status = kBadBlockType;
            sh.same[1] = status; sh.same[2] = last; sh.same[3] = type; sh.same[4] = a; sh.same[5] = b; sh.same[6] = in.pos();
        });
        ctx.mark(0);
        uint32_t status = ctx.same(sh.same[1]);
        if (kUnit && status == kStopHere) { info[0] = bitpos; info[1] = outpos; info[2] = 1u; return kOk; }
        if (status != kOk) return status;
        const uint32_t last = ctx.same(sh.same[2]), type = ctx.same(sh.same[3]);
        bitpos = ctx.same(sh.same[6]);
        if (bitpos > total_bits) return kInputOverrun;
        if (type == 0u) {
            const uint32_t n = ctx.same(sh.same[4]), first = bitpos / 8u;
            if (outpos + n > out_len) return kOutputOverrun;
            if (first + n > comp_len) return kInputOverrun;
            ctx.lanes([&](uint32_t lane) { for (uint32_t k = lane; k < n; k += L) { out[outpos + k] = comp[first + k]; if (out2) out2[outpos + k] = comp[first + k]; } });
            outpos += n; bitpos += 8u * n;
            ctx.mark(6);
            if (last) break;
            continue;
        }
        if (type == 1u) {
            ctx.lanes([&](uint32_t lane) {
                for (uint32_t s = lane; s < kLitSymbols; s += L) sh.lens[s] = uint8_t(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u);
                for (uint32_t s = lane; s < 32u; s += L) sh.lens[kLitSymbols + s] = 5;    // 32 five-bit codes; 30 and 31 never occur in valid data
            });
            build_code(ctx, sh, 0, kLitSymbols, sh.lens, sh.single, kLitBits, false);
            pack_literals(ctx, sh);
            build_code(ctx, sh, 1, 32u, sh.lens + kLitSymbols, sh.dist_lut, kDistBits, false);
        } else {
            const uint32_t nlen = ctx.same(sh.same[4]), ndist = ctx.same(sh.same[5]);
            // the code-length code borrows the distance tables until the real ones are built
            if (!build_code(ctx, sh, 1, 19u, sh.cl, sh.dist_lut, kDistBits, false)) return kBadLengths;
            ctx.mark(1);
            ctx.lanes([&](uint32_t lane) {
                if (lane != 0) return;
                Bits in; in.open(comp, comp_len); in.seek(bitpos);
                uint32_t i = 0, bad = 0;
                while (i < nlen + ndist) {
                    in.ensure();
                    const uint32_t e = sh.dist_lut[in.peek(kDistBits)];
                    if ((e & 15u) == 0u) { bad = 1; break; }              // (these codes have 7 bits at most: all of them are in the table)
                    in.skip(e & 15u);
                    const uint32_t sym = e >> 4;
                    if (sym < 16u) { sh.lens[i++] = uint8_t(sym); continue; }
                    uint32_t prev = 0, rep;
                    if (sym == 16u) { if (i == 0) { bad = 1; break; } prev = sh.lens[i - 1u]; rep = 3u + in.take(2); }
                    else if (sym == 17u) rep = 3u + in.take(3);
                    else if (sym == 18u) rep = 11u + in.take(7);
                    else { bad = 1; break; }
                    if (i + rep > nlen + ndist) { bad = 1; break; }
                    while (rep--) sh.lens[i++] = uint8_t(prev);
                }
                if (!bad && sh.lens[256] == 0) bad = 1;
                if (!bad && in.pos() > total_bits) bad = 2;
                sh.same[1] = bad; sh.same[6] = in.pos();
            });
            ctx.mark(0);
            const uint32_t bad = ctx.same(sh.same[1]);
            if (bad) return bad == 2u ? kInputOverrun : kBadLengths;
            bitpos = ctx.same(sh.same[6]);
            // (the distance lengths move out of the way first: building the literal code reads lens[0..nlen) only, but
            //  the distance code is built into tables the code-length code no longer needs)
            if (!build_code(ctx, sh, 0, nlen, sh.lens, sh.single, kLitBits, false)) return kBadLengths;
            pack_literals(ctx, sh);
            if (!build_code(ctx, sh, 1, ndist, sh.lens + nlen, sh.dist_lut, kDistBits, true)) return kBadLengths;
        }
        ctx.mark(1);
        FQD_WINF_COUNT(blocks, 1);
        // ---- the block's codes, a window of L subsequences at a time ----------------------------------------------
        for (;;) {
            if (bitpos >= total_bits) return kInputOverrun;               // no end-of-block code before the member's end
            const uint32_t remaining = total_bits - bitpos;
            uint32_t sub = ((remaining + L - 1u) / L + 31u) & ~31u;
            sub = sub < kMinSub ? kMinSub : sub > kMaxSub ? kMaxSub : sub;
            const uint32_t active = (remaining + sub - 1u) / sub < L ? (remaining + sub - 1u) / sub : L;
            const uint64_t active_mask = active >= 64u ? ~0ull : ((1ull << active) - 1ull);
            ctx.lanes([&](uint32_t lane) { sh.pl.start[lane] = bitpos + lane * sub; sh.pl.flags[lane] = 0; sh.pl.nbytes[lane] = 0; sh.pl.ntok[lane] = 0; sh.pl.end[lane] = 0; });
            uint64_t moved = active_mask;
            FQD_WINF_COUNT(windows, 1);
            while (moved) {
                FQD_WINF_COUNT(rounds, 1); FQD_WINF_COUNT(lane_decodes, popcount64(moved));
                ctx.lanes([&](uint32_t lane) {
                    if (!((moved >> lane) & 1ull)) return;
                    Bits in; in.open(comp, comp_len);
                    uint32_t end, nb, nt, fl;
                    decode_range<false>(sh, in, sh.pl.start[lane], bitpos + (lane + 1u) * sub, total_bits, end, nb, nt, fl, nullptr, 0, 0, nullptr, 0);
                    sh.pl.end[lane] = end; sh.pl.nbytes[lane] = nb; sh.pl.ntok[lane] = nt; sh.pl.flags[lane] = fl;
                });
                moved = ctx.ballot([&](uint32_t lane) {
                    return lane != 0u && lane < active && !(sh.pl.flags[lane - 1u] & (kEndOfBlock | kOffTheEnd)) && sh.pl.end[lane - 1u] != sh.pl.start[lane];
                });
                ctx.lanes([&](uint32_t lane) { if ((moved >> lane) & 1ull) sh.pl.start[lane] = sh.pl.end[lane - 1u]; });
            }
            ctx.mark(2);
            // the chain is exact up to the first lane that stopped early; what lies beyond it belongs to no block yet
            const uint64_t stopped = ctx.ballot([&](uint32_t lane) { return lane < active && sh.pl.flags[lane] != 0u; });
            uint32_t valid = stopped ? lowest_bit64(stopped) + 1u : active;
            // where every lane's bytes and tokens go; a window with more matches than the token room holds ends earlier
            ctx.lanes([&](uint32_t lane) {
                if (lane != 0) return;
                uint32_t bytes = 0, toks = 0, fit = 0;
                for (uint32_t i = 0; i < valid; ++i) {
                    if (toks + sh.pl.ntok[i] > kTokenRoom) break;
                    sh.pl.pre_bytes[i] = bytes; sh.pl.pre_tok[i] = toks;
                    bytes += sh.pl.nbytes[i]; toks += sh.pl.ntok[i];
                    fit = i + 1u;
                }
                sh.same[7] = bytes; sh.same[8] = toks; sh.same[9] = fit;
            });
            const uint32_t fit = ctx.same(sh.same[9]);
            if (fit == 0u) return kBadCode;                               // (a lane's subsequence cannot hold that many matches)
            const bool cut = fit < valid;
            FQD_WINF_COUNT(cut, cut ? 1 : 0);
            valid = fit;
            const uint32_t bytes = ctx.same(sh.same[7]), toks = ctx.same(sh.same[8]);
            const uint32_t how = cut ? 0u : ctx.same(sh.pl.flags[valid - 1u]);
            if (how & kBroken) return how & kOffTheEnd ? kInputOverrun : kBadCode;
            if (outpos + bytes > out_len) return kOutputOverrun;
            ctx.mark(3);
            ctx.lanes([&](uint32_t lane) {
                if (lane >= valid) return;
                Bits in; in.open(comp, comp_len);
                uint32_t end, nb, nt, fl;
                decode_range<true>(sh, in, sh.pl.start[lane], bitpos + (lane + 1u) * sub, total_bits, end, nb, nt, fl,
                                   out, out_len, outpos + sh.pl.pre_bytes[lane], tok, sh.pl.pre_tok[lane], out2);
            });
            ctx.mark(4);
            if (!resolve_matches(ctx, sh, out, tok, toks, out2)) return kBadDistance;
            ctx.mark(5);
            outpos += bytes;
            bitpos = ctx.same(sh.pl.end[valid - 1u]);
            if (bitpos > total_bits) return kInputOverrun;                // the last code read bits the member does not have
            if (how & kEndOfBlock) break;
        }
        if (last) break;
    }
    if (kUnit) { info[0] = bitpos; info[1] = outpos; info[2] = 2u; return kOk; }
    return outpos == out_len ? uint32_t(kOk) : uint32_t(kShortOutput);
}
template <class Ctx, class S>
FQD_HD uint32_t inflate_member(Ctx& ctx, S& sh, const uint8_t* comp, uint32_t comp_len, uint8_t* out, uint32_t out_len, Token* tok)
{
    return inflate_impl<false>(ctx, sh, comp, comp_len, 0u, 0xFFFFFFFFu, out, 0u, out_len, tok, nullptr);
}
template <class Ctx, class S>
FQD_HD uint32_t inflate_stretch(Ctx& ctx, S& sh, const uint8_t* comp, uint32_t comp_len, uint32_t first_bit, uint32_t stop_bit,
                                uint8_t* out, uint32_t out_start, uint32_t out_len, Token* tok, uint32_t* info, uint8_t* out2 = nullptr)
{   // out2: a second room of the same size whose text differs from out's only through what the caller put before out_start in each
    return inflate_impl<true>(ctx, sh, comp, comp_len, first_bit, stop_bit, out, out_start, out_len, tok, info, out2);
}

} // namespace winf
} // namespace fqd
