// fqd_bgzf_core.hpp — BGZF members (gzip, RFC 1952, with the 'BC' size field) made on the GPU:
// the pieces that do not care where they run.
//
// The reference writes `.gz` outputs through Boost's zlib filter (file_utils.hpp:71-82), one
// thread, level 6.  Here the surviving records are already in HBM when they are to be written
// (host/run_resident.cpp + survivor_writer.cpp, the resident runs), so the deflate happens there too and only
// the compressed bytes cross PCIe.  What a member holds is text of FASTQ/FASTA records, and the
// coder is built for that and nothing else:
//   * a member (65280 input bytes) belongs to one workgroup of 512 threads; thread t owns the 128
//     bytes [L-(512-t)*128, L-(511-t)*128) of it (right-aligned, the leftmost chunks of a short
//     member are empty) and parses them greedily into literals and matches that stay inside the
//     chunk — so every chunk is parsed, priced and emitted independently of its neighbours;
//   * two match candidates per position, both found without any hash table: the byte run ending
//     here (distance 1: quality strings) and the same column of the line `lines_per_record` lines
//     up (the previous record's ID line under this one: instrument, run, flowcell, lane, tile
//     prefixes); sequence lines are left to the Huffman code, as they are incompressible by LZ77;
//   * ONE pair of dynamic Huffman codes (RFC 1951 §3.2.7) per call, from the token histogram of
//     the whole buffer (kernel 1), built on the host in microseconds, written into the header of
//     every member's single block; a member the code would expand is stored instead (§3.2.4);
//   * CRC-32 of the member on the device: a table-driven register per chunk, combined pairwise
//     with precomputed "advance by 128 * 2^k zero bytes" matrices (the register update is affine).
//
// Everything here is `FQD_HD`: the kernels in fqd_bgzf.hip call these functions between
// barriers, and tests/native/bgzf_core_check.cpp runs the very same functions thread by thread on
// the CPU (no GPU needed) and inflates the result with zlib.  The product path is the HIP one only.
#pragma once

#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define FQD_HD __host__ __device__ __forceinline__
#else
#define FQD_HD inline
#endif

namespace fqd {
namespace bgzf {

constexpr uint32_t kMember = 65280;                    // input bytes per member (as bgzip)
constexpr uint32_t kSlot = 65536;                      // bytes reserved per member before compaction
constexpr uint32_t kThreads = 512;
constexpr uint32_t kChunk = 128;                       // kThreads * kChunk >= kMember
constexpr uint32_t kMaxLines = 3800;                   // line starts kept per member; beyond: no column matches
constexpr uint32_t kMinMatch = 4;
constexpr uint32_t kMaxMatch = 258;
constexpr uint32_t kLitLen = 286, kDist = 30;
constexpr uint32_t kHeadBytes = 18, kTailBytes = 8;    // BGZF header, CRC-32 + ISIZE
constexpr uint32_t kLevels = 9;                        // log2(kThreads)

static_assert(kThreads * kChunk >= kMember, "chunks must cover a member");

// What the host hands to the emit kernel (device memory; small enough to be read through the
// scalar cache).  Codes are stored bit-reversed: deflate packs Huffman codes starting from their
// most significant bit into a stream that is otherwise filled from bit 0 upwards.
struct Codes {
    uint32_t lit[kLitLen];                             // reversed code | length << 16
    uint32_t dist[kDist];
    uint32_t header_bits;                              // BFINAL, BTYPE = 10, HLIT, HDIST, HCLEN, the three tables
    uint32_t header[48];                               // ... as a bit string, bit 0 of word 0 first
    uint32_t crc_table[256];
    uint32_t crc_shift[kLevels][32];                   // [k][b] = register after 128 * 2^k zero bytes, started from 1 << b
};

// -------------------------------------------------------------------------------------------
// RFC 1951 §3.2.5: symbol, number of extra bits and their value for a match length / distance.
struct Sym { uint32_t sym, ebits, eval; };

FQD_HD Sym length_symbol(uint32_t len)                 // 3..258
{
    const uint32_t y = len - 3u;
    if (y < 8u) return {257u + y, 0u, 0u};
    if (len == 258u) return {285u, 0u, 0u};
    const uint32_t hb = 31u - uint32_t(__builtin_clz(y)), eb = hb - 2u;
    return {257u + 4u * (hb - 1u) + ((y >> eb) & 3u), eb, y & ((1u << eb) - 1u)};
}

FQD_HD Sym dist_symbol(uint32_t dist)                  // 1..32768
{
    const uint32_t x = dist - 1u;
    if (x < 4u) return {x, 0u, 0u};
    const uint32_t hb = 31u - uint32_t(__builtin_clz(x)), eb = hb - 1u;
    return {2u * hb + ((x >> eb) & 1u), eb, x & ((1u << eb) - 1u)};
}

// -------------------------------------------------------------------------------------------
// The chunk of thread t in a member of L bytes.
FQD_HD void chunk_of(uint32_t t, uint32_t L, uint32_t& lo, uint32_t& hi)
{
    const uint32_t right = (kThreads - 1u - t) * kChunk;           // bytes owned by the threads after t
    hi = L > right ? L - right : 0u;
    lo = hi > kChunk ? hi - kChunk : 0u;
}

// The member's bytes as the parser sees them.  In LDS every 128-byte chunk is followed by 4 bytes of
// padding: the 64 lanes of a wave work 128 bytes apart, which unpadded is the same bank for all of them.
constexpr uint32_t kSkewedBytes = kThreads * (kChunk + 4u);
struct Skewed {
    const uint8_t* base;
    FQD_HD static uint32_t at(uint32_t p) { return p + ((p >> 7) << 2); }
    FQD_HD uint8_t operator[](uint32_t p) const { return base[at(p)]; }
    FQD_HD uint32_t word(uint32_t p) const { return *reinterpret_cast<const uint32_t*>(base + at(p)); }   // p a multiple of 4
};
struct Aligned {                                       // plain bytes at a 4-byte aligned base
    const uint8_t* base;
    FQD_HD uint8_t operator[](uint32_t p) const { return base[p]; }
    FQD_HD uint32_t word(uint32_t p) const { return *reinterpret_cast<const uint32_t*>(base + p); }      // p a multiple of 4
};
struct Linear {
    const uint8_t* base;
    FQD_HD uint8_t operator[](uint32_t p) const { return base[p]; }
    FQD_HD uint32_t word(uint32_t p) const
    {
        return uint32_t(base[p]) | (uint32_t(base[p + 1]) << 8) | (uint32_t(base[p + 2]) << 16) | (uint32_t(base[p + 3]) << 24);
    }
};

// Every `sample_every`-th member is parsed for the token histogram once there are enough of them (the codes
// of a call describe FASTQ statistics, which do not change along a file); every symbol then gets a count of
// at least one, so that whatever the other members hold can be written.
FQD_HD uint32_t sample_every(uint64_t members) { return members >= 64u ? 8u : 1u; }

// What a thread knows about its chunk before it parses it, as 128-bit masks (bit i = byte lo + i): which bytes
// equal the byte before them (the runs), which are newlines, and — inside ID lines — which equal the byte
// `delta` back, the same column one record up.  The masks come from 32-bit reads and byte-parallel arithmetic, a
// fixed amount of work for every lane; the parse then jumps from token to token with count-trailing-zeros
// instead of comparing byte by byte in loops whose lengths differ from lane to lane.
struct Mask128 { uint64_t lo, hi; };
struct Scan { Mask128 eq, nl; };
struct Columns {
    Mask128 same; uint32_t delta0, delta1, split;                        // bits below `split` belong to delta0
    Mask128 same_r; uint32_t rdelta0, rdelta1;                           // the same with the two lines held END to end (below)
};

FQD_HD uint32_t zero_bytes(uint32_t x)                               // bit k = byte k of x is zero
{
    uint32_t y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    y = ~(y | x | 0x7F7F7F7Fu);                                      // 0x80 in every zero byte, exactly
    return ((y >> 7) * 0x01020408u) >> 24;
}

template <class Data>
FQD_HD Scan scan_chunk(const Data& data, uint32_t lo, uint32_t hi)
{
    Scan s{{0, 0}, {0, 0}};
    const uint32_t L = hi - lo;
    if (L == 0) return s;
    const uint32_t a = lo & ~3u, sh = (lo & 3u) * 8u;
    uint32_t next = data.word(a);
    uint32_t carry = lo > 0 ? uint32_t(data[lo - 1]) : 0u;
#pragma unroll
    for (uint32_t k = 0; k < kChunk / 4u; ++k) {                     // bytes at and beyond hi are masked off below
        const uint32_t w0 = next;
        next = data.word(a + 4u * k + 4u);
        const uint32_t cur = sh ? (w0 >> sh) | (next << (32u - sh)) : w0;
        const uint64_t eq = zero_bytes(cur ^ ((cur << 8) | carry)), nl = zero_bytes(cur ^ 0x0A0A0A0Au);
        carry = cur >> 24;
        if (k < 16u) { s.eq.lo |= eq << (k * 4u); s.nl.lo |= nl << (k * 4u); }
        else { s.eq.hi |= eq << ((k - 16u) * 4u); s.nl.hi |= nl << ((k - 16u) * 4u); }
    }
    if (lo == 0) s.eq.lo &= ~uint64_t(1);                            // byte 0 has no byte before it
    if (L < 64u) { const uint64_t m = (uint64_t(1) << L) - 1u; s.eq.lo &= m; s.nl.lo &= m; s.eq.hi = s.nl.hi = 0; }
    else if (L < 128u) { const uint64_t m = (uint64_t(1) << (L - 64u)) - 1u; s.eq.hi &= m; s.nl.hi &= m; }
    return s;
}

FQD_HD bool bit_of(const Mask128& m, uint32_t p) { return ((p < 64u ? m.lo >> p : m.hi >> (p - 64u)) & 1u) != 0; }

// Consecutive set bits from bit p on (p < 128).
FQD_HD uint32_t run_from(const Mask128& m, uint32_t p)
{
    if (p >= 64u) { const uint64_t inv = ~(m.hi >> (p - 64u)); return inv ? uint32_t(__builtin_ctzll(inv)) : 64u; }
    const uint64_t v = p ? (m.lo >> p) | (m.hi << (64u - p)) : m.lo;
    const uint64_t inv = ~v;
    if (inv) return uint32_t(__builtin_ctzll(inv));
    const uint64_t inv2 = ~(p ? m.hi >> p : m.hi);                   // bits p + 64 and up
    return 64u + (inv2 ? uint32_t(__builtin_ctzll(inv2)) : 64u);
}

// The column masks of the (at most two) ID lines that reach into [lo, hi): lines whose first byte is '@' or '>',
// compared with the line lines_per_record lines up.  ls[j] = start of line j, n_lines = newlines in the member.
// Twice: the two lines held start to start (`same`) and — when they differ in length — END to end (`same_r`): the
// coordinates in the middle of an Illumina ID have four to six digits, and whatever follows them (" 1:N:0:ACGTACGT") stands a
// column further left or right than in the record before: nine of its bytes in ten were literals (round 4: 20 -> 11 literal
// bytes per ID line, 3.6 % of the output on Illumina-style text).
template <class Data>
FQD_HD Columns column_masks(const Data& data, uint32_t lo, uint32_t hi, const uint16_t* ls, uint32_t line, uint32_t n_lines,
                            uint32_t member_len, bool lines_on, uint32_t lines_per_record)
{
    Columns c{{0, 0}, 0, 0, kChunk, {0, 0}, 0, 0};
    if (!lines_on || lo == hi) return c;
    uint32_t found = 0, j = line;
    for (uint32_t tries = 0; tries < 8u && found < 2u; ++tries, ++j) {
        const uint32_t s = ls[j];
        if (s >= hi) break;
        const uint32_t e = j < n_lines ? uint32_t(ls[j + 1]) : member_len;
        if (j >= lines_per_record && s < member_len) {
            const uint32_t sp = uint32_t(ls[j - lines_per_record]);
            const uint32_t first = data[s], d = s - sp;
            if ((first == uint32_t('@') || first == uint32_t('>')) && d <= 32768u) {
                const uint32_t from = s > lo ? s : lo, to = e < hi ? e : hi;
                for (uint32_t i = from; i < to; ++i) {
                    const uint64_t same = data[i] == data[i - d] ? 1u : 0u;
                    const uint32_t at = i - lo;
                    if (at < 64u) c.same.lo |= same << at; else c.same.hi |= same << (at - 64u);
                }
                // end to end: this line's end is known (its newline lies in the member), the other line's end is ls[j - k + 1]
                const uint32_t dr = j < n_lines ? e - uint32_t(ls[j - lines_per_record + 1u]) : d;
                if (dr != d && dr >= 1u && dr <= 32768u) {
                    const uint32_t least = sp + dr;                          // i - dr stays inside the line above
                    for (uint32_t i = from > least ? from : least; i < to; ++i) {
                        const uint64_t same = data[i] == data[i - dr] ? 1u : 0u;
                        const uint32_t at = i - lo;
                        if (at < 64u) c.same_r.lo |= same << at; else c.same_r.hi |= same << (at - 64u);
                    }
                }
                if (found == 0u) { c.delta0 = d; c.rdelta0 = dr; } else { c.delta1 = d; c.rdelta1 = dr; c.split = from - lo; }
                ++found;
            }
        }
        if (e >= hi) break;
    }
    return c;
}

// Greedy parse of [lo, hi) from the masks.  The sink sees every token in order.
template <class Data, class Sink>
FQD_HD void parse_chunk(const Data& data, uint32_t lo, uint32_t hi, const Scan& sc, const Columns& col, Sink& sink)
{
    const uint32_t L = hi - lo;
    uint32_t p = 0;
    while (p < L) {
        const uint32_t room = L - p;                                 // <= 128 < kMaxMatch
        uint32_t best = 0, dist = 0;
        if (room >= kMinMatch) {
            if (bit_of(sc.eq, p)) {                                  // the run goes on: distance 1
                uint32_t r = run_from(sc.eq, p);
                r = r < room ? r : room;
                if (r >= kMinMatch) { best = r; dist = 1; }
            }
            if (bit_of(col.same, p) && best < room) {                // same column, one record up
                uint32_t r = run_from(col.same, p);
                r = r < room ? r : room;
                const bool first = p < col.split;
                if (first && r > col.split - p) r = col.split - p;   // a run never reaches into the next ID line's distance
                if (r >= kMinMatch && r > best) { best = r; dist = first ? col.delta0 : col.delta1; }
            }
            if (bit_of(col.same_r, p) && best < room) {              // same column counted from the line's end
                uint32_t r = run_from(col.same_r, p);
                r = r < room ? r : room;
                const bool first = p < col.split;
                if (first && r > col.split - p) r = col.split - p;
                if (r >= kMinMatch && r > best) { best = r; dist = first ? col.rdelta0 : col.rdelta1; }
            }
        }
        if (best) { sink.match(best, dist); p += best; }
        else { sink.literal(data[lo + p]); ++p; }
    }
}

// Sink 1: the size of the chunk under a pair of codes (entries: reversed code | length << 16).
struct BitCounter {
    const uint32_t* lit; const uint32_t* dst;
    uint32_t bits = 0;
    FQD_HD void literal(uint32_t b) { bits += lit[b] >> 16; }
    FQD_HD void match(uint32_t len, uint32_t dist)
    {
        const Sym l = length_symbol(len), d = dist_symbol(dist);
        bits += (lit[l.sym] >> 16) + l.ebits + (dst[d.sym] >> 16) + d.ebits;
    }
};

// A bit string written into 32-bit words that other writers share at its two ends: the first and
// last word are OR-ed in (the words start out zero), whole words in between are stored.
template <class Or>
struct BitWriter {
    uint32_t* words; Or or_word;
    uint64_t acc = 0; uint32_t n = 0, w = 0; bool first = true;
    FQD_HD BitWriter(uint32_t* out, uint32_t bit_pos, Or o) : words(out), or_word(o), n(bit_pos & 31u), w(bit_pos >> 5) {}
    FQD_HD void put(uint32_t value, uint32_t nbits)    // nbits <= 32, value < 2^nbits
    {
        acc |= uint64_t(value) << n;
        n += nbits;
        if (n >= 32u) {
            if (first) { or_word(words + w, uint32_t(acc)); first = false; } else words[w] = uint32_t(acc);
            ++w; acc >>= 32; n -= 32u;
        }
    }
    FQD_HD void finish() { if (n) or_word(words + w, uint32_t(acc)); }
};

// Sink 2: the tokens as bits.
template <class Or>
struct Emitter {
    const uint32_t* lit; const uint32_t* dst;
    BitWriter<Or>& out;
    FQD_HD void literal(uint32_t b) { const uint32_t e = lit[b]; out.put(e & 0xFFFFu, e >> 16); }
    FQD_HD void match(uint32_t len, uint32_t dist)
    {
        const Sym l = length_symbol(len), d = dist_symbol(dist);
        const uint32_t el = lit[l.sym], ed = dst[d.sym];
        out.put((el & 0xFFFFu) | (l.eval << (el >> 16)), (el >> 16) + l.ebits);     // <= 15 + 5
        out.put((ed & 0xFFFFu) | (d.eval << (ed >> 16)), (ed >> 16) + d.ebits);     // <= 15 + 13
    }
};

// -------------------------------------------------------------------------------------------
// CRC-32 (the gzip one: reflected 0xEDB88320).
template <class Data>
FQD_HD uint32_t crc_chunk(const uint32_t* table, const Data& data, uint32_t lo, uint32_t hi)
{
    uint32_t reg = (lo == 0u && hi > 0u) ? 0xFFFFFFFFu : 0u;       // the chunk holding byte 0 carries the preset
    uint32_t p = lo;
    for (; p < hi && (p & 3u); ++p) reg = table[(reg ^ data[p]) & 0xFFu] ^ (reg >> 8);
    for (; p + 4u <= hi; p += 4u) {                                 // one read of the text per four bytes
        reg ^= data.word(p);
        reg = table[reg & 0xFFu] ^ (reg >> 8);
        reg = table[reg & 0xFFu] ^ (reg >> 8);
        reg = table[reg & 0xFFu] ^ (reg >> 8);
        reg = table[reg & 0xFFu] ^ (reg >> 8);
    }
    for (; p < hi; ++p) reg = table[(reg ^ data[p]) & 0xFFu] ^ (reg >> 8);
    return reg;
}

FQD_HD uint32_t crc_advance(const uint32_t* matrix, uint32_t reg)   // matrix[b] = image of 1 << b
{
    uint32_t out = 0;
    for (uint32_t b = 0; b < 32u; ++b) out ^= (reg >> b) & 1u ? matrix[b] : 0u;
    return out;
}

// -------------------------------------------------------------------------------------------
// Host side: the two codes of a call from its token histogram.

// Code lengths of a length-limited prefix code for `n` symbols: Huffman's tree by the two-queue
// construction over the symbols sorted by count, depths above `limit` folded back by the usual
// Kraft-sum repair (move one leaf a level down for every unit of excess), lengths then dealt out
// again by rank so that a more frequent symbol never gets a longer code.
inline void build_lengths(const uint64_t* freq, uint32_t n, uint32_t limit, uint8_t* len)
{
    uint32_t order[kLitLen], used = 0;
    for (uint32_t s = 0; s < n; ++s) { len[s] = 0; if (freq[s]) order[used++] = s; }
    if (used == 0) return;
    if (used == 1) { len[order[0]] = 1; return; }
    for (uint32_t i = 1; i < used; ++i) {                            // insertion sort, ascending (count, symbol)
        const uint32_t s = order[i]; uint32_t k = i;
        while (k > 0 && (freq[order[k - 1]] > freq[s] || (freq[order[k - 1]] == freq[s] && order[k - 1] > s))) { order[k] = order[k - 1]; --k; }
        order[k] = s;
    }
    // nodes 0..used-1 leaves (sorted), used.. internal, created in non-decreasing weight order
    uint64_t weight[2 * kLitLen]; uint32_t parent[2 * kLitLen];
    for (uint32_t i = 0; i < used; ++i) weight[i] = freq[order[i]];
    uint32_t leaf = 0, inner = used, made = used;
    auto take = [&]() -> uint32_t {
        if (leaf < used && (inner >= made || weight[leaf] <= weight[inner])) return leaf++;
        return inner++;
    };
    while (made < 2 * used - 1) {
        const uint32_t a = take(), b = take();
        weight[made] = weight[a] + weight[b]; parent[a] = made; parent[b] = made; ++made;
    }
    uint32_t count[64] = {0};
    uint32_t depth[2 * kLitLen];
    depth[made - 1] = 0;
    for (uint32_t i = made - 1; i-- > 0;) depth[i] = depth[parent[i]] + 1;
    for (uint32_t i = 0; i < used; ++i) ++count[depth[i] < limit ? depth[i] : limit];
    uint64_t total = 0;
    for (uint32_t l = 1; l <= limit; ++l) total += uint64_t(count[l]) << (limit - l);
    while (total > (uint64_t(1) << limit)) {
        --count[limit];
        for (uint32_t l = limit - 1; l >= 1; --l) if (count[l]) { --count[l]; count[l + 1] += 2; break; }
        --total;
    }
    uint32_t i = used;                                               // most frequent symbol first
    for (uint32_t l = 1; l <= limit; ++l) for (uint32_t c = 0; c < count[l]; ++c) len[order[--i]] = uint8_t(l);
}

// Canonical codes (RFC 1951 §3.2.2), bit-reversed for the writer.
inline void assign_codes(const uint8_t* len, uint32_t n, uint16_t* code)
{
    uint32_t count[16] = {0}, next[16] = {0};
    for (uint32_t s = 0; s < n; ++s) ++count[len[s]];
    count[0] = 0;
    for (uint32_t l = 1, c = 0; l < 16; ++l) { c = (c + count[l - 1]) << 1; next[l] = c; }
    for (uint32_t s = 0; s < n; ++s) {
        code[s] = 0;
        if (!len[s]) continue;
        const uint32_t c = next[len[s]]++;
        uint32_t r = 0;
        for (uint32_t b = 0; b < len[s]; ++b) r |= ((c >> b) & 1u) << (len[s] - 1u - b);
        code[s] = uint16_t(r);
    }
}

// hist: kLitLen literal/length counts followed by kDist distance counts (end-of-block NOT included:
// it is counted here, once per member).
inline void build_codes(const uint64_t* hist, uint64_t members, Codes& c)
{
    uint64_t lit[kLitLen], dst[kDist];
    const uint64_t floor = sample_every(members) > 1u ? 1u : 0u;      // a sampled histogram must not leave a symbol without a code
    for (uint32_t s = 0; s < kLitLen; ++s) lit[s] = hist[s] + floor;
    for (uint32_t s = 0; s < kDist; ++s) dst[s] = hist[kLitLen + s] + floor;
    lit[256] = members ? members : 1;
    bool any = false;
    for (uint32_t s = 0; s < 256; ++s) any |= lit[s] != 0;
    for (uint32_t s = 257; s < kLitLen; ++s) any |= lit[s] != 0;
    if (!any) lit[0] = 1;                                            // a complete code needs two symbols
    uint8_t lit_len[kLitLen], dist_len[kDist];
    uint16_t lit_code[kLitLen], dist_code[kDist];
    build_lengths(lit, kLitLen, 15, lit_len);
    build_lengths(dst, kDist, 15, dist_len);
    bool any_dist = false;
    for (uint32_t s = 0; s < kDist; ++s) any_dist |= dist_len[s] != 0;
    if (!any_dist) dist_len[0] = 1;                                  // one unused 1-bit code: §3.2.7 allows it, every inflater takes it
    assign_codes(lit_len, kLitLen, lit_code);
    assign_codes(dist_len, kDist, dist_code);
    for (uint32_t s = 0; s < kLitLen; ++s) c.lit[s] = uint32_t(lit_code[s]) | (uint32_t(lit_len[s]) << 16);
    for (uint32_t s = 0; s < kDist; ++s) c.dist[s] = uint32_t(dist_code[s]) | (uint32_t(dist_len[s]) << 16);

    uint32_t nlit = kLitLen, ndist = kDist;
    while (nlit > 257 && lit_len[nlit - 1] == 0) --nlit;
    while (ndist > 1 && dist_len[ndist - 1] == 0) --ndist;
    for (uint32_t& w : c.header) w = 0;
    uint32_t at = 0;
    auto put = [&](uint32_t value, uint32_t nbits) {
        for (uint32_t b = 0; b < nbits; ++b, ++at) c.header[at >> 5] |= ((value >> b) & 1u) << (at & 31u);
    };
    put(1, 1); put(2, 2);                                            // BFINAL, BTYPE = dynamic
    put(nlit - 257, 5); put(ndist - 1, 5); put(15, 4);               // HCLEN: all 19 entries
    // code-length alphabet: symbols 0..15 with 4 bits each (a complete code), no repeat symbols
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (uint32_t k = 0; k < 19; ++k) put(order[k] < 16 ? 4u : 0u, 3);
    auto put_len = [&](uint32_t l) { uint32_t r = 0; for (uint32_t b = 0; b < 4; ++b) r |= ((l >> b) & 1u) << (3u - b); put(r, 4); };
    for (uint32_t s = 0; s < nlit; ++s) put_len(lit_len[s]);
    for (uint32_t s = 0; s < ndist; ++s) put_len(dist_len[s]);
    c.header_bits = at;

    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (r & 1u ? 0xEDB88320u : 0u);
        c.crc_table[i] = r;
    }
    for (uint32_t b = 0; b < 32; ++b) {
        uint32_t reg = 1u << b;
        for (uint32_t k = 0; k < kChunk; ++k) reg = c.crc_table[reg & 0xFFu] ^ (reg >> 8);
        c.crc_shift[0][b] = reg;
    }
    for (uint32_t k = 1; k < kLevels; ++k)
        for (uint32_t b = 0; b < 32; ++b) c.crc_shift[k][b] = crc_advance(c.crc_shift[k - 1], c.crc_shift[k - 1][b]);
}

} // namespace bgzf
} // namespace fqd
