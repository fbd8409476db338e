// fqd_join.hip — the `--unordered` read-ID join on the GPU (same library as fqd_engine.hip).
//
// The reference sorts both files by ID tag with an on-disk merge sort (ExternalSorter<T>,
// external_sort.hpp:66-215; order = FastqViewWithId::cmp, fastqview.cpp:168-178: bytes over the
// shorter length, then shorter first) and merge-joins the sorted files
// (hash_dup_remover.hpp:279-340); matched pairs are deduplicated and written in tag order.
// Because the OUTPUT is in tag order, the tags have to be ordered whatever finds the matches;
// here one sort does both jobs, and every kernel is hand-written for gfx950:
//
//   1. extract   tag of every record from its ID line where it lies in the uploaded text
//                (FastqViewWithId::read_new, fastqview.cpp:190-204)
//   2. census    which byte values occur at which tag position, over BOTH files (LDS bitmaps)
//   3. compact   an order-preserving dense code per position: rank of the byte among the values
//                seen there (+ one code for "tag has ended"), ceil(log2) bits wide, ZERO bits for a
//                position where every tag agrees — so "@A00123:45:HXXXXXXX:1:" costs nothing and
//                "r000123456" becomes 9 x 4 bits.  Concatenated big-endian the codes form an
//                integer key of B bits whose order IS the reference's tag order
//   4. sort      the records of both files together (file 1's first) by that key with a stable
//                LSD radix sort, 8 bits per pass, 64 key bits per word: per pass a tile
//                histogram, a per-digit scan and a stable scatter (ranks by wave ballots)
//   5. join      in the sorted union equal tags are neighbours, file 1's records first: the k-th
//                record of file 1 with a tag pairs with the k-th of file 2 with that tag — what the
//                merge-join does — found with three forward scans (records of file 1 so far, start
//                of the run, start of file 2's part of the run); pairs are compacted in tag order
//
// Nothing in the pass loop waits for the host: the host reads back the tag length range (to size
// the census) and the key width B (to know how many passes to launch) before the first pass, and
// the number of pairs after the last kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "../../include/fqdupaway.h"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out);

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kFileB = 0x80000000u;             // payload of a record of file 2: kFileB | index

#define JOIN_TRY(e, expr)                                                                   \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess) { (void)hipGetLastError();       \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } } while (0)

inline uint32_t grid_for(uint64_t n, uint32_t per_block = kBlock, uint32_t cap = 4096)
{
    return uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((n + per_block - 1) / per_block, cap)));
}

// ---------------------------------------------------------------------------------------------
// 1. tag extraction (fastqview.cpp:190-204; fastaview.cpp:153-167 is the same rule): the tag
// starts after the first '.' of the ID line (searched over the whole line, newline included), or
// after the leading '@' / '>' when there is none, and ends before the first ' ' at or after its
// start, or else runs through the end of the line INCLUDING the newline.
// First k in [from, L) with p[k] == c, else L: eight bytes per load while eight are left (one thread reads one ID line;
// byte by byte that is one scattered load instruction per byte of the line for the wave).
__device__ __forceinline__ uint32_t first_byte(const uint8_t* __restrict__ p, uint32_t from, uint32_t L, uint8_t c)
{
    const uint64_t ones = 0x0101010101010101ull, pattern = ones * c;
    uint32_t k = from;
    for (; k + 8u <= L; k += 8u) {
        uint64_t v;
        __builtin_memcpy(&v, p + k, 8);
        const uint64_t x = v ^ pattern, hit = (x - ones) & ~x & (ones << 7);       // the lowest set bit marks the first equal byte
        if (hit) return k + uint32_t(__builtin_ctzll(hit)) / 8u;
    }
    for (; k < L; ++k) if (p[k] == c) return k;
    return L;
}

__global__ __launch_bounds__(kBlock)
void extract_tags_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ id_start,
                         const uint32_t* __restrict__ id_len, uint64_t n,
                         uint64_t* __restrict__ tag_off, uint32_t* __restrict__ tag_len)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const uint64_t s = id_start[i];
        const uint32_t L = id_len[i];
        const uint8_t* __restrict__ p = text + s;
        const uint32_t dot = first_byte(p, 0, L, uint8_t('.'));
        const uint32_t from = dot < L ? dot + 1u : (L ? 1u : 0u);
        const uint32_t to = first_byte(p, from, L, uint8_t(' '));
        tag_off[i] = s + from;
        tag_len[i] = to > from ? to - from : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// The union of both files' tags: element u < n_a is record u of file 1, else record u - n_a of file 2.
struct Union {
    const uint8_t*  bytes_a; const uint64_t* off_a; const uint32_t* len_a; uint64_t n_a;
    const uint8_t*  bytes_b; const uint64_t* off_b; const uint32_t* len_b; uint64_t n_b;
    __device__ __forceinline__ const uint8_t* tag(uint64_t u, uint32_t& len) const
    {
        if (u < n_a) { len = len_a[u]; return bytes_a + off_a[u]; }
        u -= n_a; len = len_b[u]; return bytes_b + off_b[u];
    }
    __device__ __forceinline__ const uint8_t* tag_of_payload(uint32_t v, uint32_t& len) const
    {
        if (v & kFileB) { const uint32_t r = v & ~kFileB; len = len_b[r]; return bytes_b + off_b[r]; }
        len = len_a[v]; return bytes_a + off_a[v];
    }
};

__global__ __launch_bounds__(kBlock)
void len_range_kernel(Union u, unsigned int* __restrict__ min_max)
{
    unsigned int lo = 0xFFFFFFFFu, hi = 0;
    const uint64_t N = u.n_a + u.n_b;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < N; i += uint64_t(gridDim.x) * kBlock) {
        const uint32_t L = i < u.n_a ? u.len_a[i] : u.len_b[i - u.n_a];
        lo = min(lo, L); hi = max(hi, L);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { lo = min(lo, __shfl_down(lo, d, 64)); hi = max(hi, __shfl_down(hi, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&min_max[0], lo); atomicMax(&min_max[1], hi); }
}

// 2. census: bitmap[p][c] set <=> some tag has byte c at position p.  The first kCensusLds
// positions are collected in LDS (a bit is tested before it is set: after the first few tags almost
// every test hits, and tests of one address are a broadcast), later positions go to global memory.
constexpr uint32_t kCensusLds = 160;
__global__ __launch_bounds__(kBlock)
void census_kernel(Union u, uint32_t max_len, uint32_t* __restrict__ bitmap /* [max_len][8] */)
{
    __shared__ uint32_t lb[kCensusLds][8];
    for (uint32_t k = threadIdx.x; k < kCensusLds * 8u; k += kBlock) (&lb[0][0])[k] = 0;
    __syncthreads();
    const uint64_t N = u.n_a + u.n_b;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < N; i += uint64_t(gridDim.x) * kBlock) {
        uint32_t L;
        const uint8_t* __restrict__ p = u.tag(i, L);
        for (uint32_t k = 0; k < L; ++k) {
            const uint32_t c = p[k], w = c >> 5, bit = 1u << (c & 31u);
            if (k < kCensusLds) { if (!(lb[k][w] & bit)) atomicOr(&lb[k][w], bit); }
            else if (!(bitmap[k * 8u + w] & bit)) atomicOr(&bitmap[k * 8u + w], bit);
        }
    }
    __syncthreads();
    const uint32_t lim = (max_len < kCensusLds ? max_len : kCensusLds) * 8u;
    for (uint32_t k = threadIdx.x; k < lim; k += kBlock) { const uint32_t v = (&lb[0][0])[k]; if (v) atomicOr(&bitmap[k], v); }
}

// 3. the code table (one block).  Position p has the codes 0 = "tag ended before p" (only where
// p >= min_len) and then the byte values present, in byte order; width[p] = bits needed, 0 when
// there is a single code.  The key is the concatenation, position 0 most significant, right
// aligned in B = sum(width) bits; lsb[p] = bit position of p's field counted from the key's least
// significant bit.  Fields of positions with width 0 do not exist.  vary[] lists the positions
// with width > 0 in order, which is what the encoder walks.
struct CodeTable {
    uint8_t*  rank;     // [max_len][256] code of byte c at position p (valid where the byte occurs)
    uint32_t* width;    // [max_len]
    uint32_t* lsb;      // [max_len]
    uint32_t* vary;     // [max_len] positions with width > 0, ascending; n_vary of them
    uint32_t* info;     // [0] = B (total bits), [1] = n_vary
};

__global__ __launch_bounds__(1024)
void build_codes_kernel(const uint32_t* __restrict__ bitmap, uint32_t min_len, uint32_t max_len, CodeTable t)
{
    // widths and rank rows, positions strided over the block.  rank[p][c] counts the byte values
    // below c that occur at p (so it fits a byte even when all 256 occur); the encoder adds 1
    // where the END code exists (p >= min_len).
    for (uint32_t p = threadIdx.x; p < max_len; p += blockDim.x) {
        uint32_t below = 0;
        for (uint32_t w = 0; w < 8u; ++w) {
            const uint32_t m = bitmap[p * 8u + w];
            for (uint32_t b = 0; b < 32u; ++b) {
                t.rank[p * 256u + w * 32u + b] = uint8_t(below);
                below += (m >> b) & 1u;
            }
        }
        const uint32_t codes = below + (p >= min_len ? 1u : 0u);
        t.width[p] = codes <= 1u ? 0u : 32u - __clz(codes - 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                           // max_len is small (tens to hundreds): one lane walks it
        uint32_t bits = 0, nv = 0;
        for (uint32_t p = max_len; p-- > 0;) { t.lsb[p] = bits; bits += t.width[p]; }
        for (uint32_t p = 0; p < max_len; ++p) if (t.width[p]) t.vary[nv++] = p;
        t.info[0] = bits; t.info[1] = nv;
    }
}

// 4a. key word `word` (0 = least significant 64 bits) of the record behind each payload.
// The block caches the rank rows of the positions that reach into this word.
constexpr uint32_t kMaxRows = 64;                     // a 64-bit word holds at most 64 fields (+1 straddling in): see below
__global__ __launch_bounds__(kBlock)
void encode_word_kernel(Union u, const uint32_t* __restrict__ payload, uint64_t N, uint32_t word, uint32_t min_len,
                        CodeTable t, uint64_t* __restrict__ keys)
{
    __shared__ uint8_t  rows[kMaxRows + 2][256];
    __shared__ uint32_t pos_of[kMaxRows + 2], lsb_of[kMaxRows + 2], wid_of[kMaxRows + 2];
    __shared__ uint32_t n_rows;
    const uint32_t lo_bit = word * 64u, hi_bit = lo_bit + 64u;
    if (threadIdx.x == 0) {
        // fields overlapping [lo_bit, hi_bit): vary[] is ascending in position = descending in lsb
        const uint32_t nv = t.info[1];
        uint32_t k = 0;
        for (uint32_t j = 0; j < nv; ++j) {
            const uint32_t p = t.vary[j], l = t.lsb[p], w = t.width[p];
            if (l < hi_bit && l + w > lo_bit && k < kMaxRows + 2) { pos_of[k] = p; lsb_of[k] = l; wid_of[k] = w; ++k; }
        }
        n_rows = k;
    }
    __syncthreads();
    const uint32_t R = n_rows;
    for (uint32_t x = threadIdx.x; x < R * 64u; x += kBlock) {               // 4 bytes per access
        const uint32_t r = x >> 6, c4 = x & 63u;
        reinterpret_cast<uint32_t*>(rows[r])[c4] = reinterpret_cast<const uint32_t*>(t.rank + size_t(pos_of[r]) * 256u)[c4];
    }
    __syncthreads();
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < N; i += uint64_t(gridDim.x) * kBlock) {
        uint32_t L;
        const uint8_t* __restrict__ p = u.tag_of_payload(payload[i], L);
        uint64_t key = 0;
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t pos = pos_of[r];
            uint64_t code = 0;                                                // END
            if (pos < L) code = uint64_t(rows[r][p[pos]]) + (pos >= min_len ? 1u : 0u);
            const uint32_t l = lsb_of[r];
            key |= l >= lo_bit ? (code << (l - lo_bit)) : (code >> (lo_bit - l));   // bits beyond 64 fall off the top
        }
        keys[i] = key;
    }
}

__global__ __launch_bounds__(kBlock)
void iota_payload_kernel(uint32_t* __restrict__ payload, uint64_t n_a, uint64_t n_b)
{
    const uint64_t N = n_a + n_b;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < N; i += uint64_t(gridDim.x) * kBlock)
        payload[i] = i < n_a ? uint32_t(i) : (kFileB | uint32_t(i - n_a));
}

// ---------------------------------------------------------------------------------------------
// 4b. one stable radix pass over (key, payload): digit = (key >> shift) & mask, 256 bins.
//   hist    : counts[d][tile] per tile of kSortTile elements
//   rowscan : one block per digit: exclusive scan of its row in place, row total to tot[d]
//   scatter : tile re-read; a wave owns kSortTile/4 consecutive elements and takes them 64 at a
//             time, so (wave, round, lane) is input order; lanes with the same digit find each
//             other with eight ballots, their rank is the count of earlier lanes plus the wave's
//             running count for that digit; destinations = digit base + scanned tile count +
//             counts of the earlier waves + rank.  Stable by construction, no atomics.
constexpr int kSortPer = 16;
constexpr int kSortTile = kBlock * kSortPer;          // 4096

__global__ __launch_bounds__(kBlock)
void radix_hist_kernel(const uint64_t* __restrict__ keys, uint64_t N, uint32_t shift, uint32_t mask,
                       uint32_t* __restrict__ counts, uint32_t n_tiles)
{
    __shared__ uint32_t h[256];
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t base = uint64_t(tile) * kSortTile;
        uint64_t k[kSortPer];
#pragma unroll
        for (int e = 0; e < kSortPer; ++e) { const uint64_t i = base + uint32_t(e) * kBlock + threadIdx.x; k[e] = keys[i < N ? i : N - 1]; }
#pragma unroll
        for (int e = 0; e < kSortPer; ++e)
            if (base + uint32_t(e) * kBlock + threadIdx.x < N) atomicAdd(&h[uint32_t(k[e] >> shift) & mask], 1u);
        __syncthreads();
        counts[uint64_t(threadIdx.x) * n_tiles + tile] = h[threadIdx.x];
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024)
void radix_rowscan_kernel(uint32_t* __restrict__ counts, uint32_t n_tiles, uint32_t* __restrict__ tot)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    uint32_t* row = counts + uint64_t(blockIdx.x) * n_tiles;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += 1024u) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < n_tiles ? row[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(threadIdx.x & 63u) >= d) inc += up; }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
        if (i < n_tiles) row[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(kBlock)
void radix_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t N,
                          uint32_t shift, uint32_t mask, const uint32_t* __restrict__ counts, uint32_t n_tiles,
                          const uint32_t* __restrict__ tot, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t cnt[4][256];                   // per wave, per digit: running count, then start
    __shared__ uint32_t dbase[256];
    __shared__ uint32_t wsum[4];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    // digit bases: exclusive scan of tot[0..255] (every block repeats it: 256 values)
    {
        const uint32_t v = tot[t];
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
        if (lane == 63u) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        dbase[t] = before + inc - v;
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int w = 0; w < 4; ++w) cnt[w][t] = 0;
        __syncthreads();
        const uint64_t base = uint64_t(tile) * kSortTile + uint64_t(wave) * (64u * kSortPer);
        uint64_t k[kSortPer]; uint32_t v[kSortPer], rk[kSortPer];
#pragma unroll
        for (int e = 0; e < kSortPer; ++e) {
            const uint64_t i = base + uint32_t(e) * 64u + lane;
            k[e] = keys[i < N ? i : N - 1]; v[e] = vals[i < N ? i : N - 1];
        }
#pragma unroll
        for (int e = 0; e < kSortPer; ++e) {
            const bool live = base + uint32_t(e) * 64u + lane < N;
            const uint32_t d = uint32_t(k[e] >> shift) & mask;
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long vote = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? vote : ~vote;
            }
            const uint32_t prev = cnt[wave][d];          // every lane reads before the leader below writes (one wave, in order)
            rk[e] = prev + uint32_t(__popcll(peers & lt_mask));
            if (live && (peers & lt_mask) == 0) cnt[wave][d] = prev + uint32_t(__popcll(peers));
        }
        __syncthreads();
        {   // digit t: where each wave's elements start
            uint32_t run = dbase[t] + counts[uint64_t(t) * n_tiles + tile];
#pragma unroll
            for (int w = 0; w < 4; ++w) { const uint32_t c = cnt[w][t]; cnt[w][t] = run; run += c; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kSortPer; ++e) {
            if (base + uint32_t(e) * 64u + lane < N) {
                const uint32_t d = uint32_t(k[e] >> shift) & mask;
                const uint32_t at = cnt[wave][d] + rk[e];
                keys_out[at] = k[e]; vals_out[at] = v[e];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// 5. the join over the sorted union.
// head[p] = 1 when element p starts a run of equal tags.  One-word keys ARE the tags' order and
// identity; with longer keys the sorted array only holds the most significant word, so the tags
// themselves are compared.
__global__ __launch_bounds__(kBlock)
void heads_kernel(Union u, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t N,
                  uint32_t key_words, uint8_t* __restrict__ head)
{
    for (uint64_t p = blockIdx.x * uint64_t(kBlock) + threadIdx.x; p < N; p += uint64_t(gridDim.x) * kBlock) {
        uint8_t h = 1;
        if (p > 0) {
            if (key_words <= 1 || keys[p] != keys[p - 1]) h = keys[p] != keys[p - 1];   // the (most significant) key words differ: another tag
            else {
                uint32_t la, lb;
                const uint8_t* __restrict__ a = u.tag_of_payload(vals[p - 1], la);
                const uint8_t* __restrict__ b = u.tag_of_payload(vals[p], lb);
                if (la == lb) {
                    // 32 bytes of each tag per step, all eight loads issued before the first compare (a
                    // byte-at-a-time loop pays one memory round trip per byte: 92 ms per 100 M 47-byte tags)
                    uint64_t d = 0;
                    uint32_t k = 0;
                    for (; k + 32u <= la && d == 0; k += 32u) {
                        uint64_t x[4], y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { __builtin_memcpy(&x[j], a + k + 8 * j, 8); __builtin_memcpy(&y[j], b + k + 8 * j, 8); }
#pragma unroll
                        for (int j = 0; j < 4; ++j) d |= x[j] ^ y[j];
                    }
                    for (; k + 8u <= la && d == 0; k += 8u) { uint64_t x, y; __builtin_memcpy(&x, a + k, 8); __builtin_memcpy(&y, b + k, 8); d |= x ^ y; }
                    for (; k < la && d == 0; ++k) d |= uint64_t(a[k] ^ b[k]);
                    h = d != 0;
                }
            }
        }
        head[p] = h;
    }
}

// Tile summaries, then their scan by one block, then the per-element pass.  Positions are kept
// as position + 1 (0 = none yet), so "latest position" is a running maximum.
constexpr int kJoinPer = 8;
constexpr int kJoinTile = kBlock * kJoinPer;          // 2048
struct JoinCarry { uint32_t n_a; uint32_t head1; uint32_t bhead1; uint32_t pad; };   // of everything BEFORE the tile (after the scan)

__device__ __forceinline__ bool is_b(uint32_t v) { return (v & kFileB) != 0; }

__global__ __launch_bounds__(kBlock)
void join_summary_kernel(const uint32_t* __restrict__ vals, const uint8_t* __restrict__ head, uint64_t N, JoinCarry* __restrict__ sums)
{
    __shared__ uint32_t s_a[4], s_h[4], s_b[4];
    const uint64_t p0 = uint64_t(blockIdx.x) * kJoinTile + uint64_t(threadIdx.x) * kJoinPer;
    uint32_t na = 0, h1 = 0, b1 = 0;
    bool prev_a = p0 > 0 && p0 - 1 < N ? !is_b(vals[p0 - 1]) : false;
#pragma unroll
    for (int e = 0; e < kJoinPer; ++e) {
        const uint64_t p = p0 + uint32_t(e);
        if (p < N) {
            const bool b = is_b(vals[p]); const bool h = head[p] != 0;
            na += b ? 0u : 1u;
            if (h) h1 = uint32_t(p) + 1u;
            if (b && (h || prev_a)) b1 = uint32_t(p) + 1u;
            prev_a = !b;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { na += __shfl_down(na, d, 64); h1 = max(h1, __shfl_down(h1, d, 64)); b1 = max(b1, __shfl_down(b1, d, 64)); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = na; s_h[threadIdx.x >> 6] = h1; s_b[threadIdx.x >> 6] = b1; }
    __syncthreads();
    if (threadIdx.x == 0)
        sums[blockIdx.x] = JoinCarry{s_a[0] + s_a[1] + s_a[2] + s_a[3], max(max(s_h[0], s_h[1]), max(s_h[2], s_h[3])),
                                     max(max(s_b[0], s_b[1]), max(s_b[2], s_b[3])), 0u};
}

// One block: sums[t] <- (sum, max, max) over tiles before t.
__global__ __launch_bounds__(1024)
void join_carry_scan_kernel(JoinCarry* __restrict__ sums, uint32_t n_tiles)
{
    __shared__ uint32_t w_a[16], w_h[16], w_b[16];
    __shared__ uint32_t c_a, c_h, c_b;
    if (threadIdx.x == 0) { c_a = 0; c_h = 0; c_b = 0; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024u) {
        const uint32_t t = t0 + threadIdx.x;
        const JoinCarry v = t < n_tiles ? sums[t] : JoinCarry{0, 0, 0, 0};
        uint32_t a = v.n_a, h = v.head1, b = v.bhead1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ua = __shfl_up(a, d, 64), uh = __shfl_up(h, d, 64), ub = __shfl_up(b, d, 64);
            if (int(lane) >= d) { a += ua; h = max(h, uh); b = max(b, ub); }
        }
        if (lane == 63u) { w_a[wave] = a; w_h[wave] = h; w_b[wave] = b; }
        __syncthreads();
        uint32_t ba = c_a, bh = c_h, bb = c_b;
        for (uint32_t w = 0; w < wave; ++w) { ba += w_a[w]; bh = max(bh, w_h[w]); bb = max(bb, w_b[w]); }
        // exclusive: what came before this tile = block carry + earlier waves + earlier lanes
        const uint32_t ea = __shfl_up(a, 1, 64), eh = __shfl_up(h, 1, 64), eb = __shfl_up(b, 1, 64);
        const uint32_t xa = ba + (lane ? ea : 0u), xh = max(bh, lane ? eh : 0u), xb = max(bb, lane ? eb : 0u);
        if (t < n_tiles) sums[t] = JoinCarry{xa, xh, xb, 0u};
        __syncthreads();
        if (threadIdx.x == 1023u) { c_a = ba + a; c_h = max(bh, h); c_b = max(bb, b); }
        __syncthreads();
    }
}

// perm_a[i] / perm_b[j]: record index of the i-th / j-th smallest tag of file 1 / file 2 (equal
// tags in input order); match_a[i] = j of the partner or kNone, match_b[j] = i or kNone.
// match_a is preset to kNone by the host code.
__global__ __launch_bounds__(kBlock)
void join_emit_kernel(const uint32_t* __restrict__ vals, const uint8_t* __restrict__ head, uint64_t N,
                      const JoinCarry* __restrict__ carry, uint32_t* __restrict__ perm_a, uint32_t* __restrict__ perm_b,
                      uint32_t* __restrict__ match_a, uint32_t* __restrict__ match_b)
{
    __shared__ uint32_t w_a[4], w_h[4], w_b[4];
    const uint64_t p0 = uint64_t(blockIdx.x) * kJoinTile + uint64_t(threadIdx.x) * kJoinPer;
    uint32_t val[kJoinPer]; uint8_t hd[kJoinPer];
    uint32_t na = 0, h1 = 0, b1 = 0;
    const bool first_prev_a = p0 > 0 && p0 - 1 < N ? !is_b(vals[p0 - 1]) : false;
    bool prev_a = first_prev_a;
#pragma unroll
    for (int e = 0; e < kJoinPer; ++e) {
        const uint64_t p = p0 + uint32_t(e);
        val[e] = 0; hd[e] = 0;
        if (p < N) {
            val[e] = vals[p]; hd[e] = head[p];
            const bool b = is_b(val[e]);
            na += b ? 0u : 1u;
            if (hd[e]) h1 = uint32_t(p) + 1u;
            if (b && (hd[e] || prev_a)) b1 = uint32_t(p) + 1u;
            prev_a = !b;
        }
    }
    // exclusive scan of (na, h1, b1) over the block's threads
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t a = na, h = h1, b = b1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ua = __shfl_up(a, d, 64), uh = __shfl_up(h, d, 64), ub = __shfl_up(b, d, 64);
        if (int(lane) >= d) { a += ua; h = max(h, uh); b = max(b, ub); }
    }
    if (lane == 63u) { w_a[wave] = a; w_h[wave] = h; w_b[wave] = b; }
    __syncthreads();
    const JoinCarry c = carry[blockIdx.x];
    uint32_t ra = c.n_a, rh = c.head1, rb = c.bhead1;
    for (uint32_t w = 0; w < wave; ++w) { ra += w_a[w]; rh = max(rh, w_h[w]); rb = max(rb, w_b[w]); }
    const uint32_t ea = __shfl_up(a, 1, 64), eh = __shfl_up(h, 1, 64), eb = __shfl_up(b, 1, 64);
    if (lane) { ra += ea; rh = max(rh, eh); rb = max(rb, eb); }
    // replay the thread's elements with the running state
    prev_a = first_prev_a;
#pragma unroll
    for (int e = 0; e < kJoinPer; ++e) {
        const uint64_t p = p0 + uint32_t(e);
        if (p < N) {
            const bool isb = is_b(val[e]);
            if (hd[e]) rh = uint32_t(p) + 1u;
            if (isb && (hd[e] || prev_a)) rb = uint32_t(p) + 1u;
            if (!isb) {
                perm_a[ra] = val[e];
                ++ra;
            } else {
                const uint32_t pos_b = uint32_t(p) - ra;                 // records of file 2 before p
                perm_b[pos_b] = val[e] & ~kFileB;
                const uint32_t run_start = rh - 1u, first_b = rb - 1u;   // rb >= rh here: the run's first record of file 2 is a b-head
                const uint32_t in_a = first_b - run_start;               // records of file 1 in this run
                const uint32_t s = uint32_t(p) - first_b;                // my rank among the run's records of file 2
                uint32_t partner = kNone;
                if (s < in_a) { partner = ra - in_a + s; match_a[partner] = pos_b; }
                match_b[pos_b] = partner;
            }
            prev_a = !isb;
        }
    }
}

// Pairs in tag order: compaction of the matched records of file 1 over sorted positions.
constexpr int kPairTile = kBlock * 8;
__global__ __launch_bounds__(kBlock)
void pair_count_kernel(const uint32_t* __restrict__ match_a, uint64_t n_a, uint32_t* __restrict__ tile_count)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kPairTile;
    uint32_t c = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const uint64_t i = base + uint32_t(e) * kBlock + threadIdx.x; c += (i < n_a && match_a[i] != kNone) ? 1u : 0u; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// One block: exclusive scan of tile_count in place; the total goes to *total (u64).
__global__ __launch_bounds__(1024)
void u32_scan_kernel(uint32_t* __restrict__ data, uint32_t n, unsigned long long* __restrict__ total)
{
    __shared__ uint32_t wt[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t t0 = 0; t0 < n; t0 += 1024u) {
        const uint32_t i = t0 + threadIdx.x;
        const uint32_t v = i < n ? data[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
        if (lane == 63u) wt[wave] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wt[w];
        if (i < n) data[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kBlock)
void pair_emit_kernel(const uint32_t* __restrict__ match_a, const uint32_t* __restrict__ perm_a, const uint32_t* __restrict__ perm_b,
                      uint64_t n_a, const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ pair_a, uint32_t* __restrict__ pair_b)
{
    __shared__ uint32_t ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kPairTile;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t at = tile_start[blockIdx.x];
    for (int e = 0; e < 8; ++e) {                      // element order inside the tile: round e, then thread
        const uint64_t i = base + uint32_t(e) * kBlock + threadIdx.x;
        const uint32_t m = i < n_a ? match_a[i] : kNone;
        const unsigned long long vote = __ballot(m != kNone);
        if (lane == 0) ws[wave] = uint32_t(__popcll(vote));
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4u; ++w) { const uint32_t c = ws[w]; before += w < wave ? c : 0u; total += c; }
        if (m != kNone) {
            const uint32_t k = at + before + uint32_t(__popcll(vote & ((1ull << lane) - 1ull)));
            pair_a[k] = perm_a[i]; pair_b[k] = perm_b[m];
        }
        at += total;
        __syncthreads();
    }
}

// out[k] = table[idx[k]] for the two per-record arrays a dedup batch needs (sequence offset, length).
__global__ __launch_bounds__(kBlock)
void gather_seq_kernel(const uint32_t* __restrict__ idx, uint64_t n, const uint64_t* __restrict__ off_table,
                       const uint32_t* __restrict__ len_table, uint64_t* __restrict__ off_out, uint32_t* __restrict__ len_out)
{
    for (uint64_t k = blockIdx.x * uint64_t(kBlock) + threadIdx.x; k < n; k += uint64_t(gridDim.x) * kBlock) {
        const uint32_t r = idx[k];
        off_out[k] = off_table[r]; len_out[k] = len_table[r];
    }
}

// ---------------------------------------------------------------------------------------------
// Support kernels of the bounded-memory `--unordered` run (host/run_unordered.cpp,
// run_unordered_streaming): the host streams both files through small pinned blocks; what the join
// and the dedup need of every record — its tag and its sequence — is copied out of the uploaded block
// into stores that stay in HBM, and where every surviving record goes in the output is computed here.

// dst[dst_off[i] .. +len[i]) = src[src_off[i] .. +len[i]).  EIGHT LANES per span, sixteen bytes each, side by side: a
// record is a few hundred bytes, and a lane that copies one alone asks for a line of its own with every load and store
// (64 requests per instruction to the CU's one address unit; tools/vmem_probe.hip) — 277 GB/s; eight lanes per record
// ask for two or three lines between them.  The last sixteen bytes are copied again if the length is no multiple of
// sixteen (same bytes, same place); a span shorter than sixteen goes byte by byte.
constexpr uint32_t kSpanLanes = 8;
__global__ __launch_bounds__(kBlock)
void copy_spans_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ len,
                       uint64_t n, uint8_t* __restrict__ dst, const uint64_t* __restrict__ dst_off)
{
    const uint32_t l = threadIdx.x % kSpanLanes;
    for (uint64_t i = (blockIdx.x * uint64_t(kBlock) + threadIdx.x) / kSpanLanes; i < n; i += uint64_t(gridDim.x) * (kBlock / kSpanLanes)) {
        const uint8_t* __restrict__ a = src + src_off[i];
        uint8_t* __restrict__ b = dst + dst_off[i];
        const uint32_t L = len[i];
        if (L < 16u) { for (uint32_t k = l; k < L; k += kSpanLanes) b[k] = a[k]; continue; }
        for (uint32_t k = 16u * l; k + 16u <= L; k += 16u * kSpanLanes) {
            uint64_t w[2];
            __builtin_memcpy(w, a + k, 16);
            __builtin_memcpy(b + k, w, 16);
        }
        if ((L & 15u) && l == ((L / 16u) % kSpanLanes)) {              // (the lane whose turn the next chunk would have been)
            uint64_t w[2];
            __builtin_memcpy(w, a + L - 16u, 16);
            __builtin_memcpy(b + L - 16u, w, 16);
        }
    }
}

// Number of tags of `t` that are <= the tag `probe` (FastqViewWithId::cmp order).
__global__ __launch_bounds__(kBlock)
void count_le_kernel(fqd_tags t, const uint8_t* __restrict__ probe, uint32_t probe_len, unsigned long long* __restrict__ count)
{
    uint32_t c = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < t.n; i += uint64_t(gridDim.x) * kBlock) {
        const uint8_t* __restrict__ a = t.bytes + t.offsets[i];
        const uint32_t L = t.lengths[i], m = L < probe_len ? L : probe_len;
        uint32_t k = 0;
        while (k < m && a[k] == probe[k]) ++k;
        const bool le = k < m ? a[k] < probe[k] : L <= probe_len;
        c += le ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, static_cast<unsigned long long>(c));
}

// ---- `--unordered` over several GPUs (host: run_unordered_multi): records are dealt to the GPU that owns their tag's
// RANGE (splitters picked from a sample of tags), so that every GPU joins a contiguous stretch of the tag order.
// range of a tag = number of splitters that are < the tag (FastqViewWithId::cmp order; splitters ascending): equal tags
// always share a range.
__device__ __forceinline__ bool tag_less(const uint8_t* __restrict__ a, uint32_t la, const uint8_t* __restrict__ b, uint32_t lb)
{
    const uint32_t m = la < lb ? la : lb;
    uint32_t k = 0;
    while (k < m && a[k] == b[k]) ++k;
    return k < m ? a[k] < b[k] : la < lb;
}

__global__ __launch_bounds__(kBlock)
void classify_tags_kernel(fqd_tags t, const uint8_t* __restrict__ split_bytes, uint32_t split_stride, const uint32_t* __restrict__ split_len,
                          uint32_t n_split, uint32_t* __restrict__ range_out)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < t.n; i += uint64_t(gridDim.x) * kBlock) {
        const uint8_t* __restrict__ a = t.bytes + t.offsets[i];
        const uint32_t L = t.lengths[i];
        uint32_t lo = 0, hi = n_split;                       // splitters [0, lo) are < tag, [hi, n) are not
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tag_less(split_bytes + uint64_t(mid) * split_stride, split_len[mid], a, L)) lo = mid + 1; else hi = mid;
        }
        range_out[i] = lo;
    }
}

__global__ __launch_bounds__(kBlock)
void range_keep_kernel(const uint32_t* __restrict__ range, uint64_t n, uint32_t which, uint8_t* __restrict__ keep, unsigned long long* __restrict__ count)
{
    uint32_t c = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const bool k = range[i] == which;
        keep[i] = k ? 1 : 0;
        c += k ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, static_cast<unsigned long long>(c));
}

// Sample k of n_samples = the tag of record k * n / n_samples, cut to `stride` bytes (a cut tag still splits the order).
__global__ __launch_bounds__(kBlock)
void sample_tags_kernel(fqd_tags t, uint32_t n_samples, uint32_t stride, uint8_t* __restrict__ out_bytes, uint32_t* __restrict__ out_len)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_samples) return;
    const uint64_t i = uint64_t(k) * t.n / n_samples;
    const uint8_t* __restrict__ a = t.bytes + t.offsets[i];
    const uint32_t L = t.lengths[i] < stride ? t.lengths[i] : stride;
    for (uint32_t b = 0; b < L; ++b) out_bytes[uint64_t(k) * stride + b] = a[b];
    out_len[k] = L;
}

__global__ __launch_bounds__(kBlock)
void max_u32_kernel(const uint32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out)
{
    uint32_t m = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) m = v[i] > m ? v[i] : m;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// Output offsets: pair k (tag order) is written iff keep[k]; its record idx[k] of this file then
// starts at the sum of the sizes of the kept records before it.  dest[] (per record of the file) is
// preset to ~0 by the host code.
constexpr int kOffTile = kBlock * 8;
__global__ __launch_bounds__(kBlock)
void out_sizes_kernel(const uint8_t* __restrict__ keep, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ sizes,
                      unsigned long long* __restrict__ tile_sum)
{
    __shared__ unsigned long long ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kOffTile + uint64_t(threadIdx.x) * 8u;
    unsigned long long s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const uint64_t k = base + uint32_t(e); if (k < n && keep[k]) s += sizes[idx[k]]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(1024)
void u64_scan_kernel(unsigned long long* __restrict__ data, uint32_t n, unsigned long long* __restrict__ total)
{
    __shared__ unsigned long long wt[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t t0 = 0; t0 < n; t0 += 1024u) {
        const uint32_t i = t0 + threadIdx.x;
        const unsigned long long v = i < n ? data[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
        if (lane == 63u) wt[wave] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wt[w];
        if (i < n) data[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kBlock)
void out_offsets_kernel(const uint8_t* __restrict__ keep, const uint32_t* __restrict__ idx, uint64_t n, const uint32_t* __restrict__ sizes,
                        const unsigned long long* __restrict__ tile_start, unsigned long long* __restrict__ dest)
{
    __shared__ unsigned long long ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kOffTile + uint64_t(threadIdx.x) * 8u;
    uint32_t sz[8]; uint32_t rec[8];
    unsigned long long s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t k = base + uint32_t(e);
        sz[e] = 0; rec[e] = 0;
        if (k < n && keep[k]) { rec[e] = idx[k]; sz[e] = sizes[rec[e]]; s += sz[e]; }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
    if (lane == 63u) ws[wave] = inc;
    __syncthreads();
    unsigned long long at = tile_start[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < wave; ++w) at += ws[w];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t k = base + uint32_t(e);
        if (k < n && keep[k]) { dest[rec[e]] = at; at += sz[e]; }
    }
}

// Output plan of a run whose text stays in HBM: for pair k (tag order) the record idx[k] of this file is
// copied from src_out[k] (its place in the text) to dst_out[k] (its place in the output) when keep[k];
// len_out[k] = its size, or 0 for a pair that is not written.
__global__ __launch_bounds__(kBlock)
void plan_gather_kernel(const uint8_t* __restrict__ keep, const uint32_t* __restrict__ idx, uint64_t n,
                        const uint64_t* __restrict__ starts, const uint32_t* __restrict__ sizes,
                        uint64_t* __restrict__ src_out, uint32_t* __restrict__ len_out, unsigned long long* __restrict__ tile_sum)
{
    __shared__ unsigned long long ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kOffTile + uint64_t(threadIdx.x) * 8u;
    unsigned long long s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t k = base + uint32_t(e);
        if (k < n) {
            const uint64_t r = idx ? uint64_t(idx[k]) : k;               // no list: pair k is record k
            const uint32_t L = keep[k] ? sizes[r] : 0u;
            src_out[k] = starts[r]; len_out[k] = L; s += L;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(kBlock)
void plan_offsets_kernel(const uint32_t* __restrict__ len, uint64_t n, const unsigned long long* __restrict__ tile_start,
                         unsigned long long* __restrict__ dst_out)
{
    __shared__ unsigned long long ws[4];
    const uint64_t base = uint64_t(blockIdx.x) * kOffTile + uint64_t(threadIdx.x) * 8u;
    uint32_t L[8];
    unsigned long long s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const uint64_t k = base + uint32_t(e); L[e] = k < n ? len[k] : 0u; s += L[e]; }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
    if (lane == 63u) ws[wave] = inc;
    __syncthreads();
    unsigned long long at = tile_start[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < wave; ++w) at += ws[w];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const uint64_t k = base + uint32_t(e); if (k < n) { dst_out[k] = at; at += L[e]; } }
}

// ---------------------------------------------------------------------------------------------
struct Carver {                                       // 256-byte aligned pieces of one scratch block
    char* p; size_t used = 0;
    template <class T> T* take(size_t count)
    {
        T* r = p ? reinterpret_cast<T*>(p + used) : nullptr;
        used += (count * sizeof(T) + 255) & ~size_t(255);
        return r;
    }
};

struct SortBuffers {
    uint64_t *keys[2]; uint32_t *vals[2]; uint32_t *counts, *tot;
    uint32_t *bitmap; uint8_t* rank; uint32_t *width, *lsb, *vary, *info;
    unsigned int* min_max;
    uint8_t* head; JoinCarry* carry; uint32_t* pair_tiles; unsigned long long* n_pairs;
};

// Sorts the union; on return keys[cur]/vals[cur] hold the sorted order (cur is returned).
// key_words_out: number of 64-bit key words (0: every tag is the same / at most one record).
int sort_union(fqd_engine* e, hipStream_t stream, const Union& u, const SortBuffers& sb, uint32_t max_len_hint,
               uint32_t min_len, uint32_t max_len, int* cur_out, uint32_t* key_words_out)
{
    (void)max_len_hint;
    const uint64_t N = u.n_a + u.n_b;
    const uint32_t n_tiles = uint32_t((N + kSortTile - 1) / kSortTile);
    const CodeTable ct{sb.rank, sb.width, sb.lsb, sb.vary, sb.info};
    hipLaunchKernelGGL(iota_payload_kernel, dim3(grid_for(N)), dim3(kBlock), 0, stream, sb.vals[0], u.n_a, u.n_b);
    uint32_t info[2] = {0, 0};
    if (max_len) {
        JOIN_TRY(e, hipMemsetAsync(sb.bitmap, 0, size_t(max_len) * 8 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(census_kernel, dim3(grid_for(N, kBlock, 1024)), dim3(kBlock), 0, stream, u, max_len, sb.bitmap);
        hipLaunchKernelGGL(build_codes_kernel, dim3(1), dim3(1024), 0, stream,
                           static_cast<const uint32_t*>(sb.bitmap), min_len, max_len, ct);
        JOIN_TRY(e, hipMemcpyAsync(info, sb.info, sizeof info, hipMemcpyDeviceToHost, stream));
        JOIN_TRY(e, hipStreamSynchronize(stream));       // the key width decides how many passes are launched
    }
    const uint32_t B = info[0];
    // Every 64 key bits cost an encode and up to eight radix passes over all records: tags that need tens of
    // thousands of bits (IDs of kilobytes that differ all along) would keep the device busy for minutes to
    // hours.  Refused with a clear message rather than left running; the reference has no such limit.
    if (B > 32768u)
        return fqd_internal_fail(e, FQD_ERR_ARG, "ID tags need more than 32768 key bits (tags of several kilobytes that differ "
                                                 "throughout): not supported by the device join");
    const uint32_t W = (B + 63u) / 64u;
    int cur = 0;
    for (uint32_t word = 0; word < W; ++word) {        // least significant word first
        hipLaunchKernelGGL(encode_word_kernel, dim3(grid_for(N, kBlock, 2048)), dim3(kBlock), 0, stream,
                           u, static_cast<const uint32_t*>(sb.vals[cur]), N, word, min_len, ct, sb.keys[cur]);
        const uint32_t nbits = word + 1 == W ? B - 64u * word : 64u;
        for (uint32_t shift = 0; shift < nbits; shift += 8u) {
            const uint32_t dbits = std::min(8u, nbits - shift), mask = (1u << dbits) - 1u;
            const uint32_t grid = std::min<uint32_t>(n_tiles, 256u * 8u);
            hipLaunchKernelGGL(radix_hist_kernel, dim3(grid), dim3(kBlock), 0, stream,
                               static_cast<const uint64_t*>(sb.keys[cur]), N, shift, mask, sb.counts, n_tiles);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(1024), 0, stream, sb.counts, n_tiles, sb.tot);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(grid), dim3(kBlock), 0, stream,
                               static_cast<const uint64_t*>(sb.keys[cur]), static_cast<const uint32_t*>(sb.vals[cur]), N, shift, mask,
                               static_cast<const uint32_t*>(sb.counts), n_tiles, static_cast<const uint32_t*>(sb.tot),
                               sb.keys[cur ^ 1], sb.vals[cur ^ 1]);
            cur ^= 1;
        }
    }
    JOIN_TRY(e, hipGetLastError());
    *cur_out = cur; *key_words_out = W;
    return FQD_OK;
}

int carve(fqd_engine* e, uint64_t N, uint64_t n_a, uint32_t max_len_cap, bool sizing_only, SortBuffers& sb, void** base_io, size_t* bytes)
{
    Carver c{static_cast<char*>(sizing_only ? nullptr : *base_io)};
    const uint32_t n_tiles = uint32_t((N + kSortTile - 1) / kSortTile);
    sb.min_max = c.take<unsigned int>(64);
    sb.info = c.take<uint32_t>(64);
    sb.n_pairs = c.take<unsigned long long>(32);
    sb.tot = c.take<uint32_t>(256);
    sb.keys[0] = c.take<uint64_t>(N); sb.keys[1] = c.take<uint64_t>(N);
    sb.vals[0] = c.take<uint32_t>(N); sb.vals[1] = c.take<uint32_t>(N);
    sb.counts = c.take<uint32_t>(size_t(256) * n_tiles);
    sb.bitmap = c.take<uint32_t>(size_t(max_len_cap) * 8);
    sb.rank = c.take<uint8_t>(size_t(max_len_cap) * 256);
    sb.width = c.take<uint32_t>(max_len_cap); sb.lsb = c.take<uint32_t>(max_len_cap); sb.vary = c.take<uint32_t>(max_len_cap);
    sb.head = c.take<uint8_t>(N);
    sb.carry = c.take<JoinCarry>((N + kJoinTile - 1) / kJoinTile + 1);
    sb.pair_tiles = c.take<uint32_t>((n_a + kPairTile - 1) / kPairTile + 1);
    *bytes = c.used + 256;
    (void)e;
    return FQD_OK;
}

// Length range of the union's tags (one host round trip, before anything is sized by it).
int length_range(fqd_engine* e, hipStream_t stream, const Union& u, unsigned int* d_min_max, uint32_t* min_len, uint32_t* max_len)
{
    unsigned int init[2] = {0xFFFFFFFFu, 0u}, got[2] = {0, 0};
    JOIN_TRY(e, hipMemcpyAsync(d_min_max, init, sizeof init, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(len_range_kernel, dim3(grid_for(u.n_a + u.n_b, kBlock, 1024)), dim3(kBlock), 0, stream, u, d_min_max);
    JOIN_TRY(e, hipMemcpyAsync(got, d_min_max, sizeof got, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    *min_len = got[0]; *max_len = got[1];
    return FQD_OK;
}

int run_join(fqd_engine* e, const fqd_tags* a, const fqd_tags* b, const fqd_join* out, uint32_t* perm_only)
{
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const Union u{a->bytes, a->offsets, a->lengths, a->n, b ? b->bytes : nullptr, b ? b->offsets : nullptr, b ? b->lengths : nullptr, b ? b->n : 0};
    const uint64_t N = u.n_a + u.n_b;
    // a first small scratch for the length range, then the real one sized by it
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, 4096, &base);
    if (rc) return rc;
    uint32_t min_len = 0, max_len = 0;
    if ((rc = length_range(e, stream, u, static_cast<unsigned int*>(base), &min_len, &max_len))) return rc;
    SortBuffers sb{};
    size_t bytes = 0;
    carve(e, N, u.n_a, max_len + 1, true, sb, nullptr, &bytes);
    if ((rc = fqd_internal_scratch(e, 0, bytes, &base))) return rc;
    carve(e, N, u.n_a, max_len + 1, false, sb, &base, &bytes);

    int cur = 0; uint32_t W = 0;
    if ((rc = sort_union(e, stream, u, sb, max_len, min_len, max_len, &cur, &W))) return rc;
    const uint64_t* keys = sb.keys[cur]; const uint32_t* vals = sb.vals[cur];
    if (perm_only) {                                   // fqd_sort_tags: the sorted payloads are the permutation
        JOIN_TRY(e, hipMemcpyAsync(perm_only, vals, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        JOIN_TRY(e, hipStreamSynchronize(stream));      // the scratch may be reused by the next call
        return FQD_OK;
    }
    if (W == 0) JOIN_TRY(e, hipMemsetAsync(sb.keys[cur], 0, N * sizeof(uint64_t), stream));   // all tags equal: one run
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(N, kBlock, 2048)), dim3(kBlock), 0, stream, u, keys, vals, N, W, sb.head);
    const uint32_t j_tiles = uint32_t((N + kJoinTile - 1) / kJoinTile);
    hipLaunchKernelGGL(join_summary_kernel, dim3(j_tiles), dim3(kBlock), 0, stream, vals, static_cast<const uint8_t*>(sb.head), N, sb.carry);
    hipLaunchKernelGGL(join_carry_scan_kernel, dim3(1), dim3(1024), 0, stream, sb.carry, j_tiles);
    if (u.n_a) JOIN_TRY(e, hipMemsetAsync(out->match_a, 0xFF, u.n_a * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(join_emit_kernel, dim3(j_tiles), dim3(kBlock), 0, stream, vals, static_cast<const uint8_t*>(sb.head), N,
                       static_cast<const JoinCarry*>(sb.carry), out->perm_a, out->perm_b, out->match_a, out->match_b);
    const uint32_t p_tiles = uint32_t((u.n_a + kPairTile - 1) / kPairTile);
    JOIN_TRY(e, hipMemsetAsync(sb.n_pairs, 0, sizeof(unsigned long long), stream));
    if (p_tiles) {
        hipLaunchKernelGGL(pair_count_kernel, dim3(p_tiles), dim3(kBlock), 0, stream, static_cast<const uint32_t*>(out->match_a), u.n_a, sb.pair_tiles);
        hipLaunchKernelGGL(u32_scan_kernel, dim3(1), dim3(1024), 0, stream, sb.pair_tiles, p_tiles, sb.n_pairs);
        hipLaunchKernelGGL(pair_emit_kernel, dim3(p_tiles), dim3(kBlock), 0, stream, static_cast<const uint32_t*>(out->match_a),
                           static_cast<const uint32_t*>(out->perm_a), static_cast<const uint32_t*>(out->perm_b), u.n_a,
                           static_cast<const uint32_t*>(sb.pair_tiles), out->pair_a, out->pair_b);
    }
    JOIN_TRY(e, hipGetLastError());
    unsigned long long n_pairs = 0;
    JOIN_TRY(e, hipMemcpyAsync(&n_pairs, sb.n_pairs, sizeof n_pairs, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    if (out->n_pairs) *out->n_pairs = n_pairs;
    return FQD_OK;
}

} // namespace

extern "C" {

int fqd_extract_tags(fqd_engine* e, const uint8_t* text, const uint64_t* id_start, const uint32_t* id_len, uint64_t n,
                     uint64_t* tag_off, uint32_t* tag_len)
{
    if (!e) return FQD_ERR_ARG;
    if (n && (!text || !id_start || !id_len || !tag_off || !tag_len))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_extract_tags: bad arguments");
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipLaunchKernelGGL(extract_tags_kernel, dim3(grid_for(n, kBlock, 4096)), dim3(kBlock), 0, fqd_internal_stream(e),
                       text, id_start, id_len, n, tag_off, tag_len);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_sort_tags(fqd_engine* e, const fqd_tags* t, uint32_t* perm)
{
    if (!e) return FQD_ERR_ARG;
    if (!t || (t->n && (!perm || !t->offsets || !t->lengths)) || t->n >= 0x80000000ull)
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_sort_tags: bad arguments (at most 2^31-1 records)");
    if (t->n == 0) return FQD_OK;
    return run_join(e, t, nullptr, nullptr, perm);
}

int fqd_join_tags(fqd_engine* e, const fqd_tags* a, const fqd_tags* b, const fqd_join* out)
{
    if (!e) return FQD_ERR_ARG;
    if (!a || !b || !out || a->n >= 0x80000000ull || b->n >= 0x80000000ull ||
        (a->n && (!a->offsets || !a->lengths || !out->perm_a || !out->match_a)) ||
        (b->n && (!b->offsets || !b->lengths || !out->perm_b || !out->match_b)) ||
        (a->n && b->n && (!out->pair_a || !out->pair_b)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_join_tags: bad arguments (at most 2^31-1 records per file)");
    if (out->n_pairs) *out->n_pairs = 0;
    if (a->n + b->n == 0) return FQD_OK;
    return run_join(e, a, b, out, nullptr);
}

int fqd_copy_spans(fqd_engine* e, const uint8_t* src, const uint64_t* src_off, const uint32_t* len, uint64_t n,
                   uint8_t* dst, const uint64_t* dst_off)
{
    if (!e) return FQD_ERR_ARG;
    if (n && (!src || !src_off || !len || !dst || !dst_off)) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_copy_spans: bad arguments");
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipLaunchKernelGGL(copy_spans_kernel, dim3(grid_for(n * kSpanLanes, kBlock, 8192)), dim3(kBlock), 0, fqd_internal_stream(e), src, src_off, len, n, dst, dst_off);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_count_tags_le(fqd_engine* e, const fqd_tags* t, const fqd_tags* other, uint64_t other_index, uint64_t* count)
{
    if (!e) return FQD_ERR_ARG;
    if (!t || !other || !count || other_index >= other->n || (t->n && (!t->offsets || !t->lengths)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_count_tags_le: bad arguments");
    *count = 0;
    if (t->n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, 4096, &base);
    if (rc) return rc;
    unsigned long long* d_count = static_cast<unsigned long long*>(base);
    uint64_t off = 0; uint32_t len = 0;
    JOIN_TRY(e, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    JOIN_TRY(e, hipMemcpyAsync(&off, other->offsets + other_index, sizeof off, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipMemcpyAsync(&len, other->lengths + other_index, sizeof len, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    hipLaunchKernelGGL(count_le_kernel, dim3(grid_for(t->n, kBlock, 2048)), dim3(kBlock), 0, stream, *t, other->bytes + off, len, d_count);
    unsigned long long got = 0;
    JOIN_TRY(e, hipMemcpyAsync(&got, d_count, sizeof got, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    *count = got;
    return FQD_OK;
}

int fqd_output_offsets(fqd_engine* e, const uint8_t* keep, const uint32_t* idx, uint64_t n, const uint32_t* sizes,
                       uint64_t* dest, uint64_t* total)
{
    if (!e) return FQD_ERR_ARG;
    if (!total || (n && (!keep || !idx || !sizes || !dest))) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_output_offsets: bad arguments");
    *total = 0;
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint32_t tiles = uint32_t((n + kOffTile - 1) / kOffTile);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, (size_t(tiles) + 2) * sizeof(unsigned long long), &base);
    if (rc) return rc;
    unsigned long long* tile = static_cast<unsigned long long*>(base);
    unsigned long long* d_total = tile + tiles;
    hipLaunchKernelGGL(out_sizes_kernel, dim3(tiles), dim3(kBlock), 0, stream, keep, idx, n, sizes, tile);
    hipLaunchKernelGGL(u64_scan_kernel, dim3(1), dim3(1024), 0, stream, tile, tiles, d_total);
    hipLaunchKernelGGL(out_offsets_kernel, dim3(tiles), dim3(kBlock), 0, stream, keep, idx, n, sizes,
                       static_cast<const unsigned long long*>(tile), reinterpret_cast<unsigned long long*>(dest));
    JOIN_TRY(e, hipGetLastError());
    unsigned long long got = 0;
    JOIN_TRY(e, hipMemcpyAsync(&got, d_total, sizeof got, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    *total = got;
    return FQD_OK;
}

int fqd_output_plan(fqd_engine* e, const uint8_t* keep, const uint32_t* idx, uint64_t n, const uint64_t* starts, const uint32_t* sizes,
                    uint64_t* src_off, uint32_t* len, uint64_t* dst_off, uint64_t* total)
{
    if (!e) return FQD_ERR_ARG;
    if (!total || (n && (!keep || !starts || !sizes || !src_off || !len || !dst_off)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_output_plan: bad arguments");
    *total = 0;
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint32_t tiles = uint32_t((n + kOffTile - 1) / kOffTile);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, (size_t(tiles) + 2) * sizeof(unsigned long long), &base);
    if (rc) return rc;
    unsigned long long* tile = static_cast<unsigned long long*>(base);
    unsigned long long* d_total = tile + tiles;
    hipLaunchKernelGGL(plan_gather_kernel, dim3(tiles), dim3(kBlock), 0, stream, keep, idx, n, starts, sizes, src_off, len, tile);
    hipLaunchKernelGGL(u64_scan_kernel, dim3(1), dim3(1024), 0, stream, tile, tiles, d_total);
    hipLaunchKernelGGL(plan_offsets_kernel, dim3(tiles), dim3(kBlock), 0, stream, static_cast<const uint32_t*>(len), n,
                       static_cast<const unsigned long long*>(tile), reinterpret_cast<unsigned long long*>(dst_off));
    JOIN_TRY(e, hipGetLastError());
    unsigned long long got = 0;
    JOIN_TRY(e, hipMemcpyAsync(&got, d_total, sizeof got, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    *total = got;
    return FQD_OK;
}

int fqd_gather_seqs(fqd_engine* e, const uint32_t* idx, uint64_t n, const uint64_t* off_table, const uint32_t* len_table,
                    uint64_t* off_out, uint32_t* len_out)
{
    if (!e) return FQD_ERR_ARG;
    if (n && (!idx || !off_table || !len_table || !off_out || !len_out))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_gather_seqs: bad arguments");
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipLaunchKernelGGL(gather_seq_kernel, dim3(grid_for(n, kBlock, 4096)), dim3(kBlock), 0, fqd_internal_stream(e),
                       idx, n, off_table, len_table, off_out, len_out);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_classify_tags(fqd_engine* e, const fqd_tags* t, const uint8_t* split_bytes, uint32_t split_stride, const uint32_t* split_len,
                      uint32_t n_split, uint32_t* range_out)
{
    if (!e) return FQD_ERR_ARG;
    if (!t || (t->n && (!t->offsets || !t->lengths || !range_out)) || (n_split && (!split_bytes || !split_len)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_classify_tags: bad arguments");
    if (t->n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipLaunchKernelGGL(classify_tags_kernel, dim3(grid_for(t->n, kBlock, 4096)), dim3(kBlock), 0, fqd_internal_stream(e),
                       *t, split_bytes, split_stride, split_len, n_split, range_out);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_range_keep(fqd_engine* e, const uint32_t* range, uint64_t n, uint32_t which, uint8_t* keep, uint64_t* count)
{
    if (!e) return FQD_ERR_ARG;
    if (!count || (n && (!range || !keep))) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_range_keep: bad arguments");
    *count = 0;
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, 4096, &base);
    if (rc) return rc;
    unsigned long long* d_count = static_cast<unsigned long long*>(base);
    JOIN_TRY(e, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(range_keep_kernel, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), 0, stream, range, n, which, keep, d_count);
    unsigned long long got = 0;
    JOIN_TRY(e, hipMemcpyAsync(&got, d_count, sizeof got, hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    *count = got;
    return FQD_OK;
}

int fqd_sample_tags(fqd_engine* e, const fqd_tags* t, uint32_t n_samples, uint32_t stride, uint8_t* out_bytes, uint32_t* out_len)
{
    if (!e) return FQD_ERR_ARG;
    if (!t || !t->n || !n_samples || !stride || !out_bytes || !out_len || !t->offsets || !t->lengths)
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_sample_tags: bad arguments");
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipLaunchKernelGGL(sample_tags_kernel, dim3((n_samples + kBlock - 1) / kBlock), dim3(kBlock), 0, fqd_internal_stream(e),
                       *t, n_samples, stride, out_bytes, out_len);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_max_u32(fqd_engine* e, const uint32_t* values, uint64_t n, uint32_t* max_out)
{
    if (!e) return FQD_ERR_ARG;
    if (!max_out || (n && !values)) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_max_u32: bad arguments");
    *max_out = 0;
    if (n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 1, 4096, &base);
    if (rc) return rc;
    uint32_t* d = static_cast<uint32_t*>(base);
    JOIN_TRY(e, hipMemsetAsync(d, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(max_u32_kernel, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), 0, stream, values, n, d);
    JOIN_TRY(e, hipMemcpyAsync(max_out, d, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    return FQD_OK;
}

} // extern "C"
