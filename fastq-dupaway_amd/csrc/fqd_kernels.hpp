// fqd_kernels.hpp — gfx950 kernels of the dedup engine (included by fqd_engine.hip).
//
//   encode_*   : ASCII bases -> validated, packed key words + 64-bit placement hash
//                (replaces SeqUtils::seq2hash / pattern2number / _char2number,
//                 seq_utils.cpp:3-49, and the key constructors, hash_dup_remover.cpp:4-24)
//   insert     : probe / insert into the HBM-resident open-addressing set with exact
//                key verification and first-occurrence-wins
//                (replaces records.find + records.insert, hash_dup_remover.hpp:133-138,
//                 237-244, and operator==, hash_dup_remover.cpp:10-14,26-33)
//   rehash, scans, re-layout, partition: housekeeping around those two.
#pragma once
#include "fqd_device.hpp"

namespace fqd {

constexpr int kBlock = 256;

// Diagnostic build only (make STAMPS=1; never the shipped library): thread 0 of every workgroup adds the time it
// spends in each numbered phase (100 MHz ticks of wall_clock64, barrier waits included) to g_stamps[kernel][phase];
// the engine prints the table when it is destroyed.  Kernel ids: 0/1 scatter level 1/2, 2 dedup, 3 staged encoder.
#ifdef FQD_STAMPS
__device__ unsigned long long g_stamps[8][16];
#define STAMP_DECL __shared__ unsigned long long st_acc_[16]; __shared__ unsigned long long st_last_; \
    if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) st_acc_[q_] = 0; st_last_ = wall_clock64(); }
#define STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); st_acc_[i] += now_ - st_last_; st_last_ = now_; } } while (0)
#define STAMP_FLUSH(kid) do { if (threadIdx.x == 0) for (int q_ = 0; q_ < 16; ++q_) if (st_acc_[q_]) atomicAdd(&g_stamps[kid][q_], st_acc_[q_]); } while (0)
#define STAMP_LOADS_IN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")      /* so that a stamp behind it separates waiting for loads from working on them */
#else
#define STAMP_LOADS_IN() do { } while (0)
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#define STAMP_FLUSH(kid) do { } while (0)
#endif

// One mate's input as the kernels see it.
struct SegView {
    const uint8_t*  bases;
    const uint64_t* offsets;   // nullptr: uniform
    const uint32_t* lengths;   // nullptr: uniform
    uint32_t        ulen;
    uint32_t        ustride;
    uint32_t        clamp = 0xFFFFFFFFu;   // fqd_encode_padded: no read is taken longer than the key slot has room for (a longer one is an error reported apart)
    __device__ __forceinline__ const uint8_t* ptr(uint64_t i) const
    { return bases + (offsets ? offsets[i] : i * uint64_t(ustride)); }
    __device__ __forceinline__ uint32_t len(uint64_t i) const { const uint32_t l = lengths ? lengths[i] : ulen; return l < clamp ? l : clamp; }
};

// Where keys live.  Uniform engines: key j = keys + j*stride + lead, all with
// W0 words (lens implied).  Ragged engines: key j = keys + koff[j] = [len0 | len1<<32][words...].
struct KeyStore {
    uint64_t*       keys;
    const uint64_t* koff;      // nullptr: uniform
    uint32_t        W0;
    uint32_t        stride;    // words between consecutive uniform keys
    uint32_t        lead;      // words to skip at the start of a uniform slot (records: 1 for the hash)
    __device__ __forceinline__ uint64_t* slot(uint64_t j) const
    { return keys + (koff ? koff[j] : j * uint64_t(stride) + lead); }
};

// Where a batch's verdicts go.  keep[] is preset to 1 and only ever cleared; `first`, when given
// (hash engines of the optimistic sharded exchange), names an earlier record with the same key
// for every cleared flag.
struct Verdicts {
    uint8_t*  keep;
    uint32_t* first;       // may be nullptr
    uint32_t  first_idx;   // engine index of keep[0]
    __device__ __forceinline__ void lose(uint32_t loser, uint32_t winner) const
    {
        keep[loser - first_idx] = 0;
        if (first) first[loser - first_idx] = winner;
    }
};

// Geometry of the bulk (partitioned) insert, see bulk_* kernels below.  The encoders can fold
// the level-1 histogram of that path into their own pass (Hist1: hist == nullptr -> off).
struct BulkGeom {
    uint64_t slot_mask;
    uint32_t seg_bits;     // log2(slots per segment)
    uint32_t bits1, bits2; // partition digits: bucket = pos >> seg_bits = (d1 << bits2) | d2
    uint32_t tag_mask;     // the table's slot tag = (hash >> 32) & tag_mask (see slot_tag)
};
// A table slot is (tag << 32) | record index.  The tag only filters which owners are worth a key
// comparison, so it is kept narrow enough that a partition record — the bits of the table
// position below the level-1 digit, the tag, the record index — fits 8 bytes:
// seg_bits + bits2 + tag bits <= 32.  Every path that touches a table uses the table's mask.
__host__ __device__ __forceinline__ uint64_t slot_tag(uint64_t hash, uint32_t tag_mask) { return (hash >> 32) & tag_mask; }
__device__ __forceinline__ uint32_t bucket_of(uint64_t hash, const BulkGeom& g) { return uint32_t((hash & g.slot_mask) >> g.seg_bits); }
struct Hist1 {
    uint32_t* hist;        // 256 global counters, or nullptr
    BulkGeom  g;
    uint64_t  hash_and;    // ANDed onto every hash: all ones, except under FQD_FLAG_TEST_WEAK_HASH
};
// Per-block LDS histogram helpers shared by the three encoders.
__device__ __forceinline__ void hist1_clear(uint32_t* lh) { for (uint32_t k = threadIdx.x; k < 256u; k += blockDim.x) lh[k] = 0; __syncthreads(); }
__device__ __forceinline__ void hist1_flush(const uint32_t* lh, uint32_t* gh)
{
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 256u; k += blockDim.x) if (lh[k]) atomicAdd(&gh[k], lh[k]);
}

// ---------------------------------------------------------------------------
// Shared tail of both encoders: error word for a record whose packing saw a bad byte.
// Rare path: rescans the mate(s) byte by byte in global memory for the FIRST offender,
// which is what the reference would have reported (seq_utils.cpp:17-19).
__device__ __noinline__ uint64_t locate_bad_base(const uint8_t* p0, uint32_t l0, const uint8_t* p1, uint32_t l1,
                                                 uint64_t record)
{
    uint32_t byte = 0;
    uint32_t pos = first_bad_base(p0, l0, &byte);
    if (pos != 0xFFFFFFFFu) return make_error(record, 0, pos, byte);
    if (p1) {
        pos = first_bad_base(p1, l1, &byte);
        if (pos != 0xFFFFFFFFu) return make_error(record, 1, pos, byte);
    }
    return kNoError;
}

// ---------------------------------------------------------------------------
// encode_general: one record (all its mates) per lane, loads straight from HBM.
// Handles ragged and uniform input, any length, any alignment.
//   hash_out != nullptr : hash of record i -> hash_out[i]
//   ks.lead == 1        : hash -> word 0 of the record's slot (records layout); both may apply
template <int S>
__global__ __launch_bounds__(kBlock)
void encode_general_kernel(SegView s0, SegView s1, uint64_t n, uint64_t first_idx,
                           KeyStore ks, uint64_t* __restrict__ hash_out, uint64_t* __restrict__ err, Hist1 h1)
{
    __shared__ uint32_t lhist[256];
    if (h1.hist) hist1_clear(lhist);
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const uint32_t l0 = s0.len(i);
        const uint32_t l1 = (S == 2) ? s1.len(i) : 0u;
        uint64_t* out = ks.slot(first_idx + i);
        uint64_t h = hash_begin(l0, 0);
        if (ks.koff) *out++ = uint64_t(l0) | (uint64_t(l1) << 32);
        auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
        const uint8_t* p0 = s0.ptr(i);
        const uint8_t* p1 = (S == 2) ? s1.ptr(i) : nullptr;
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(p0);
        uint32_t diff = pack_mate(reinterpret_cast<const uint32_t*>(a0 & ~uintptr_t(3)), uint32_t(a0 & 3u), l0, sink);
        if (S == 2) {
            const uint64_t m0 = h;
            h = hash_begin(l1, 0);
            const uintptr_t a1 = reinterpret_cast<uintptr_t>(p1);
            diff |= pack_mate(reinterpret_cast<const uint32_t*>(a1 & ~uintptr_t(3)), uint32_t(a1 & 3u), l1, sink);
            h = hash_pair(m0, h);
        } else {
            h = hash_end(h);
        }
        h &= h1.hash_and;
        if (ks.lead) ks.slot(first_idx + i)[-1] = h;
        if (hash_out) hash_out[i] = h;
        if (h1.hist) atomicAdd(&lhist[bucket_of(h, h1.g) >> h1.g.bits2], 1u);
        if (diff) {
            const uint64_t e = locate_bad_base(p0, l0, p1, l1, first_idx + i);
            if (e != kNoError) atomicMin(reinterpret_cast<unsigned long long*>(err), static_cast<unsigned long long>(e));
        }
    }
    if (h1.hist) hist1_flush(lhist, h1.hist);
}

// Copies n16 16-byte chunks from global memory to LDS with the whole workgroup, kCopyBatch loads
// per lane in flight at a time (a plain per-chunk loop compiles to load, wait, store: one
// outstanding load per lane).  Loads are unconditional (index clamped into the range) so the
// staging values stay in registers; the LDS stores are predicated.  The input is read once:
// non-temporal loads keep it from displacing anything in L2.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
#ifndef FQD_COPY_BATCH
#define FQD_COPY_BATCH 5
#endif
constexpr int kCopyBatch = FQD_COPY_BATCH;            // 1 -> 5: encoder 4.3 -> 4.0 ms per 100 M; 10: no better
template <int BATCH>
__device__ __forceinline__ void stage_chunks_by(const u32x4* __restrict__ src, u32x4* dst, uint32_t n16, uint32_t R)
{
    for (uint32_t c0 = threadIdx.x; c0 < n16; c0 += uint32_t(BATCH) * R) {
        u32x4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const uint32_t c = c0 + uint32_t(k) * R; v[k] = __builtin_nontemporal_load(&src[c < n16 ? c : n16 - 1u]); }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const uint32_t c = c0 + uint32_t(k) * R; if (c < n16) dst[c] = v[k]; }
    }
}
__device__ __forceinline__ void stage_chunks(const uint8_t* __restrict__ src_bytes, uint32_t* __restrict__ lds_dst, uint32_t n16, uint32_t R)
{
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(src_bytes);
    u32x4* dst = reinterpret_cast<u32x4*>(lds_dst);
    // long reads: few lanes per CU fit their tile in LDS, so each keeps twice as many loads in flight
    if (n16 >= 12u * R) stage_chunks_by<2 * kCopyBatch>(src, dst, n16, R);
    else                stage_chunks_by<kCopyBatch>(src, dst, n16, R);
}

// ---------------------------------------------------------------------------
// encode_span: ragged batches whose reads lie close together in memory — trimmed reads packed back
// to back, or the sequence lines of a FASTQ block with IDs and qualities in between.  A workgroup
// takes R consecutive records; per mate, the bytes from its first record's sequence to the end of
// its last one's (the span) are pulled into LDS with coalesced 16-byte loads when they fit the
// mate's LDS budget and every record of the tile lies inside; then each lane packs its read out of
// LDS.  A tile that does not qualify (offsets that jump around, as after the `--unordered` join, or
// very long records) is encoded straight from HBM like encode_general does.  Keys go to their slots
// with per-lane stores.   LDS: S * span_cap bytes.
template <int S>
__global__ __launch_bounds__(kBlock)
void encode_span_kernel(SegView s0, SegView s1, uint64_t n, uint64_t first_idx, KeyStore ks,
                        uint64_t* __restrict__ hash_out, uint64_t* __restrict__ err, uint32_t span_cap, Hist1 h1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t lhist[256];
    __shared__ uint32_t vote;
    if (h1.hist) hist1_clear(lhist);
    const uint32_t R = blockDim.x;
    const uint64_t n_tiles = (n + R - 1) / R;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i0 = tile * R;
        const uint32_t nr = uint32_t(n - i0 < R ? n - i0 : R);
        const uint64_t i = i0 + threadIdx.x;
        const bool live = threadIdx.x < nr;
        const uint8_t* p[2] = {nullptr, nullptr};
        uint32_t len[2] = {0, 0}, head[2] = {0, 0};
        const uint8_t* lo[2] = {nullptr, nullptr};
        if (threadIdx.x == 0) vote = 1u;
        __syncthreads();
        bool fits = true;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const SegView& sv = s ? s1 : s0;
            if (live) { p[s] = sv.ptr(i); len[s] = sv.len(i); }
            lo[s] = sv.ptr(i0);
            const uint8_t* hi = sv.ptr(i0 + nr - 1) + sv.len(i0 + nr - 1);
            head[s] = uint32_t(reinterpret_cast<uintptr_t>(lo[s]) & 15u);
            fits = fits && hi >= lo[s] && uint64_t(hi - lo[s]) + head[s] + 32u <= span_cap;
            if (live) fits = fits && p[s] >= lo[s] && p[s] + len[s] <= hi;
        }
        if (!fits) vote = 0u;                                  // any lane can veto the tile
        __syncthreads();
        const bool staged = vote != 0u;
        if (staged) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const SegView& sv = s ? s1 : s0;
                const uint8_t* hi = sv.ptr(i0 + nr - 1) + sv.len(i0 + nr - 1);
                stage_chunks(lo[s] - head[s], lds + s * (span_cap >> 2), (head[s] + uint32_t(hi - lo[s]) + 15u) >> 4, R);
            }
        }
        __syncthreads();
        if (live) {
            uint64_t* out = ks.slot(first_idx + i);
            uint64_t h = hash_begin(len[0], 0);
            if (ks.koff) *out++ = uint64_t(len[0]) | (uint64_t(len[1]) << 32);
            auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
            uint32_t diff;
            if (staged) {
                const uint32_t b0 = head[0] + uint32_t(p[0] - lo[0]);
                diff = pack_mate(lds + (b0 >> 2), b0 & 3u, len[0], sink);
            } else {
                const uintptr_t a0 = reinterpret_cast<uintptr_t>(p[0]);
                diff = pack_mate(reinterpret_cast<const uint32_t*>(a0 & ~uintptr_t(3)), uint32_t(a0 & 3u), len[0], sink);
            }
            if (S == 2) {
                const uint64_t m0 = h;
                h = hash_begin(len[1], 0);
                if (staged) {
                    const uint32_t b1 = span_cap + head[1] + uint32_t(p[1] - lo[1]);
                    diff |= pack_mate(lds + (b1 >> 2), b1 & 3u, len[1], sink);
                } else {
                    const uintptr_t a1 = reinterpret_cast<uintptr_t>(p[1]);
                    diff |= pack_mate(reinterpret_cast<const uint32_t*>(a1 & ~uintptr_t(3)), uint32_t(a1 & 3u), len[1], sink);
                }
                h = hash_pair(m0, h);
            } else {
                h = hash_end(h);
            }
            h &= h1.hash_and;
            if (ks.lead) ks.slot(first_idx + i)[-1] = h;
            if (hash_out) hash_out[i] = h;
            if (h1.hist) atomicAdd(&lhist[bucket_of(h, h1.g) >> h1.g.bits2], 1u);
            if (diff) {
                const uint64_t e = locate_bad_base(p[0], len[0], p[1], len[1], first_idx + i);
                if (e != kNoError) atomicMin(reinterpret_cast<unsigned long long*>(err), static_cast<unsigned long long>(e));
            }
        }
        __syncthreads();
    }
    if (h1.hist) hist1_flush(lhist, h1.hist);
}

// ---------------------------------------------------------------------------
// encode_staged: uniform-length, uniform-stride input (the BASELINE layout: 150 B
// per read, back to back).  A workgroup pulls a tile of R reads (R*stride bytes,
// 16-B aligned chunks) into LDS with fully coalesced 16-byte loads, then each
// lane packs one read out of LDS.  HBM sees only whole-line streaming reads.
//
// LDS_OUT: instead of 64 scattered 8-byte stores per instruction, a lane writes its
// key words over the bytes of ITS OWN read that it has already consumed (row start =
// first 8-aligned byte >= 4 bytes into the read, so the boundary dword shared with
// the previous read is never touched; key word k ends at most 27+12k bytes in, always
// behind the read pointer, which is past 32k+32 when that word is produced).  Each
// wave then streams its 64 rows out as one contiguous run.  The host picks LDS_OUT
// only when  stride0 >= 8*row_words + 15  (row_words = W0 + lead) and the batch's key
// slots are contiguous (no koff).  rw_magic = ceil(2^32 / row_words).
//   tile_reads R = blockDim.x (multiple of 64); LDS = round16(R*stride + 32) per mate
template <bool LDS_OUT>
__global__ __launch_bounds__(kBlock)
void encode_staged_kernel(SegView s0, uint64_t n, uint64_t first_idx,
                          KeyStore ks, uint64_t* __restrict__ hash_out, uint64_t* __restrict__ err, uint32_t rw_magic, Hist1 h1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ __attribute__((aligned(16))) uint32_t lhist[256];
    if (h1.hist) hist1_clear(lhist);
    uint64_t* lds64 = reinterpret_cast<uint64_t*>(lds);
    const uint32_t R = blockDim.x;
    const uint32_t row_words = ks.W0 + ks.lead;
    const uint64_t n_tiles = (n + R - 1) / R;
    STAMP_DECL
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        STAMP(0);
        const uint64_t r0 = tile * R;
        const uint32_t nr = uint32_t(n - r0 < R ? n - r0 : R);
        const uint8_t* g0 = s0.bases + r0 * uint64_t(s0.ustride);
        const uint32_t bytes = (nr - 1u) * s0.ustride + s0.ulen;
        const uintptr_t ga = reinterpret_cast<uintptr_t>(g0);
        const uint32_t in0 = uint32_t(ga & 15u);                    // bytes before g0 in its 16-B chunk
        stage_chunks(g0 - in0, lds, (in0 + bytes + 15u) >> 4, R);
        STAMP(1);                                                   // staging, my wave
        __syncthreads();
        STAMP(2);                                                   // staging, the others
        const uint32_t t = threadIdx.x;
        const uint64_t i = r0 + t;
        const uint32_t l0 = s0.ulen;
#ifdef FQD_ENC_SKIP_PACK                                            /* diagnostic builds only: what the kernel costs without its packing (results are garbage) */
        if (t < nr && hash_out) hash_out[i] = lds64[t];
        if (false) {
#else
        if (t < nr) {
#endif
            uint64_t h = hash_begin(l0, 0);
            const uint32_t b0 = in0 + t * s0.ustride;
            uint32_t diff;
            if (LDS_OUT) {
                uint64_t* row = lds64 + ((b0 + 4u + 7u) >> 3) + ks.lead;
                uint64_t* out = row;
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                diff = pack_mate(lds + (b0 >> 2), b0 & 3u, l0, sink);
                h = hash_end(h) & h1.hash_and;
                if (ks.lead) row[-1] = h;
                if (hash_out) hash_out[i] = h;
            } else {
                uint64_t* out = ks.slot(first_idx + i);
                if (ks.koff) *out++ = uint64_t(l0);
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                diff = pack_mate(lds + (b0 >> 2), b0 & 3u, l0, sink);
                h = hash_end(h) & h1.hash_and;
                if (ks.lead) ks.slot(first_idx + i)[-1] = h;
                if (hash_out) hash_out[i] = h;
            }
            if (h1.hist) atomicAdd(&lhist[bucket_of(h, h1.g) >> h1.g.bits2], 1u);
            if (diff) {
                const uint64_t e = locate_bad_base(s0.bases + i * uint64_t(s0.ustride), l0, nullptr, 0, first_idx + i);
                if (e != kNoError) atomicMin(reinterpret_cast<unsigned long long*>(err), static_cast<unsigned long long>(e));
            }
        }
        STAMP(3);                                                   // pack + hash
#ifdef FQD_ENC_SKIP_OUT                                             /* diagnostic builds only: no key leaves the CU */
        if (false) {
#else
        if (LDS_OUT) {
#endif
            // the wave's 64 rows -> one contiguous run of key slots
            const uint32_t wave = t >> 6, lane = t & 63u;
            const uint32_t wave_reads = (nr > wave * 64u) ? ((nr - wave * 64u < 64u) ? nr - wave * 64u : 64u) : 0u;
            uint64_t* __restrict__ gout = ks.keys + (first_idx + r0 + wave * 64u) * uint64_t(ks.stride);
            const uint32_t total = wave_reads * row_words;
#ifndef FQD_OLD_STREAMOUT
            if (rw_magic == 0u) {
                // rows of 2^k words (150 bp: 8): 16 bytes per lane and store — two adjacent words of a row out of LDS, one
                // global_store_dwordx4 — and a row's number and a word's place in it by shift and mask.  The general form
                // below costs a lane two quarter-rate multiplies per 8 bytes stored: 130 of the encoder's ~830 issue slots
                // per read went into moving its 64-byte key out.
                const uint32_t half_shift = uint32_t(__ffs(int(row_words))) - 2u;          // log2(row_words / 2)
                u64x2* __restrict__ gout2 = reinterpret_cast<u64x2*>(gout);
                for (uint32_t y = lane; y < (total >> 1); y += 64u) {
                    const uint32_t rr = y >> half_shift, k2 = (y - (rr << half_shift)) << 1;
                    const uint64_t* row = lds64 + ((in0 + __umul24(wave * 64u + rr, s0.ustride) + 4u + 7u) >> 3) + k2;
                    const u64x2 v = {row[0], row[1]};
                    __builtin_nontemporal_store(v, &gout2[y]);
                }
            } else
#endif
            for (uint32_t x = lane; x < total; x += 64u) {
                const uint32_t rr = __umulhi(x, rw_magic), kk = x - rr * row_words;
                __builtin_nontemporal_store(lds64[((in0 + (wave * 64u + rr) * s0.ustride + 4u + 7u) >> 3) + kk], &gout[x]);
            }
        }
        STAMP(4);                                                   // key stream-out issue
        __syncthreads();
        STAMP(5);                                                   // barrier
    }
    STAMP_FLUSH(3);
    if (h1.hist) hist1_flush(lhist, h1.hist);
}

// Paired variant: ONE LANE PER MATE (lane 2p = mate 1 of pair p, lane 2p+1 = mate 2), so a
// 256-lane workgroup stages 128 pairs (2 x 19 KiB) and runs at the same wave occupancy as the
// single-end encoder.  Each lane hashes and parks its own mate (same in-place rule, per mate:
// stride_m >= 8*row_m + 15 with row_0 = lead + W_0, row_1 = W_1); the pair's hash is combined
// across the two lanes with one DPP exchange (hash_pair).  A wave streams out its 32 pair keys.
template <bool LDS_OUT>
__global__ __launch_bounds__(kBlock)
void encode_staged_pe_kernel(SegView s0, SegView s1, uint64_t n, uint64_t first_idx,
                             KeyStore ks, uint64_t* __restrict__ hash_out, uint64_t* __restrict__ err,
                             uint32_t tile_bytes0, uint32_t rw_magic, Hist1 h1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ __attribute__((aligned(16))) uint32_t lhist[256];
    if (h1.hist) hist1_clear(lhist);
    uint64_t* lds64 = reinterpret_cast<uint64_t*>(lds);
    const uint32_t R = blockDim.x, P = R >> 1;
    const uint32_t W_0 = seg_words(s0.ulen);
    const uint32_t row_words = ks.W0 + ks.lead;
    const uint64_t n_tiles = (n + P - 1) / P;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t r0 = tile * P;
        const uint32_t np = uint32_t(n - r0 < P ? n - r0 : P);
        uint32_t in_base[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const SegView& sv = s ? s1 : s0;
            const uint8_t* g0 = sv.bases + r0 * uint64_t(sv.ustride);
            const uint32_t bytes = (np - 1u) * sv.ustride + sv.ulen;
            const uintptr_t ga = reinterpret_cast<uintptr_t>(g0);
            const uint32_t head = uint32_t(ga & 15u);
            const uint32_t off = s ? tile_bytes0 : 0u;
            in_base[s] = off + head;
            stage_chunks(g0 - head, lds + (off >> 2), (head + bytes + 15u) >> 4, R);
        }
        __syncthreads();
        const uint32_t t = threadIdx.x, pair = t >> 1, mate = t & 1u;
        const bool live = pair < np;
        const uint64_t i = r0 + pair;
        const uint32_t len = mate ? s1.ulen : s0.ulen;
        const uint32_t stride = mate ? s1.ustride : s0.ustride;
        const uint32_t b = in_base[mate] + pair * stride;
        uint64_t h = hash_begin(len, 0);
        uint32_t diff = 0;
        uint64_t* row = nullptr;
        if (live) {
            if (LDS_OUT) {
                row = lds64 + ((b + 4u + 7u) >> 3) + (mate ? 0u : ks.lead);
                uint64_t* out = row;
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                diff = pack_mate(lds + (b >> 2), b & 3u, len, sink);
            } else {
                uint64_t* out = ks.slot(first_idx + i);
                if (ks.koff) { if (!mate) *out = uint64_t(s0.ulen) | (uint64_t(s1.ulen) << 32); ++out; }
                out += mate ? W_0 : 0u;
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                diff = pack_mate(lds + (b >> 2), b & 3u, len, sink);
            }
        }
        const uint64_t other = __shfl_xor(h, 1, 64);               // the partner lane's chain
        if (live && !mate) {
            const uint64_t hh = hash_pair(h, other) & h1.hash_and;
            if (ks.lead) { if (LDS_OUT) row[-1] = hh; else ks.slot(first_idx + i)[-1] = hh; }
            if (hash_out) hash_out[i] = hh;
            if (h1.hist) atomicAdd(&lhist[bucket_of(hh, h1.g) >> h1.g.bits2], 1u);
        }
        if (live && diff) {
            uint32_t byte = 0;
            const uint8_t* gp = (mate ? s1.bases : s0.bases) + i * uint64_t(stride);
            const uint32_t pos = first_bad_base(gp, len, &byte);
            if (pos != 0xFFFFFFFFu)
                atomicMin(reinterpret_cast<unsigned long long*>(err),
                          static_cast<unsigned long long>(make_error(first_idx + i, mate, pos, byte)));
        }
        if (LDS_OUT) {
            const uint32_t wave = t >> 6, lane = t & 63u;
            const uint32_t wave_pairs = (np > wave * 32u) ? ((np - wave * 32u < 32u) ? np - wave * 32u : 32u) : 0u;
            uint64_t* __restrict__ gout = ks.keys + (first_idx + r0 + wave * 32u) * uint64_t(ks.stride);
            const uint32_t total = wave_pairs * row_words, split = ks.lead + W_0;
#ifndef FQD_OLD_STREAMOUT
            if (rw_magic == 0u) {                                  // rows of 2^k words, mate 1's part an even number of them: see encode_staged_kernel
                const uint32_t half_shift = uint32_t(__ffs(int(row_words))) - 2u;
                u64x2* __restrict__ gout2 = reinterpret_cast<u64x2*>(gout);
                for (uint32_t y = lane; y < (total >> 1); y += 64u) {
                    const uint32_t rr = y >> half_shift, kk = (y - (rr << half_shift)) << 1;
                    const uint32_t m = kk >= split ? 1u : 0u;
                    const uint32_t pb = in_base[m] + __umul24(wave * 32u + rr, m ? s1.ustride : s0.ustride);
                    const uint64_t* row = lds64 + ((pb + 4u + 7u) >> 3) + (m ? kk - split : kk);
                    const u64x2 v = {row[0], row[1]};
                    __builtin_nontemporal_store(v, &gout2[y]);
                }
            } else
#endif
            for (uint32_t x = lane; x < total; x += 64u) {
                const uint32_t rr = __umulhi(x, rw_magic), kk = x - rr * row_words;
                const uint32_t m = kk >= split ? 1u : 0u;
                const uint32_t kw = m ? kk - split : kk;
                const uint32_t pb = in_base[m] + (wave * 32u + rr) * (m ? s1.ustride : s0.ustride);
                __builtin_nontemporal_store(lds64[((pb + 4u + 7u) >> 3) + kw], &gout[x]);
            }
        }
        __syncthreads();
    }
    if (h1.hist) hist1_flush(lhist, h1.hist);
}

// ---------------------------------------------------------------------------
// Exact key comparison (setRecord::operator== / setRecordPair::operator==).
// All loads of a 4-word chunk are issued before any compare so a tag match costs one
// memory round trip per chunk, not one per word.
__device__ __forceinline__ bool keys_equal(const KeyStore& ks, uint32_t a, uint32_t b)
{
    const uint64_t* __restrict__ p = ks.slot(a);
    const uint64_t* __restrict__ q = ks.slot(b);
    uint32_t W = ks.W0;
    if (ks.koff) {
        const uint64_t ha = p[0];
        if (ha != q[0]) return false;                 // mate lengths differ
        W = seg_words(uint32_t(ha)) + seg_words(uint32_t(ha >> 32));
        ++p; ++q;
    }
    uint32_t k = 0;
    for (; k + 8u <= W; k += 8u) {
        uint64_t x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = p[k + j]; y[j] = q[k + j]; }
        uint64_t d = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d |= x[j] ^ y[j];
        if (d) return false;
    }
    uint64_t d = 0;
    for (; k < W; ++k) d |= p[k] ^ q[k];
    return d == 0;
}

// The table is cut into segments of 2^seg_bits slots (4096..16384, or the whole table when it
// is smaller); a key's probe sequence wraps inside the segment its hash selects.  That keeps the
// incremental (atomic) path below and the bulk path (bucket_dedup_kernel, one segment per
// workgroup in LDS) on one and the same table layout.
//
// insert: one record per lane.  Slot = (tag:32 | record index:32), EMPTY = all ones.
//   * atomicCAS(EMPTY -> mine) claims a free slot: the record is (so far) the first of its key.
//   * a slot whose tag matches is verified word-for-word against the owner's stored key;
//     only then atomicMin(slot, mine) decides who is first: the loser's keep flag is cleared
//     (mine if the owner is older, the displaced owner's if I am older).
//   * keep[] was preset to 1; flags are only ever cleared, so concurrent order does not matter.
// Records of earlier batches have smaller indices and can never be displaced, so a batch's
// flags are final when its launch retires.
__global__ __launch_bounds__(kBlock)
void insert_kernel(uint64_t* __restrict__ table, uint64_t slot_mask, uint64_t seg_mask, KeyStore ks,
                   const uint64_t* __restrict__ hashes, uint32_t hash_stride,
                   uint64_t n, Verdicts out, uint32_t tag_mask,
                   unsigned long long* __restrict__ counters /* [0]=dups [1]=table-full */)
{
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(table);
    uint32_t dups = 0, lost = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const uint64_t h = hashes[i * uint64_t(hash_stride)];
        if (h == kSkipHash) continue;                           // no record at this position
        const uint32_t idx = out.first_idx + uint32_t(i);
        const uint64_t tag = slot_tag(h, tag_mask);
        const unsigned long long mine = (tag << 32) | idx;
        uint64_t pos = h & slot_mask;
        bool placed = false;
        for (uint64_t probe = 0; probe <= seg_mask; ++probe) {
            // A plain load first: an occupied slot never empties and never changes its key, so only a
            // slot that looks empty is worth one of the (much scarcer) memory-side atomics.  A stale
            // view is harmless: a stale EMPTY makes the CAS fail and return the truth, a stale owner
            // is a younger record of the same key and is sorted out by the atomicMin below.
            unsigned long long old = __hip_atomic_load(&tab[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // from L2, never a stale L1 line
            if (old == kEmptySlot) {
                old = atomicCAS(&tab[pos], kEmptySlot, mine);
                if (old == kEmptySlot) { placed = true; break; }
            }
            if ((old >> 32) == tag && keys_equal(ks, idx, uint32_t(old))) {
                // The slot now belongs to my key for good.  If its owner is older I lose and the
                // table needs no update; only an owner younger than me has to be displaced.
                uint32_t owner = uint32_t(old);
                if (owner > idx) owner = uint32_t(atomicMin(&tab[pos], mine));
                if (owner < idx) out.lose(idx, owner);              // an earlier record holds this key
                else             out.lose(owner, idx);              // I am earlier: the displaced one loses
                ++dups;
                placed = true;
                break;
            }
            pos = (pos & ~seg_mask) | ((pos + 1) & seg_mask);       // probing never leaves the key's segment
        }
        if (!placed) ++lost;
    }
    // one counter update per wave, not per duplicate (a single hot address serialises)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { dups += __shfl_down(dups, d, 64); lost += __shfl_down(lost, d, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (dups) atomicAdd(&counters[0], static_cast<unsigned long long>(dups));
        if (lost) atomicAdd(&counters[1], static_cast<unsigned long long>(lost));
    }
}

// rehash: move every owner into a larger table (keys are distinct: no verification).
__global__ __launch_bounds__(kBlock)
void rehash_kernel(const uint64_t* __restrict__ old_table, uint64_t old_slots,
                   uint64_t* __restrict__ new_table, uint64_t new_mask, uint64_t new_seg_mask, KeyStore ks,
                   uint32_t len0, uint32_t len1, uint32_t paired, uint64_t hash_and,
                   uint32_t new_tag_mask, unsigned long long* __restrict__ counters /* [1] = table-full */)
{
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(new_table);
    for (uint64_t s = blockIdx.x * uint64_t(kBlock) + threadIdx.x; s < old_slots; s += uint64_t(gridDim.x) * kBlock) {
        const uint64_t e = old_table[s];
        if (e == kEmptySlot) continue;
        const uint32_t idx = uint32_t(e);
        const uint64_t* p = ks.slot(idx);
        uint32_t l0 = len0, l1 = len1, W = ks.W0;
        if (ks.koff) { l0 = uint32_t(p[0]); l1 = uint32_t(p[0] >> 32); W = seg_words(l0) + seg_words(l1); ++p; }
        const bool opaque = !ks.koff && len1 == kOpaqueKeys;      // see hash_keys_kernel
        if (opaque) paired = 0;
        const uint32_t w0 = opaque ? W : seg_words(l0);
        uint64_t h = hash_begin(l0, 0);
        for (uint32_t k = 0; k < w0; ++k) h = hash_word(h, p[k]);
        if (paired) {                                         // second chain, then combine
            uint64_t h1 = hash_begin(l1, 0);
            for (uint32_t k = w0; k < W; ++k) h1 = hash_word(h1, p[k]);
            h = hash_pair(h, h1);
        } else {
            h = hash_end(h);
        }
        h &= hash_and;
        const unsigned long long mine = (slot_tag(h, new_tag_mask) << 32) | idx;
        uint64_t pos = h & new_mask;
        uint64_t probe = 0;                                  // bounded: a full segment is reported, never spun on
        while (atomicCAS(&tab[pos], kEmptySlot, mine) != kEmptySlot) {
            if (++probe > new_seg_mask) { atomicAdd(&counters[1], 1ull); break; }
            pos = (pos & ~new_seg_mask) | ((pos + 1) & new_seg_mask);
        }
    }
}

// ---------------------------------------------------------------------------
// Bulk insert: no global atomics on the table.
//   8-byte records (q << 32 | index) are radix-partitioned by table segment (one or two passes
//   of up to 256 ways, LDS counting sort per 8192-record tile so the scatter leaves the CU as
//   runs of whole lines); q = the table position's bits below the level-1 digit, and above them
//   the slot tag (part_q).  Then one workgroup per segment replays the same probe/verify/first-wins logic
//   as insert_kernel on an LDS-resident copy of the segment and writes it back once.
// Measured motive (tools/atomic_probe.hip): device-scope atomics cap at 18-27 G/s on this chip
// wherever the table lives, LDS atomics do not.

constexpr int kPartThreads = 1024;
constexpr int kPartPer = 8;
constexpr int kPartTile = kPartThreads * kPartPer;      // 8192 records per tile (4096: runs too short, 16384: one block per CU)

// level-1 histogram straight from the batch's hashes
__global__ __launch_bounds__(kPartThreads)
void bulk_hist1_kernel(const uint64_t* __restrict__ hashes, uint32_t hash_stride, uint64_t n, BulkGeom g,
                       uint32_t* __restrict__ hist1)
{
    __shared__ uint32_t h[256];
    for (int k = threadIdx.x; k < 256; k += kPartThreads) h[k] = 0;
    __syncthreads();
    for (uint64_t i = blockIdx.x * uint64_t(kPartThreads) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kPartThreads) {
        const uint64_t v = hashes[i * uint64_t(hash_stride)];
        if (v != kSkipHash) atomicAdd(&h[bucket_of(v, g) >> g.bits2], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 256; k += kPartThreads) if (h[k]) atomicAdd(&hist1[k], h[k]);
}

// One block: start[d] = exclusive prefix of count[d] for d < nd (nd <= 256), start[nd] = total;
// cursor = copy of start; tile_start[d] = prefix of ceil(count[d] / tile) (tiles never straddle digits).
__global__ void bulk_scan256_kernel(const uint32_t* __restrict__ count, uint32_t nd, uint32_t* __restrict__ start,
                                    uint32_t* __restrict__ cursor, uint32_t* __restrict__ tile_start)
{
    __shared__ uint32_t a[257], t[257];
    const uint32_t d = threadIdx.x;
    if (d <= 256) {
        a[d] = d < nd ? count[d] : 0u;
        t[d] = d < nd ? (count[d] + kPartTile - 1) / kPartTile : 0u;
    }
    __syncthreads();
    if (d == 0) {
        uint32_t s = 0, ts = 0;
        for (uint32_t k = 0; k <= nd; ++k) { const uint32_t c = k < nd ? a[k] : 0, tc = k < nd ? t[k] : 0; a[k] = s; t[k] = ts; s += c; ts += tc; }
    }
    __syncthreads();
    if (d <= nd) { start[d] = a[d]; tile_start[d] = t[d]; if (d < nd) cursor[d] = a[d]; }
}

// q of a partition record (see BulkGeom / slot_tag): low seg_bits + bits2 bits of the table
// position, the slot tag above them.
__device__ __forceinline__ uint32_t part_q(uint64_t hash, const BulkGeom& g)
{
    const uint32_t qshift = g.seg_bits + g.bits2;
    return (uint32_t(hash & g.slot_mask) & ((1u << qshift) - 1u)) | (uint32_t(slot_tag(hash, g.tag_mask)) << qshift);
}

// Scatter pass shared by both levels.  LEVEL 1 reads the batch's hashes (index implicit),
// LEVEL 2 reads level-1 records; the tile's records are counting-sorted by digit in LDS, each
// digit's run reserves its place with ONE atomicAdd on that digit's cursor, and the runs are
// copied out contiguously.
template <int LEVEL>
__global__ __launch_bounds__(kPartThreads, 8)            // 8 waves per SIMD = two of these workgroups per CU (LDS allows two): at most 64 VGPRs (level 2 took 65)
void bulk_scatter_kernel(const uint64_t* __restrict__ hashes, uint32_t hash_stride, uint32_t first_idx,
                         const uint64_t* __restrict__ in, uint64_t n, BulkGeom g,
                         const uint32_t* __restrict__ start1, const uint32_t* __restrict__ tile_start1,
                         uint32_t* __restrict__ cursor, uint64_t* __restrict__ out,
                         uint8_t* __restrict__ digit2_out /* LEVEL 1: level-2 digit per output record, for the level-2 count */)
{
    __shared__ uint64_t stage[kPartTile];
    __shared__ uint8_t sdig[kPartTile];
    constexpr int kBins = (LEVEL == 1) ? 256 : 512;         // level 1: up to 8 bits, level 2: up to 9
    __shared__ uint32_t cnt[kBins], lstart[kBins], gbase[kBins];
    __shared__ uint32_t n_valid;
    const uint32_t nd1 = 1u << g.bits1, mask2 = (1u << g.bits2) - 1u;
    const uint64_t n_tiles = (LEVEL == 1) ? (n + kPartTile - 1) / kPartTile : tile_start1[nd1];
    STAMP_DECL
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        STAMP(0);
        uint64_t lo, hi; uint32_t d1 = 0;
        if (LEVEL == 1) { lo = tile * kPartTile; hi = lo + kPartTile < n ? lo + kPartTile : n; }
        else {
            // which level-1 digit does this tile belong to?  (binary search over <= 256 entries)
            uint32_t a = 0, b = nd1;
            while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (tile_start1[m] <= tile) a = m; else b = m; }
            d1 = a;
            lo = start1[d1] + (tile - tile_start1[d1]) * uint64_t(kPartTile);
            hi = lo + kPartTile < start1[d1 + 1] ? lo + kPartTile : start1[d1 + 1];
        }
        const uint32_t count = uint32_t(hi - lo);
        for (int k = threadIdx.x; k < kBins; k += kPartThreads) cnt[k] = 0;
        __syncthreads();
        STAMP(1);                                             // tile lookup + clear + barrier
        uint64_t rec[kPartPer]; uint32_t dig[kPartPer], rank[kPartPer];
#pragma unroll
        for (int k = 0; k < kPartPer; ++k) {                  // every load of the tile first (index clamped): one round trip
            const uint32_t r = threadIdx.x + k * kPartThreads;
            const uint64_t at = lo + (r < count ? r : count - 1u);
            rec[k] = (LEVEL == 1) ? hashes[at * uint64_t(hash_stride)] : in[at];
        }
        STAMP_LOADS_IN(); STAMP(10);                          // (diagnostic build) the tile's round trip, apart from the ranking
#pragma unroll
        for (int k = 0; k < kPartPer; ++k) {
            const uint32_t r = threadIdx.x + k * kPartThreads;
            dig[k] = 0xFFFFFFFFu;                                // no record: past the tile's end, or a skipped position
            if (r < count && (LEVEL != 1 || rec[k] != kSkipHash)) {
                if (LEVEL == 1) {
                    const uint64_t h = rec[k];
                    rec[k] = (uint64_t(part_q(h, g)) << 32) | (first_idx + uint32_t(lo + r));
                    dig[k] = bucket_of(h, g) >> g.bits2;
                } else {
                    dig[k] = (uint32_t(rec[k] >> 32) >> g.seg_bits) & mask2;
                }
                rank[k] = atomicAdd(&cnt[dig[k]], 1u);
            }
        }
        STAMP(2);                                             // loads + ranks (LDS atomics), my wave
        __syncthreads();
        STAMP(3);                                             // ... the other waves
        if (threadIdx.x < 64) {                               // exclusive scan of the counts by one wave
            constexpr int kPer = kBins / 64;
            uint32_t c[kPer], s = 0;
#pragma unroll
            for (int k = 0; k < kPer; ++k) { c[k] = cnt[threadIdx.x * kPer + k]; s += c[k]; }
            uint32_t inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(threadIdx.x) >= d) inc += up; }
            uint32_t ex = inc - s;
#pragma unroll
            for (int k = 0; k < kPer; ++k) { lstart[threadIdx.x * kPer + k] = ex; ex += c[k]; }
            if (threadIdx.x == 63) n_valid = inc;
        }
        __syncthreads();
        STAMP(4);                                             // scan + barrier
        for (int k = threadIdx.x; k < kBins; k += kPartThreads)
            gbase[k] = cnt[k] ? atomicAdd(&cursor[(LEVEL == 1 ? 0u : (d1 << g.bits2)) + k], cnt[k]) : 0u;
        STAMP(5);                                             // cursor atomics (wave 0 has them)
#pragma unroll
        for (int k = 0; k < kPartPer; ++k)
            if (dig[k] != 0xFFFFFFFFu) {
                const uint32_t at = lstart[dig[k]] + rank[k];
                stage[at] = rec[k];
                if (LEVEL == 1) sdig[at] = uint8_t(dig[k]);       // level 2 finds its digit in the record itself
            }
        STAMP(6);                                             // stage writes
        __syncthreads();
        STAMP(7);                                             // barrier
        const uint32_t valid = n_valid;
#pragma unroll
        for (int k = 0; k < kPartPer; ++k) {
            const uint32_t r = threadIdx.x + k * kPartThreads;
            if (r < valid) {
                const uint64_t v = stage[r];
                const uint32_t d = (LEVEL == 1) ? uint32_t(sdig[r]) : ((uint32_t(v >> 32) >> g.seg_bits) & mask2);
                out[gbase[d] + (r - lstart[d])] = v;      // plain stores: the short runs of neighbouring tiles combine in L2
                if (LEVEL == 1 && digit2_out) digit2_out[gbase[d] + (r - lstart[d])] = uint8_t((uint32_t(v >> 32) >> g.seg_bits) & mask2);
            }
        }
        STAMP(8);                                             // write-out issue
        __syncthreads();
        STAMP(9);                                             // barrier
    }
    STAMP_FLUSH(LEVEL - 1);
}

// Level-2 histogram over the level-1 output (read through the 1-byte level-2 digits the
// level-1 scatter leaves beside the records: an eighth of the bytes).  A tile lies inside one
// level-1 digit, so its counts go to consecutive buckets; LDS-aggregated, one global add per bin.
// FROM_RECS: 9-bit level-2 digits do not fit the byte array; they are read out of the records.
template <bool FROM_RECS>
__global__ __launch_bounds__(kPartThreads)
void bulk_hist2_kernel(const uint8_t* __restrict__ digit2_in, const uint64_t* __restrict__ recs, BulkGeom g, const uint32_t* __restrict__ start1,
                       const uint32_t* __restrict__ tile_start1, uint32_t* __restrict__ hist2)
{
    __shared__ uint32_t h[512];
    const uint32_t nd1 = 1u << g.bits1, nd2 = 1u << g.bits2;
    const uint64_t n_tiles = tile_start1[nd1];
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t a = 0, b = nd1;
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (tile_start1[m] <= tile) a = m; else b = m; }
        const uint32_t d1 = a;
        const uint64_t lo = start1[d1] + (tile - tile_start1[d1]) * uint64_t(kPartTile);
        const uint64_t hi = lo + kPartTile < start1[d1 + 1] ? lo + kPartTile : start1[d1 + 1];
        for (uint32_t k = threadIdx.x; k < nd2; k += kPartThreads) h[k] = 0;
        __syncthreads();
        uint32_t dg[kPartPer];                                // a tile is kPartPer digits per lane: all loads first
#pragma unroll
        for (int k = 0; k < kPartPer; ++k) {
            const uint64_t r = lo + threadIdx.x + uint64_t(k) * kPartThreads, at = r < hi ? r : hi - 1;
            dg[k] = FROM_RECS ? (uint32_t(recs[at] >> 32) >> g.seg_bits) : uint32_t(digit2_in[at]);
        }
#pragma unroll
        for (int k = 0; k < kPartPer; ++k)
            if (lo + threadIdx.x + uint64_t(k) * kPartThreads < hi) atomicAdd(&h[dg[k] & (nd2 - 1u)], 1u);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nd2; k += kPartThreads) if (h[k]) atomicAdd(&hist2[(d1 << g.bits2) + k], h[k]);
        __syncthreads();
    }
}

// One 512-thread block per level-1 digit d1: the level-1 output is grouped by d1, so the buckets
// (d1, 0..nd2) start at start1[d1] plus the exclusive prefix of their own counts — no scan over
// all buckets is needed.  start[b] for every bucket, start[nb] = total; cursor = start.
__global__ __launch_bounds__(512)
void bulk_scan_buckets_kernel(const uint32_t* __restrict__ count, uint32_t bits2, const uint32_t* __restrict__ start1,
                              uint32_t nd1, uint32_t* __restrict__ start, uint32_t* __restrict__ cursor)
{
    __shared__ uint32_t wave_tot[8];
    const uint32_t nd2 = 1u << bits2, d1 = blockIdx.x, t = threadIdx.x;
    const uint32_t c = t < nd2 ? count[(d1 << bits2) + t] : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(t & 63u) >= d) inc += up; }
    if ((t & 63u) == 63u) wave_tot[t >> 6] = inc;
    __syncthreads();
    uint32_t before = start1[d1];
    for (uint32_t w = 0; w < (t >> 6); ++w) before += wave_tot[w];
    if (t < nd2) { start[(d1 << bits2) + t] = before + inc - c; cursor[(d1 << bits2) + t] = before + inc - c; }
    if (d1 == nd1 - 1u && t == 0) start[nd1 << bits2] = start1[nd1];
}

// One workgroup per table segment.  FRESH: the segment is known to be empty (engine just reset):
// it is built in LDS from scratch and written out, so the table needs no clearing pass.
//
// A chunk of the segment's records goes through three phases, so that the two random 64-byte
// key gathers a duplicate needs are never waited for one record at a time (measured: with the
// verification inline, every loop iteration of every wave paid a full gather round trip and
// 20 M duplicates cost 1.6 ms of a 2.6 ms kernel):
//   1. probe: every lane walks its records through the LDS table (CAS on EMPTY claims; a tag
//      match is only QUEUED as a candidate);
//   2. verify: eight lanes per candidate load one key word each of the record and of the slot's
//      current owner — whole lines, dozens of candidates in flight per wave — and lane 0 applies
//      the verdict (atomicMin / keep flags) or queues a retry from the next slot;
//   3. retry (rare: a tag matched an unequal key): per-lane probing with inline verification.
// A queue entry is self-contained — (tag << seg_bits | slot) << 32 | record index — so the verify
// phase goes straight from LDS to the two key gathers.  Candidates fill the queue from the front,
// retries from the back; a retry that would reach the candidates is walked on the spot instead.
// LDS: segment (2^seg_bits * 8 B) + kDedupChunk queue entries of 8 B + 2 counters.
// RAGGED = the key store is ragged (ks.koff): compiled apart so that neither verify variant pays
// for the other's registers.
constexpr uint32_t kDedupChunk = 1536;                   // queue entries (candidates from the front, retries from the back)
constexpr uint32_t kDedupRecords = 3 * kDedupChunk;      // records per chunk: a segment's whole bucket, normally; a candidate that
                                                         // finds the queue full is verified on the spot
constexpr uint32_t kDedupFly = 4;
#ifndef FQD_DEDUP_LOADS
#define FQD_DEDUP_LOADS 6
#endif
#ifndef FQD_DEDUP_EARLY
#define FQD_DEDUP_EARLY 1
#endif
constexpr uint32_t kDedupLoads = FQD_DEDUP_LOADS;                      // records a lane fetches at once in the probe phase: a whole bucket at 512 threads, one
                                                         // round trip (3: two round trips per bucket; the phase stamps of round 3 put 48 % of the
                                                         // kernel's workgroup time in this phase: profiles/r03_phase_stamps_before.txt)

// VL: lanes per candidate in the verify phase of uniform key stores.  VL = 4 / 8: every lane
// loads 16 bytes of each of the two keys (keys of up to 8 / 16 words with an even word count: 150 bp
// single-end = 8 words = 4 lanes, 2 x 150 bp = 16 words = 8 lanes), so a 512-thread workgroup has
// 128 / 64 groups of kFly candidates in flight; VL = 0: eight lanes, one or two 8-byte words each
// (any length; ragged stores always).
template <bool FRESH, bool RAGGED, int VL>
__global__ __launch_bounds__(1024)
void bucket_dedup_kernel(const uint64_t* __restrict__ recs, const uint32_t* __restrict__ bstart, uint32_t n_buckets,
                         uint64_t* __restrict__ table, uint32_t seg_bits, uint32_t qshift, KeyStore ks, Verdicts out,
                         unsigned long long* __restrict__ counters,
                         uint32_t heavy_above, uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list,
                         uint32_t write_back)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long seg[];
    const uint32_t seg_slots = 1u << seg_bits, seg_mask = seg_slots - 1u;
    unsigned long long* queue = seg + seg_slots;
    uint32_t* qn = reinterpret_cast<uint32_t*>(queue + kDedupChunk);     // qn[0] candidates, qn[1] retries
    uint32_t dups = 0, lost = 0;

    // One record's walk from `pos`: claims, or (inline == false) queues the first tag match,
    // or (inline == true) verifies tag matches on the spot.
    auto walk = [&](uint32_t idx, uint32_t tag32, uint32_t pos, bool verify_inline) {
        const uint64_t tag = tag32;
        const unsigned long long mine = (tag << 32) | idx;
        for (uint32_t probe = 0; probe < seg_slots; ++probe) {
            const unsigned long long old = atomicCAS(&seg[pos], kEmptySlot, mine);
            if (old == kEmptySlot) return;
            if ((old >> 32) == tag) {
                if (!verify_inline) {
                    const uint32_t at = atomicAdd(&qn[0], 1u);
                    if (at < kDedupChunk) { queue[at] = (uint64_t((tag32 << seg_bits) | pos) << 32) | idx; return; }
                    // queue full (a bucket with far more duplicates than expected): settle this one right here
                }
                if (keys_equal(ks, idx, uint32_t(old))) {
                    uint32_t owner = uint32_t(old);
                    if (owner > idx) owner = uint32_t(atomicMin(&seg[pos], mine));
                    if (owner < idx) out.lose(idx, owner);
                    else             out.lose(owner, idx);
                    ++dups;
                    return;
                }
            }
            pos = (pos + 1u) & seg_mask;
        }
        ++lost;
    };

    STAMP_DECL
    for (uint32_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        STAMP(0);
        unsigned long long* gseg = reinterpret_cast<unsigned long long*>(table) + uint64_t(b) * seg_slots;
        const uint32_t lo = bstart[b], hi = bstart[b + 1];
        if (!FRESH && lo == hi) continue;                     // nothing to add: leave the segment alone
        if (hi - lo > heavy_above) {
            // A bucket swollen by one massively repeated key (poly-G reads, adapter dimers) would
            // pin a single workgroup for the whole batch: leave it to heavy_bucket_insert_kernel,
            // which spreads its records over the chip with the atomic path.
            if (FRESH) for (uint32_t k = threadIdx.x; k < seg_slots; k += blockDim.x) gseg[k] = kEmptySlot;
            if (threadIdx.x == 0) heavy_list[atomicAdd(heavy_count, 1u)] = b;
            continue;
        }
        // the bucket's first records are requested before the segment is set up in LDS: the fill hides in their round trip
        uint64_t v0[kDedupLoads];
        {
            const uint32_t first_n = hi - lo < kDedupRecords ? hi - lo : kDedupRecords;
#pragma unroll
            for (uint32_t u = 0; u < kDedupLoads; ++u) { const uint32_t c = threadIdx.x + u * blockDim.x; v0[u] = (FQD_DEDUP_EARLY && first_n) ? recs[lo + (c < first_n ? c : first_n - 1u)] : 0ull; }
        }
        {   // 16 bytes per lane and access; loading an existing segment keeps four loads per lane in flight
            ulonglong2* seg2 = reinterpret_cast<ulonglong2*>(seg);
            const ulonglong2* gseg2 = reinterpret_cast<const ulonglong2*>(gseg);
            const uint32_t n2 = seg_slots >> 1;
            if (FRESH) {
                for (uint32_t k = threadIdx.x; k < n2; k += blockDim.x) seg2[k] = ulonglong2{kEmptySlot, kEmptySlot};
            } else {
                for (uint32_t k0 = threadIdx.x; k0 < n2; k0 += 4u * blockDim.x) {
                    ulonglong2 v[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) { const uint32_t k = k0 + u * blockDim.x; v[u] = gseg2[k < n2 ? k : n2 - 1u]; }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) { const uint32_t k = k0 + u * blockDim.x; if (k < n2) seg2[k] = v[u]; }
                }
            }
        }
        for (uint32_t chunk_lo = lo; chunk_lo < hi; chunk_lo += kDedupRecords) {
            const uint32_t chunk_n = hi - chunk_lo < kDedupRecords ? hi - chunk_lo : kDedupRecords;
            if (threadIdx.x == 0) { qn[0] = 0; qn[1] = 0; }
            __syncthreads();
            STAMP(1);                                         // bucket bounds + segment init/load + barrier
            // 1. probe (a lane's records are all fetched before the first walk: one round trip, not several)
            for (uint32_t c0 = threadIdx.x; c0 < chunk_n; c0 += kDedupLoads * blockDim.x) {
                uint64_t v[kDedupLoads];
                if (FQD_DEDUP_EARLY && chunk_lo == lo && c0 == threadIdx.x) {
#pragma unroll
                    for (uint32_t u = 0; u < kDedupLoads; ++u) v[u] = v0[u];
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < kDedupLoads; ++u) { const uint32_t c = c0 + u * blockDim.x; v[u] = recs[chunk_lo + (c < chunk_n ? c : chunk_n - 1u)]; }
                }
                STAMP_LOADS_IN(); STAMP(9);                   // (diagnostic build) the records' round trip, apart from the walks
#pragma unroll
                for (uint32_t u = 0; u < kDedupLoads; ++u) {
                    if (c0 + u * blockDim.x < chunk_n) {
                        const uint32_t q = uint32_t(v[u] >> 32);
                        walk(uint32_t(v[u]), q >> qshift, q & seg_mask, false);
                    }
                }
            }
            STAMP(2);                                         // probe, my wave
            __syncthreads();
            STAMP(3);                                         // probe, the others
            // 2. verify: VL (or eight) lanes per candidate, kFly candidates per group in flight
            const uint32_t n_cand = qn[0] < kDedupChunk ? qn[0] : kDedupChunk;
            constexpr uint32_t kFly = RAGGED ? 3u : (VL == 4 ? 6u : kDedupFly);     // ragged keys need more registers per candidate
            constexpr uint32_t kLanes = VL ? uint32_t(VL) : 8u;
            const uint32_t grp = threadIdx.x / kLanes, n_grp = blockDim.x / kLanes, sub = threadIdx.x % kLanes;
            for (uint32_t q0 = grp; q0 < n_cand; q0 += n_grp * kFly) {
                uint32_t pos[kFly], idx[kFly], seen[kFly], tag[kFly];
                uint64_t diff[kFly];
                bool live[kFly];
                uint64_t wa[kFly][2], wb[kFly][2];
#pragma unroll
                for (uint32_t u = 0; u < kFly; ++u) {        // issue every load of the batch first
                    const uint32_t q = q0 + u * n_grp;
                    live[u] = q < n_cand;
                    diff[u] = 0; pos[u] = 0; idx[u] = 0; seen[u] = 0; tag[u] = 0;
                    wa[u][0] = wa[u][1] = wb[u][0] = wb[u][1] = 0;
                    if (live[u]) {
                        const unsigned long long ent = queue[q];
                        idx[u] = uint32_t(ent); pos[u] = uint32_t(ent >> 32) & seg_mask; tag[u] = uint32_t(ent >> 32) >> seg_bits;
                        seen[u] = uint32_t(seg[pos[u]]);      // the slot's owner right now: same key class for good
                    }
                    if (!RAGGED && VL) {
                        // 16 bytes of each key per lane: the whole key pair of a candidate is one request per lane
                        const ulonglong2* __restrict__ pa = reinterpret_cast<const ulonglong2*>(ks.keys + idx[u] * uint64_t(ks.stride));
                        const ulonglong2* __restrict__ pb = reinterpret_cast<const ulonglong2*>(ks.keys + seen[u] * uint64_t(ks.stride));
                        if (live[u] && 2u * sub < ks.W0) {
                            const ulonglong2 x = pa[sub], y = pb[sub];
                            wa[u][0] = x.x; wa[u][1] = x.y; wb[u][0] = y.x; wb[u][1] = y.y;
                        }
                    } else if (!RAGGED) {
                        // uniform keys: addresses are arithmetic, so the (up to) two words a lane owns of
                        // each key are requested here and only looked at after the whole batch is in flight
                        const uint64_t* __restrict__ pa = ks.keys + idx[u] * uint64_t(ks.stride) + ks.lead;
                        const uint64_t* __restrict__ pb = ks.keys + seen[u] * uint64_t(ks.stride) + ks.lead;
                        const bool w0 = live[u] && sub < ks.W0, w1 = live[u] && sub + 8u < ks.W0;
                        wa[u][0] = w0 ? pa[sub] : 0; wb[u][0] = w0 ? pb[sub] : 0;
                        wa[u][1] = w1 ? pa[sub + 8u] : 0; wb[u][1] = w1 ? pb[sub + 8u] : 0;
                    } else {
                        // ragged keys, first of three dependent steps (slot offsets -> headers -> words),
                        // each taken for the whole batch before the next: wa/wb[.][0] hold the offsets for now
                        wa[u][0] = live[u] ? ks.koff[idx[u]] : 0; wb[u][0] = live[u] ? ks.koff[seen[u]] : 0;
                    }
                }
                if (RAGGED) {
                    uint64_t oa[kFly], ob[kFly];
#pragma unroll
                    for (uint32_t u = 0; u < kFly; ++u) {        // headers: mate lengths
                        oa[u] = wa[u][0]; ob[u] = wb[u][0];
                        wa[u][1] = live[u] ? ks.keys[oa[u]] : 0; wb[u][1] = live[u] ? ks.keys[ob[u]] : 0;
                    }
                    uint32_t Wr[kFly];
#pragma unroll
                    for (uint32_t u = 0; u < kFly; ++u) {        // words
                        const uint64_t ha = wa[u][1];
                        diff[u] = ha ^ wb[u][1];
                        Wr[u] = seg_words(uint32_t(ha)) + seg_words(uint32_t(ha >> 32));
                        const bool same = live[u] && diff[u] == 0;
                        const uint64_t* __restrict__ pa = ks.keys + oa[u] + 1;
                        const uint64_t* __restrict__ pb = ks.keys + ob[u] + 1;
                        const bool w0 = same && sub < Wr[u], w1 = same && sub + 8u < Wr[u];
                        wa[u][0] = w0 ? pa[sub] : 0; wb[u][0] = w0 ? pb[sub] : 0;
                        wa[u][1] = w1 ? pa[sub + 8u] : 0; wb[u][1] = w1 ? pb[sub + 8u] : 0;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kFly; ++u) {
                        if (!live[u] || diff[u]) continue;
                        diff[u] = (wa[u][0] ^ wb[u][0]) | (wa[u][1] ^ wb[u][1]);
                        if (Wr[u] > 16u) {
                            const uint64_t* __restrict__ pa = ks.keys + oa[u] + 1;
                            const uint64_t* __restrict__ pb = ks.keys + ob[u] + 1;
                            for (uint32_t w = sub + 16u; w < Wr[u]; w += 8u) diff[u] |= pa[w] ^ pb[w];
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < kFly; ++u) {
                        if (!live[u]) continue;
                        diff[u] = (wa[u][0] ^ wb[u][0]) | (wa[u][1] ^ wb[u][1]);
                        if (!VL && ks.W0 > 16u) {             // longer keys: the remaining words, the plain way
                            const uint64_t* __restrict__ pa = ks.keys + idx[u] * uint64_t(ks.stride) + ks.lead;
                            const uint64_t* __restrict__ pb = ks.keys + seen[u] * uint64_t(ks.stride) + ks.lead;
                            for (uint32_t w = sub + 16u; w < ks.W0; w += 8u) diff[u] |= pa[w] ^ pb[w];
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kFly; ++u) {
                    uint64_t d = diff[u];
                    d |= __shfl_xor(d, 1, 64); d |= __shfl_xor(d, 2, 64);
                    if (kLanes == 8u) d |= __shfl_xor(d, 4, 64);
                    if (live[u] && sub == 0u) {
                        if (d == 0) {
                            const unsigned long long mine = (uint64_t(tag[u]) << 32) | idx[u];
                            uint32_t owner = seen[u];
                            if (owner > idx[u]) owner = uint32_t(atomicMin(&seg[pos[u]], mine));
                            if (owner < idx[u]) out.lose(idx[u], owner);
                            else                out.lose(owner, idx[u]);
                            ++dups;
                        } else {
                            const uint32_t next = (pos[u] + 1u) & seg_mask;
                            const uint32_t at = kDedupChunk - 1u - atomicAdd(&qn[1], 1u);
                            if (at >= n_cand && at < kDedupChunk) queue[at] = (uint64_t((tag[u] << seg_bits) | next) << 32) | idx[u];
                            else                                   walk(idx[u], tag[u], next, true);      // queue full: settle it now
                        }
                    }
                }
            }
            STAMP(4);                                         // verify, my wave
            __syncthreads();
            STAMP(5);                                         // verify, the others
            // 3. retry the few whose tag matched an unequal key
            const uint32_t n_tried = qn[1];
            const uint32_t n_retry = n_tried < kDedupChunk - n_cand ? n_tried : kDedupChunk - n_cand;   // the rest were settled in phase 2
            for (uint32_t q = threadIdx.x; q < n_retry; q += blockDim.x) {
                const unsigned long long ent = queue[kDedupChunk - 1u - q];
                walk(uint32_t(ent), uint32_t(ent >> 32) >> seg_bits, uint32_t(ent >> 32) & seg_mask, true);
            }
            __syncthreads();
            STAMP(6);                                         // retry + barrier
        }
        __syncthreads();
        if (write_back) {                                     // 0: the run's last batch (fqd_submit_final) — nobody reads the table again
            const ulonglong2* seg2 = reinterpret_cast<const ulonglong2*>(seg);
            ulonglong2* gseg2 = reinterpret_cast<ulonglong2*>(gseg);
            for (uint32_t k = threadIdx.x; k < (seg_slots >> 1); k += blockDim.x) {   // written once, not read again by this launch
                const ulonglong2 v = seg2[k];
                __builtin_nontemporal_store(v.x, &gseg2[k].x);
                __builtin_nontemporal_store(v.y, &gseg2[k].y);
            }
        }
        STAMP(7);                                             // write-back issue
        __syncthreads();
        STAMP(8);                                             // barrier
    }
    STAMP_FLUSH(2);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { dups += __shfl_down(dups, d, 64); lost += __shfl_down(lost, d, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (dups) atomicAdd(&counters[0], static_cast<unsigned long long>(dups));
        if (lost) atomicAdd(&counters[1], static_cast<unsigned long long>(lost));
    }
}

// Second half of the skew guard: records of the buckets the LDS kernel skipped (it lists them)
// go through the device-atomic protocol of insert_kernel, spread over the whole grid.  Launched
// after every bulk dedup; when no bucket was heavy every workgroup returns on its first load.
__global__ __launch_bounds__(kBlock)
void heavy_bucket_insert_kernel(const uint64_t* __restrict__ recs, const uint32_t* __restrict__ bstart,
                                BulkGeom g, uint64_t* __restrict__ table, KeyStore ks, Verdicts out,
                                unsigned long long* __restrict__ counters,
                                const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list)
{
    const uint32_t n_heavy = *heavy_count;
    if (n_heavy == 0u) return;
    unsigned long long* tab = reinterpret_cast<unsigned long long*>(table);
    const uint64_t seg_mask = (1ull << g.seg_bits) - 1ull;
    const uint32_t qshift = g.seg_bits + g.bits2;
    uint32_t dups = 0, lost = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t k = 0; k < n_heavy; ++k) {
        const uint32_t b = heavy_list[k];
        const uint64_t lo = bstart[b], hi = bstart[b + 1];
        // wave-uniform trip count: the lanes of a wave settle their duplicates together below
        for (uint64_t r0 = lo + blockIdx.x * uint64_t(kBlock) + (threadIdx.x & ~63u); r0 < hi; r0 += uint64_t(gridDim.x) * kBlock) {
            const uint64_t r = r0 + lane;
            const bool live = r < hi;
            const uint64_t v = live ? recs[r] : 0;
            const uint32_t idx = uint32_t(v), q = uint32_t(v >> 32);
            const uint64_t tag = q >> qshift;
            const unsigned long long mine = (tag << 32) | idx;
            uint64_t pos = (uint64_t(b) << g.seg_bits) | (q & seg_mask);
            // 1. probe, the wave in step: claim an empty slot, or stop at the slot that holds my key.
            //    Of the lanes that see the same empty slot only the earliest record tries the CAS
            //    (the others look again): a million copies of one key starting together would
            //    otherwise queue up on one address, one atomic each.
            bool done = !live, found = false;
            uint32_t owner = 0, steps = 0;
            while (__any(!done)) {
                unsigned long long old = done ? 0ull : __hip_atomic_load(&tab[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // from L2, never a stale L1 line
                const bool empty = !done && old == kEmptySlot;
                bool try_claim = false;
                unsigned long long e_todo = __ballot(empty);
                while (e_todo) {
                    const int leader = __ffsll(static_cast<long long>(e_todo)) - 1;
                    const uint64_t lpos = __shfl(pos, leader, 64);
                    const bool same = empty && pos == lpos;
                    const unsigned long long group = __ballot(same);
                    uint32_t m = same ? idx : 0xFFFFFFFFu;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = o < m ? o : m; }
                    if (same && idx == m) try_claim = true;
                    e_todo &= ~group;
                }
                if (try_claim) {
                    old = atomicCAS(&tab[pos], kEmptySlot, mine);
                    if (old == kEmptySlot) done = true;
                }
                if (!done && (!empty || try_claim)) {              // `old` is a real owner now
                    if ((old >> 32) == tag && keys_equal(ks, idx, uint32_t(old))) { owner = uint32_t(old); found = true; done = true; }
                    else if (++steps > seg_mask) { ++lost; done = true; }
                    else pos = (pos & ~seg_mask) | ((pos + 1) & seg_mask);
                }
            }
            // 2. the wave's duplicates, slot by slot: only the earliest record of the wave goes to the
            //    table (one atomicMin instead of up to 64 on the same address), the others lose to it
            unsigned long long todo = __ballot(found);
            while (todo) {
                const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                const uint64_t lpos = __shfl(pos, leader, 64);
                const bool mine_too = found && pos == lpos;
                const unsigned long long group = __ballot(mine_too);
                uint32_t m = mine_too ? idx : 0xFFFFFFFFu;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(m, d, 64); m = o < m ? o : m; }
                if (mine_too) {
                    if (idx == m) {
                        uint32_t prev = owner;
                        if (prev > idx) prev = uint32_t(atomicMin(&tab[pos], mine));
                        if (prev < idx) out.lose(idx, prev);        // an earlier record holds this key
                        else            out.lose(prev, idx);        // I am earlier: the displaced one loses
                    } else {
                        out.lose(idx, m);
                    }
                    ++dups;
                }
                todo &= ~group;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { dups += __shfl_down(dups, d, 64); lost += __shfl_down(lost, d, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (dups) atomicAdd(&counters[0], static_cast<unsigned long long>(dups));
        if (lost) atomicAdd(&counters[1], static_cast<unsigned long long>(lost));
    }
}

// ---------------------------------------------------------------------------
// Ragged bookkeeping: words per record, exclusive scan, uniform -> ragged re-layout.

// need[i] = 1 (header) + key words of record i
template <int S>
__global__ __launch_bounds__(kBlock)
void slot_words_kernel(SegView s0, SegView s1, uint64_t n, uint64_t* __restrict__ need)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock)
        need[i] = 1u + seg_words(s0.len(i)) + ((S == 2) ? seg_words(s1.len(i)) : 0u);
}

constexpr int kScanItems = 4;                       // per thread
constexpr int kScanTile = kBlock * kScanItems;      // 1024 per block

// In-place exclusive scan of each 1024-item tile; tile totals go to sums[block].
__global__ __launch_bounds__(kBlock)
void scan_tiles_kernel(uint64_t* __restrict__ data, uint64_t n, uint64_t* __restrict__ sums)
{
    __shared__ uint64_t wave_tot[kBlock / 64];
    const uint64_t base = blockIdx.x * uint64_t(kScanTile) + threadIdx.x * uint64_t(kScanItems);
    uint64_t v[kScanItems], run = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = (base + k < n) ? data[base + k] : 0; run += v[k]; }
    // inclusive scan of `run` across the wave
    uint64_t inc = run;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    const int wave = threadIdx.x >> 6;
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint64_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    uint64_t ex = before + inc - run;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == kBlock - 1) sums[blockIdx.x] = before + inc;
}

// data[i] += tile_prefix[i / 1024] + add
__global__ __launch_bounds__(kBlock)
void scan_add_kernel(uint64_t* __restrict__ data, uint64_t n, const uint64_t* __restrict__ tile_prefix, uint64_t add)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock)
        data[i] += (tile_prefix ? tile_prefix[i / kScanTile] : 0) + add;
}

// Uniform -> ragged: key j moves from src + j*W0 to dst + j*(W0+1) + 1 behind a header, koff[j] = j*(W0+1).
__global__ __launch_bounds__(kBlock)
void relayout_ragged_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint64_t* __restrict__ koff,
                            uint64_t n, uint32_t W0, uint64_t header)
{
    const uint64_t total = n * uint64_t(W0 + 1u);
    for (uint64_t x = blockIdx.x * uint64_t(kBlock) + threadIdx.x; x < total; x += uint64_t(gridDim.x) * kBlock) {
        const uint64_t j = x / (W0 + 1u);
        const uint32_t k = uint32_t(x - j * (W0 + 1u));
        dst[x] = k ? src[j * W0 + (k - 1u)] : header;
        if (k == 0) koff[j] = x;
    }
}

// fqd_widen_keys: uniform keys of W0 words -> opaque keys of W1 words: `lead` header words (0 or 1) first, then the old
// words, then zeros.  One output word per lane and step: both sides stream.
__global__ __launch_bounds__(kBlock)
void widen_keys_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint64_t n, uint32_t W0, uint32_t W1,
                       uint32_t lead, uint64_t header)
{
    const uint64_t total = n * uint64_t(W1);
    for (uint64_t x = blockIdx.x * uint64_t(kBlock) + threadIdx.x; x < total; x += uint64_t(gridDim.x) * kBlock) {
        const uint64_t j = x / W1;
        const uint32_t k = uint32_t(x - j * W1);
        dst[x] = k < lead ? header : (k - lead < W0 ? src[j * W0 + (k - lead)] : 0ull);
    }
}

// ---------------------------------------------------------------------------
// Stable partition of records by owner = (hash >> 40) % n_parts  (SURVEY §8e step 2).
// counts2d is part-major: counts2d[p * n_blocks + b]; an exclusive scan of it is
// the destination of the first record of part p in block b.
__device__ __forceinline__ uint32_t owner_of(uint64_t hash, uint32_t n_parts) { return uint32_t((hash >> 40) % n_parts); }

__global__ __launch_bounds__(kBlock)
void part_count_kernel(const uint64_t* __restrict__ hashes, uint32_t hash_stride, uint64_t n, uint32_t n_parts,
                       uint64_t* __restrict__ counts2d, uint32_t n_blocks)
{
    extern __shared__ uint32_t hist[];                 // n_parts
    for (uint32_t p = threadIdx.x; p < n_parts; p += kBlock) hist[p] = 0;
    __syncthreads();
    const uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x;
    if (i < n) atomicAdd(&hist[owner_of(hashes[i * uint64_t(hash_stride)], n_parts)], 1u);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < n_parts; p += kBlock) counts2d[uint64_t(p) * n_blocks + blockIdx.x] = hist[p];
}

// The 256 records of a block are staged in LDS with coalesced loads, ranked per owner in input
// order (wave ballots, so the partition is stable), and copied out word by word: consecutive
// lanes write consecutive words, and records bound for one owner are neighbours in the output,
// so the stores leave as runs instead of one scattered 8-byte store per lane.
// LDS: 256*rec_words uint64 + 256 uint64 (destinations) + n_parts*4 uint32.  rw_magic = ceil(2^32/rec_words).
// slab_cap != 0 (fqd_partition_slabs): part p's records go to the fixed-size slab p (slots p*slab_cap ..), the ones
// beyond its capacity to the overflow region behind the last slab, part after part; slab_of[p] = {first grouped
// index of part p, first overflow slot of part p} (part_slab_offsets_kernel).
__global__ __launch_bounds__(kBlock)
void part_scatter_kernel(const uint64_t* __restrict__ rec, uint64_t n, uint32_t rec_words, uint32_t n_parts,
                         const uint64_t* __restrict__ starts2d, uint32_t n_blocks,
                         uint64_t* __restrict__ out, uint32_t* __restrict__ origin, uint32_t rw_magic,
                         uint32_t strip /* 1: leave word 0 (the hash) out of the output rows */,
                         uint64_t slab_cap, const ulonglong2* __restrict__ slab_of)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t ptile[];
    uint64_t* dst_of = ptile + size_t(kBlock) * rec_words;
    uint32_t* wave_cnt = reinterpret_cast<uint32_t*>(dst_of + kBlock);            // [n_parts][4 waves]
    const uint64_t base = blockIdx.x * uint64_t(kBlock);
    const uint32_t cnt = uint32_t(n - base < kBlock ? n - base : kBlock);
    const uint32_t words = cnt * rec_words;
    const uint64_t* src = rec + base * rec_words;
    for (uint32_t w0 = threadIdx.x; w0 < words; w0 += 5u * kBlock) {       // five loads per lane in flight (see stage_chunks)
        uint64_t v[5];
#pragma unroll
        for (uint32_t k = 0; k < 5u; ++k) { const uint32_t w = w0 + k * kBlock; v[k] = src[w < words ? w : words - 1u]; }
#pragma unroll
        for (uint32_t k = 0; k < 5u; ++k) { const uint32_t w = w0 + k * kBlock; if (w < words) ptile[w] = v[k]; }
    }
    __syncthreads();
    const bool live = threadIdx.x < cnt;
    const uint32_t mine = live ? owner_of(ptile[threadIdx.x * rec_words], n_parts) : 0xFFFFFFFFu;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t rank_in_wave = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const unsigned long long m = __ballot(mine == p);
        if (mine == p) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[p * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    if (live) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[mine * 4 + w];
        uint64_t dst = starts2d[uint64_t(mine) * n_blocks + blockIdx.x] + before + rank_in_wave;
        if (slab_cap) {
            const ulonglong2 so = slab_of[mine];
            const uint64_t local = dst - so.x;
            dst = local < slab_cap ? uint64_t(mine) * slab_cap + local : so.y + (local - slab_cap);
        }
        dst_of[threadIdx.x] = dst;
        origin[dst] = uint32_t(base + threadIdx.x);
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < words; w += kBlock) {
        const uint32_t r = rec_words == 1u ? w : __umulhi(w, rw_magic), k = w - r * rec_words;
        if (k >= strip) out[dst_of[r] * (rec_words - strip) + (k - strip)] = ptile[w];
    }
}

// slab_of[p] = {first grouped index of part p, first overflow slot of part p}: parts that exceed slab_cap spill,
// in part order, into the slots from n_parts * slab_cap on.  One thread: n_parts is the number of GPUs.
__global__ void part_slab_offsets_kernel(const uint64_t* __restrict__ starts2d, uint32_t n_parts, uint32_t n_blocks,
                                         uint64_t n, uint64_t slab_cap, ulonglong2* __restrict__ slab_of)
{
    if (blockIdx.x || threadIdx.x) return;
    uint64_t spill = uint64_t(n_parts) * slab_cap;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const uint64_t lo = starts2d[uint64_t(p) * n_blocks];
        const uint64_t hi = (p + 1 < n_parts) ? starts2d[uint64_t(p + 1) * n_blocks] : n;
        slab_of[p] = ulonglong2{lo, spill};
        if (hi - lo > slab_cap) spill += hi - lo - slab_cap;
    }
}

// ---------------------------------------------------------------------------
// encode_chunks: the source side of the sharded exchange in ONE pass (the three-step form — encode 72-byte records,
// count owners, scan, copy the keys to their slabs — writes every key twice and reads it once more: 378 bytes of
// traffic per 150-base read against 218 here).  Input order has to survive the grouping (it is what makes
// first-occurrence-wins global), so a key's place in its owner's slab would depend on every read before it — unless
// the slab is CUT the way the input is: the batch is cut into chunks of chunk_reads consecutive reads, every owner's
// slab into one sub-slab of sub_cap slots per chunk (a fair share of a chunk plus 5.5 standard deviations), and chunk
// c's keys for owner p go, in input order, to sub-slab (p, c).  A workgroup takes a whole chunk (an atomic counter
// hands them out), walks its tiles in order with one running count per owner in LDS, and needs nobody else's
// counts: lanes rank themselves among the tile's reads of the same owner with wave ballots, the wave streams its
// rows to their slots straight out of LDS.  (A first version chained the counts of ALL tiles by a decoupled
// look-back instead: with tiles of 256 reads ~1000 of them are in flight and each summed its way back through
// most of them — the encoder took twice its time; git 4299492..) .  counts[p * n_chunks + c] = keys of chunk c for
// owner p — the TRUE count: a key that finds its sub-slab full is not written, and the caller groups that batch
// again the three-step way (fqd_encode_slabs, FQD_SLABS_EXACT) — totals[p] += the same.  n_parts <= kGroupParts.
constexpr uint32_t kGroupParts = 16;

__global__ __launch_bounds__(kBlock)
void encode_chunks_kernel(SegView s0, uint64_t n, uint32_t W0, uint32_t n_parts, uint32_t chunk_tiles, uint32_t n_chunks, uint32_t used_chunks, uint64_t sub_cap,
                          uint64_t* __restrict__ out_keys, uint32_t* __restrict__ origin, uint64_t* __restrict__ counts,
                          unsigned long long* __restrict__ totals, uint32_t* __restrict__ next_chunk,
                          uint64_t* __restrict__ err, uint32_t rw_magic, uint64_t hash_and,
                          uint64_t* __restrict__ out_hash /* nullable: the key's placement hash to the same slot of a second array (it travels with the key) */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t wave_cnt[kGroupParts][kBlock / 64];
    __shared__ uint32_t running[kGroupParts];                     // keys of this chunk so far, per owner
    __shared__ uint32_t dst_of[kBlock];
    __shared__ uint32_t my_chunk;
    uint64_t* lds64 = reinterpret_cast<uint64_t*>(lds);
    const uint32_t R = blockDim.x;
    const uint64_t n_tiles = (n + R - 1) / R;
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    for (;;) {
        if (t == 0) my_chunk = atomicAdd(next_chunk, 1u);
        if (t < n_parts) running[t] = 0;
        __syncthreads();
        const uint32_t chunk = my_chunk;
        if (chunk >= used_chunks) break;                          // the same for every lane of the workgroup
        const uint64_t tile_lo = uint64_t(chunk) * chunk_tiles, tile_hi = tile_lo + chunk_tiles < n_tiles ? tile_lo + chunk_tiles : n_tiles;
        for (uint64_t tile = tile_lo; tile < tile_hi; ++tile) {
            const uint64_t r0 = tile * R;
            const uint32_t nr = uint32_t(n - r0 < R ? n - r0 : R);
            const uint8_t* g0 = s0.bases + r0 * uint64_t(s0.ustride);
            const uint32_t bytes = (nr - 1u) * s0.ustride + s0.ulen;
            const uint32_t in0 = uint32_t(reinterpret_cast<uintptr_t>(g0) & 15u);
            stage_chunks(g0 - in0, lds, (in0 + bytes + 15u) >> 4, R);
            __syncthreads();
            const uint64_t i = r0 + t;
            const uint32_t l0 = s0.ulen;
            uint32_t owner = 0xFFFFFFFFu;
            uint64_t my_hash = 0;
            if (t < nr) {
                uint64_t h = hash_begin(l0, 0);
                const uint32_t b0 = in0 + t * s0.ustride;
                uint64_t* out = lds64 + ((b0 + 4u + 7u) >> 3);     // the key is parked over the read's own consumed bytes (encode_staged, LDS_OUT)
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                const uint32_t diff = pack_mate(lds + (b0 >> 2), b0 & 3u, l0, sink);
                h = hash_end(h) & hash_and;
                my_hash = h;
                owner = owner_of(h, n_parts);
                if (diff) {
                    const uint64_t e = locate_bad_base(s0.bases + i * uint64_t(s0.ustride), l0, nullptr, 0, i);
                    if (e != kNoError) atomicMin(reinterpret_cast<unsigned long long*>(err), static_cast<unsigned long long>(e));
                }
            }
            uint32_t rank_in_wave = 0;                            // among the tile's reads of the same owner, in input order
            for (uint32_t p = 0; p < n_parts; ++p) {
                const unsigned long long m = __ballot(owner == p);
                if (owner == p) rank_in_wave = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
                if (lane == 0) wave_cnt[p][wave] = uint32_t(__popcll(m));
            }
            __syncthreads();
            if (t < nr) {
                uint32_t local = running[owner] + rank_in_wave;
                for (uint32_t w = 0; w < wave; ++w) local += wave_cnt[owner][w];
                const uint32_t slot = local < sub_cap ? uint32_t((uint64_t(owner) * n_chunks + chunk) * sub_cap + local) : 0xFFFFFFFFu;
                dst_of[t] = slot;
                if (slot != 0xFFFFFFFFu) { origin[slot] = uint32_t(i); if (out_hash) out_hash[slot] = my_hash; }
            }
            __syncthreads();
            if (t < n_parts) { uint32_t c = 0; for (uint32_t w = 0; w < R / 64u; ++w) c += wave_cnt[t][w]; running[t] += c; }
            {   // every wave streams its rows to their slots: eight lanes per 64-byte key
                const uint32_t wave_reads = (nr > wave * 64u) ? ((nr - wave * 64u < 64u) ? nr - wave * 64u : 64u) : 0u;
                const uint32_t total = wave_reads * W0;
                for (uint32_t x = lane; x < total; x += 64u) {
                    const uint32_t rr = __umulhi(x, rw_magic), kk = x - rr * W0;
                    const uint32_t slot = dst_of[wave * 64u + rr];
                    if (slot != 0xFFFFFFFFu)
                        __builtin_nontemporal_store(lds64[((in0 + (wave * 64u + rr) * s0.ustride + 4u + 7u) >> 3) + kk], &out_keys[uint64_t(slot) * W0 + kk]);
                }
            }
            __syncthreads();
        }
        if (t < n_parts) { counts[uint64_t(t) * n_chunks + chunk] = running[t]; if (running[t]) atomicAdd(&totals[t], static_cast<unsigned long long>(running[t])); }
        __syncthreads();
    }
}

// The paired form: one lane per MATE as in encode_staged_pe_kernel (lane 2q = mate 1 of pair q, lane 2q+1 = mate 2), 128
// pairs per 256-lane tile; the pair's owner comes from the pair's hash (even lane), ranks count pairs.
__global__ __launch_bounds__(kBlock)
void encode_chunks_pe_kernel(SegView s0, SegView s1, uint64_t n, uint32_t W_0, uint32_t W_1, uint32_t n_parts, uint32_t chunk_tiles, uint32_t n_chunks,
                             uint32_t used_chunks, uint64_t sub_cap, uint64_t* __restrict__ out_keys, uint32_t* __restrict__ origin, uint64_t* __restrict__ counts,
                             unsigned long long* __restrict__ totals, uint32_t* __restrict__ next_chunk,
                             uint64_t* __restrict__ err, uint32_t tile_bytes0, uint32_t rw_magic, uint64_t hash_and,
                             uint64_t* __restrict__ out_hash)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t wave_cnt[kGroupParts][kBlock / 64];
    __shared__ uint32_t running[kGroupParts];
    __shared__ uint32_t dst_of[kBlock / 2];
    __shared__ uint32_t my_chunk;
    uint64_t* lds64 = reinterpret_cast<uint64_t*>(lds);
    const uint32_t R = blockDim.x, P = R >> 1, row_words = W_0 + W_1;
    const uint64_t n_tiles = (n + P - 1) / P;
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u, pair = t >> 1, mate = t & 1u;
    for (;;) {
        if (t == 0) my_chunk = atomicAdd(next_chunk, 1u);
        if (t < n_parts) running[t] = 0;
        __syncthreads();
        const uint32_t chunk = my_chunk;
        if (chunk >= used_chunks) break;
        const uint64_t tile_lo = uint64_t(chunk) * chunk_tiles, tile_hi = tile_lo + chunk_tiles < n_tiles ? tile_lo + chunk_tiles : n_tiles;
        for (uint64_t tile = tile_lo; tile < tile_hi; ++tile) {
            const uint64_t r0 = tile * P;
            const uint32_t np = uint32_t(n - r0 < P ? n - r0 : P);
            uint32_t in_base[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const SegView& sv = s ? s1 : s0;
                const uint8_t* g0 = sv.bases + r0 * uint64_t(sv.ustride);
                const uint32_t bytes = (np - 1u) * sv.ustride + sv.ulen;
                const uint32_t head = uint32_t(reinterpret_cast<uintptr_t>(g0) & 15u);
                const uint32_t off = s ? tile_bytes0 : 0u;
                in_base[s] = off + head;
                stage_chunks(g0 - head, lds + (off >> 2), (head + bytes + 15u) >> 4, R);
            }
            __syncthreads();
            const bool live = pair < np;
            const uint64_t i = r0 + pair;
            const uint32_t len = mate ? s1.ulen : s0.ulen;
            const uint32_t stride = mate ? s1.ustride : s0.ustride;
            const uint32_t b = in_base[mate] + pair * stride;
            uint64_t h = hash_begin(len, 0);
            uint32_t diff = 0;
            if (live) {
                uint64_t* out = lds64 + ((b + 4u + 7u) >> 3);
                auto sink = [&](uint64_t w) { *out++ = w; h = hash_word(h, w); };
                diff = pack_mate(lds + (b >> 2), b & 3u, len, sink);
            }
            const uint64_t other = __shfl_xor(h, 1, 64);
            uint32_t owner = 0xFFFFFFFFu;
            const uint64_t pair_hash = hash_pair(h, other) & hash_and;
            if (live && !mate) owner = owner_of(pair_hash, n_parts);
            owner = __shfl(owner, int(lane & ~1u), 64);              // both mates know their pair's owner
            if (live && diff) {
                uint32_t byte = 0;
                const uint8_t* gp = (mate ? s1.bases : s0.bases) + i * uint64_t(stride);
                const uint32_t pos = first_bad_base(gp, len, &byte);
                if (pos != 0xFFFFFFFFu)
                    atomicMin(reinterpret_cast<unsigned long long*>(err), static_cast<unsigned long long>(make_error(i, mate, pos, byte)));
            }
            uint32_t rank_in_wave = 0;
            for (uint32_t p = 0; p < n_parts; ++p) {
                const unsigned long long m = __ballot(owner == p && mate == 0u);
                if (owner == p) rank_in_wave = uint32_t(__popcll(m & ((1ull << (lane & ~1u)) - 1ull)));
                if (lane == 0) wave_cnt[p][wave] = uint32_t(__popcll(m));
            }
            __syncthreads();
            if (live && !mate) {
                uint32_t local = running[owner] + rank_in_wave;
                for (uint32_t w = 0; w < wave; ++w) local += wave_cnt[owner][w];
                const uint32_t slot = local < sub_cap ? uint32_t((uint64_t(owner) * n_chunks + chunk) * sub_cap + local) : 0xFFFFFFFFu;
                dst_of[pair] = slot;
                if (slot != 0xFFFFFFFFu) { origin[slot] = uint32_t(i); if (out_hash) out_hash[slot] = pair_hash; }
            }
            __syncthreads();
            if (t < n_parts) { uint32_t c = 0; for (uint32_t w = 0; w < R / 64u; ++w) c += wave_cnt[t][w]; running[t] += c; }
            {
                const uint32_t wave_pairs = (np > wave * 32u) ? ((np - wave * 32u < 32u) ? np - wave * 32u : 32u) : 0u;
                const uint32_t total = wave_pairs * row_words;
                for (uint32_t x = lane; x < total; x += 64u) {
                    const uint32_t rr = __umulhi(x, rw_magic), kk = x - rr * row_words;
                    const uint32_t m = kk >= W_0 ? 1u : 0u, kw = m ? kk - W_0 : kk;
                    const uint32_t slot = dst_of[wave * 32u + rr];
                    const uint32_t pb = in_base[m] + (wave * 32u + rr) * (m ? s1.ustride : s0.ustride);
                    if (slot != 0xFFFFFFFFu)
                        __builtin_nontemporal_store(lds64[((pb + 4u + 7u) >> 3) + kw], &out_keys[uint64_t(slot) * row_words + kk]);
                }
            }
            __syncthreads();
        }
        if (t < n_parts) { counts[uint64_t(t) * n_chunks + chunk] = running[t]; if (running[t]) atomicAdd(&totals[t], static_cast<unsigned long long>(running[t])); }
        __syncthreads();
    }
}

// The three-step grouping fills an owner's slab from its first slot on (fqd_partition_slabs): seen as sub-slabs that is
// full ones, a partial one, empty ones — and what a slab of n_chunks * sub_cap slots has no room for shows as a count
// above sub_cap in the LAST sub-slab.  counts[p * n_chunks + c] from totals[p].
__global__ void classic_chunk_counts_kernel(const uint64_t* __restrict__ totals, uint32_t n_parts, uint32_t n_chunks, uint64_t sub_cap,
                                            uint64_t* __restrict__ counts)
{
    const uint64_t k = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
    if (k >= uint64_t(n_parts) * n_chunks) return;
    const uint32_t p = uint32_t(k / n_chunks), c = uint32_t(k - uint64_t(p) * n_chunks);
    const uint64_t tot = totals[p], before = uint64_t(c) * sub_cap;
    uint64_t v = tot > before ? tot - before : 0;
    if (c + 1 < n_chunks && v > sub_cap) v = sub_cap;
    counts[k] = v;
}

// counts[p] = number of records of part p, read off the scanned 2-D starts.
__global__ void part_totals_kernel(const uint64_t* __restrict__ starts2d, uint32_t n_parts, uint32_t n_blocks,
                                   uint64_t n, uint64_t* __restrict__ counts)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parts) return;
    const uint64_t lo = starts2d[uint64_t(p) * n_blocks];
    const uint64_t hi = (p + 1 < n_parts) ? starts2d[uint64_t(p + 1) * n_blocks] : n;
    counts[p] = hi - lo;
}

// Key slots of fqd_encode_padded: record i = [hash][len0 | len1 << 32][key words, zeros up to the widest key]: the
// encoders write it as a ragged slot at koff[i] (header first); a read longer than the slots have room for is counted.
__global__ __launch_bounds__(kBlock)
void padded_slots_kernel(SegView s0, SegView s1, uint32_t paired, uint64_t n, uint32_t max0, uint32_t max1, uint32_t stride,
                         uint64_t* __restrict__ koff, unsigned long long* __restrict__ too_long)
{
    uint32_t bad = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        koff[i] = i * uint64_t(stride) + 1u;
        const uint32_t l0 = s0.lengths ? s0.lengths[i] : s0.ulen, l1 = paired ? (s1.lengths ? s1.lengths[i] : s1.ulen) : 0u;
        if (l0 > max0 || l1 > max1) ++bad;
    }
    if (bad) atomicAdd(too_long, static_cast<unsigned long long>(bad));
}

// Hashes that arrived WITH their keys (fqd_insert_slabs_hashed): only the slots of a slab that hold no key need a word,
// the one every insert path passes over.
__global__ __launch_bounds__(kBlock)
void mask_unused_hashes_kernel(uint64_t* __restrict__ hashes, uint64_t n, uint64_t slab_cap, const uint64_t* __restrict__ slab_count)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const uint64_t slab = i / slab_cap;
        if (i - slab * slab_cap >= slab_count[slab]) hashes[i] = kSkipHash;
    }
}

// Placement hashes of uniform keys that arrived without them (the sharded exchange sends keys only):
// the same chains the encoders run, over the stored key words.
// slab_cap != 0: the keys lie in slabs of slab_cap slots of which only the first slab_count[slab] hold a key (the
// sharded exchange's fixed-size messages); the other positions get kSkipHash.
__global__ __launch_bounds__(kBlock)
void hash_keys_kernel(const uint64_t* __restrict__ keys, uint32_t stride, uint64_t n, uint32_t len0, uint32_t len1,
                      uint32_t paired, uint64_t hash_and, uint64_t* __restrict__ hashes,
                      uint64_t slab_cap, const uint64_t* __restrict__ slab_count)
{
    // len1 == kOpaqueKeys: the keys are len0 words of which nothing is known but that equal keys are equal words
    // (padded keys of reads of several lengths, header word first): one chain over all of them
    const bool opaque = len1 == kOpaqueKeys;
    if (opaque) paired = 0;
    const uint32_t w0 = opaque ? len0 : seg_words(len0), W = opaque ? len0 : w0 + (paired ? seg_words(len1) : 0u);
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        if (slab_cap) {
            const uint64_t slab = i / slab_cap;
            if (i - slab * slab_cap >= slab_count[slab]) { hashes[i] = kSkipHash; continue; }
        }
        const uint64_t* __restrict__ p = keys + i * uint64_t(stride);
        uint64_t h = hash_begin(len0, 0), h1 = hash_begin(len1, 0);
        for (uint32_t k0 = 0; k0 < W; k0 += 8u) {             // eight words requested at a time, then chained
            uint64_t w[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) w[j] = p[k0 + j < W ? k0 + j : W - 1u];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) {
                const uint32_t k = k0 + j;
                if (k < w0) h = hash_word(h, w[j]);
                else if (k < W) h1 = hash_word(h1, w[j]);
            }
        }
        h = paired ? hash_pair(h, h1) : hash_end(h);
        hashes[i] = h & hash_and;
    }
}

__global__ __launch_bounds__(kBlock)
void scatter_flags_kernel(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ origin, uint64_t n,
                          uint8_t* __restrict__ keep_out)
{
    for (uint64_t k = blockIdx.x * uint64_t(kBlock) + threadIdx.x; k < n; k += uint64_t(gridDim.x) * kBlock) {
        const uint32_t to = origin[k];
        if (to != 0xFFFFFFFFu) keep_out[to] = flags[k];        // 0xFFFFFFFF: a slab slot no record went to
    }
}

// ---------------------------------------------------------------------------
// Synthetic workload.  Counter-based: every value is a pure function of
// (seed, global index), so ranks generate their slices independently and the expected
// keep flags are known in closed form.
__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline bool syn_is_copy(uint64_t seed, uint64_t g, uint32_t permille)
{ return g > 0 && (mix64(seed ^ (g * 4u + 1u)) % 1000u) < permille; }
__host__ __device__ inline uint64_t syn_parent(uint64_t seed, uint64_t g) { return mix64(seed ^ (g * 4u + 2u)) % g; }
__host__ __device__ inline bool syn_same_mate2(uint64_t seed, uint64_t g) { return (mix64(seed ^ (g * 4u + 3u)) & 1u) != 0; }

__global__ __launch_bounds__(kBlock)
void synth_kernel(uint64_t seed, uint64_t first, uint64_t n, uint32_t len, uint32_t permille, int mate,
                  uint8_t* __restrict__ bases, uint8_t* __restrict__ expect_keep)
{
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        const uint64_t g = first + i;
        uint64_t r = g;
        if (mate == 0) { while (syn_is_copy(seed, r, permille)) r = syn_parent(seed, r); }
        else           { while (syn_is_copy(seed, r, permille) && syn_same_mate2(seed, r)) r = syn_parent(seed, r); }
        if (expect_keep) {
            const bool dup = syn_is_copy(seed, g, permille) && (mate == 0 || syn_same_mate2(seed, g));
            expect_keep[i] = dup ? 0 : 1;
        }
        const uint64_t stream = mix64(seed ^ mix64(r * 2u + uint64_t(mate)));
        uint8_t* out = bases + i * uint64_t(len);
        const uint64_t nv = mix64(stream ^ 0x5bd1e995u);
        const uint32_t n_at = ((nv & 7u) == 0u && len) ? uint32_t((nv >> 8) % len) : 0xFFFFFFFFu;
        for (uint32_t b = 0; b < len; b += 32) {
            uint64_t bits = mix64(stream + (b >> 5) * 0x9E3779B97F4A7C15ull);
            const uint32_t m = (len - b < 32u) ? len - b : 32u;
            for (uint32_t k = 0; k < m; ++k) {
                out[b + k] = (b + k == n_at) ? uint8_t('N') : uint8_t("ACGT"[bits & 3u]);
                bits >>= 2;
            }
        }
    }
}

} // namespace fqd
