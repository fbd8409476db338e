// fqd_bgzf.hip — `.gz` output made where the survivors already are: BGZF members deflated on the GPU
// (same library as fqd_engine.hip; the scheme is described in fqd_bgzf_core.hpp).
//
//   bgzf_count_kernel   member -> LDS, line starts, every chunk parsed, token histogram (LDS, then global)
//   host                two length-limited Huffman codes + the block header from the histogram
//   bgzf_emit_kernel    member -> LDS, line starts, chunk sizes under the codes, block scan, chunks
//                       re-parsed into bits at their offsets of the member's slot (or the member stored),
//                       CRC-32 by chunk and a pairwise combine, BGZF header and trailer
//   bgzf_offsets_kernel running sum of the member sizes
//   bgzf_compact_kernel slots -> members back to back
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "../../include/fqdupaway.h"
#include "fqd_bgzf_core.hpp"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out);

namespace {

using namespace fqd::bgzf;

#define BGZF_TRY(e, expr)                                                                   \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess) { (void)hipGetLastError();       \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } } while (0)

struct DeviceOr {
    __device__ void operator()(uint32_t* p, uint32_t v) const { if (v) atomicOr(p, v); }
};

// Exclusive scan of one value per thread over the workgroup (kThreads = 8 waves); total to all.
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* wave_sums /* LDS, 8 + 1 */, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
    __syncthreads();                                    // wave_sums may still be read from a previous scan
    if (lane == 63u) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < kThreads / 64u; ++w) { const uint32_t s = wave_sums[w]; all += s; if (w < wave) before += s; }
    total = all;
    return before + inc - v;
}

struct MemberLds {
    alignas(16) uint8_t data[kSkewedBytes + 16];   // skewed: see fqd_bgzf_core.hpp
    uint16_t ls[kMaxLines + 2];
    uint32_t line_at[kThreads];                         // newlines before the thread's chunk
    uint32_t wave_sums[16];
    uint32_t n_lines;
};

// Member m of the buffer into LDS; returns its length.
__device__ __forceinline__ uint32_t load_member(const uint8_t* __restrict__ src, uint64_t n, uint64_t m, uint8_t* data)
{
    const uint64_t from = m * kMember;
    const uint32_t L = uint32_t(n - from < kMember ? n - from : kMember);
    const uint8_t* __restrict__ p = src + from;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const uint32_t whole = L / 16u;
        for (uint32_t i = threadIdx.x; i < whole; i += kThreads) {
            const uint4 v = reinterpret_cast<const uint4*>(p)[i];
            uint32_t* d = reinterpret_cast<uint32_t*>(data + Skewed::at(16u * i));      // 16 bytes never straddle a chunk
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        for (uint32_t i = whole * 16u + threadIdx.x; i < L; i += kThreads) data[Skewed::at(i)] = p[i];
    } else {
        for (uint32_t i = threadIdx.x; i < L; i += kThreads) data[Skewed::at(i)] = p[i];
    }
    return L;
}

// Line starts of the member (ls[0] = 0, ls[j] = byte after the j-th newline) from the threads' newline masks and,
// per thread, the line its chunk starts in.  Returns whether the index is usable (not more lines than it holds).
__device__ __forceinline__ bool index_lines(MemberLds& s, const Scan& sc, uint32_t lo)
{
    const uint32_t mine = uint32_t(__popcll(sc.nl.lo) + __popcll(sc.nl.hi));
    uint32_t total;
    const uint32_t before = block_scan(mine, s.wave_sums, total);
    s.line_at[threadIdx.x] = before;
    const bool on = total <= kMaxLines;
    if (threadIdx.x == 0) { s.ls[0] = 0; s.n_lines = total; }
    if (on) {
        uint32_t k = before + 1u;
        for (uint64_t m = sc.nl.lo; m; m &= m - 1u) s.ls[k++] = uint16_t(lo + uint32_t(__builtin_ctzll(m)) + 1u);
        for (uint64_t m = sc.nl.hi; m; m &= m - 1u) s.ls[k++] = uint16_t(lo + 64u + uint32_t(__builtin_ctzll(m)) + 1u);
    }
    __syncthreads();
    return on;
}

// -------------------------------------------------------------------------------------------
struct TokenCounter {
    uint32_t* hist;                                     // LDS: kLitLen + kDist
    __device__ void literal(uint32_t b) { atomicAdd(&hist[b], 1u); }
    __device__ void match(uint32_t len, uint32_t dist)
    {
        atomicAdd(&hist[length_symbol(len).sym], 1u);
        atomicAdd(&hist[kLitLen + dist_symbol(dist).sym], 1u);
    }
};

__global__ __launch_bounds__(kThreads)
void bgzf_count_kernel(const uint8_t* __restrict__ src, uint64_t n, uint64_t members, uint32_t lines_per_record,
                       unsigned long long* __restrict__ hist_out)
{
    __shared__ MemberLds s;
    __shared__ uint32_t hist[kLitLen + kDist];
    for (uint32_t i = threadIdx.x; i < kLitLen + kDist; i += kThreads) hist[i] = 0;
    __syncthreads();
    const uint64_t every = sample_every(members);
    for (uint64_t m = blockIdx.x * every; m < members; m += gridDim.x * every) {
        const uint32_t L = load_member(src, n, m, s.data);
        __syncthreads();
        uint32_t lo, hi;
        chunk_of(threadIdx.x, L, lo, hi);
        const Skewed data{s.data};
        const Scan sc = scan_chunk(data, lo, hi);
        const bool lines_on = index_lines(s, sc, lo);
        const Columns col = column_masks(data, lo, hi, s.ls, s.line_at[threadIdx.x], s.n_lines, L, lines_on, lines_per_record);
        TokenCounter sink{hist};
        parse_chunk(data, lo, hi, sc, col, sink);
        __syncthreads();                                // before the next member overwrites data
    }
    for (uint32_t i = threadIdx.x; i < kLitLen + kDist; i += kThreads)
        if (hist[i]) atomicAdd(&hist_out[i], static_cast<unsigned long long>(hist[i]));
}

// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads)
void bgzf_emit_kernel(const uint8_t* __restrict__ src, uint64_t n, uint64_t members, uint32_t lines_per_record,
                      const Codes* __restrict__ codes, uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes)
{
    __shared__ MemberLds s;
    __shared__ uint32_t lit[kLitLen], dst[kDist];
    __shared__ uint32_t crc_table[256];
    __shared__ uint32_t crc[kThreads];
    for (uint32_t i = threadIdx.x; i < kLitLen; i += kThreads) lit[i] = codes->lit[i];
    for (uint32_t i = threadIdx.x; i < kDist; i += kThreads) dst[i] = codes->dist[i];
    for (uint32_t i = threadIdx.x; i < 256u; i += kThreads) crc_table[i] = codes->crc_table[i];
    const uint32_t header_bits = codes->header_bits;
    __syncthreads();
    const uint32_t t = threadIdx.x;
    for (uint64_t m = blockIdx.x; m < members; m += gridDim.x) {
        const uint32_t L = load_member(src, n, m, s.data);
        __syncthreads();
        uint32_t lo, hi;
        chunk_of(t, L, lo, hi);
        const Skewed data{s.data};
        const Scan sc = scan_chunk(data, lo, hi);
        const bool lines_on = index_lines(s, sc, lo);
        const Columns col = column_masks(data, lo, hi, s.ls, s.line_at[t], s.n_lines, L, lines_on, lines_per_record);

        BitCounter price{lit, dst};
        parse_chunk(data, lo, hi, sc, col, price);
        uint32_t body_bits;
        const uint32_t before = block_scan(price.bits, s.wave_sums, body_bits);
        const uint32_t total_bits = header_bits + body_bits + (lit[256] >> 16);
        uint32_t clen = (total_bits + 7u) / 8u;
        const bool stored = clen >= L + 5u;
        if (stored) clen = L + 5u;

        uint32_t* out = reinterpret_cast<uint32_t*>(slots + m * uint64_t(kSlot));
        DeviceOr orw;
        if (!stored) {
            BitWriter<DeviceOr> w(out, kHeadBytes * 8u + (t == 0 ? 0u : header_bits + before), orw);
            if (t == 0)
                for (uint32_t at = 0; at < header_bits; at += 32u)
                    w.put(header_bits - at >= 32u ? codes->header[at >> 5] : codes->header[at >> 5] & ((1u << (header_bits - at)) - 1u),
                          header_bits - at >= 32u ? 32u : header_bits - at);
            Emitter<DeviceOr> emit{lit, dst, w};
            parse_chunk(data, lo, hi, sc, col, emit);
            if (t == kThreads - 1u) w.put(lit[256] & 0xFFFFu, lit[256] >> 16);      // end of block
            w.finish();
        } else {
            // BFINAL = 1, BTYPE = 00, pad to the byte, LEN, NLEN, the bytes themselves (RFC 1951 §3.2.4)
            if (t == 0) {
                BitWriter<DeviceOr> w(out, kHeadBytes * 8u, orw);
                w.put(1u, 8); w.put(L, 16); w.put(~L & 0xFFFFu, 16);
                w.finish();
            }
            BitWriter<DeviceOr> w(out, (kHeadBytes + 5u + lo) * 8u, orw);
            for (uint32_t p = lo; p < hi; ++p) w.put(data[p], 8);
            w.finish();
        }

        crc[t] = crc_chunk(crc_table, data, lo, hi);
        __syncthreads();
        for (uint32_t k = 0; k < kLevels; ++k) {
            if ((t & ((2u << k) - 1u)) == 0u) crc[t] = crc_advance(codes->crc_shift[k], crc[t]) ^ crc[t + (1u << k)];
            __syncthreads();
        }
        if (t == 0) {
            const uint32_t total = kHeadBytes + clen + kTailBytes;
            BitWriter<DeviceOr> h(out, 0u, orw);
            h.put(31u | (139u << 8) | (8u << 16) | (4u << 24), 32);          // ID1 ID2 CM FLG = FEXTRA
            h.put(0u, 32);                                                   // MTIME
            h.put(0u | (255u << 8) | (6u << 16), 32);                        // XFL, OS = unknown, XLEN = 6
            h.put(uint32_t('B') | (uint32_t('C') << 8) | (2u << 16), 32);    // SI1 SI2 SLEN = 2
            h.put(total - 1u, 16);                                           // BSIZE
            h.finish();
            BitWriter<DeviceOr> tl(out, (kHeadBytes + clen) * 8u, orw);
            tl.put(crc[0] ^ 0xFFFFFFFFu, 32);
            tl.put(L, 32);
            tl.finish();
            sizes[m] = total;
        }
        __syncthreads();                                // data, crc, ls are reused by the next member
    }
}

// Exclusive running sum of the member sizes by one workgroup; the grand total to *total.
__global__ __launch_bounds__(1024)
void bgzf_offsets_kernel(const uint32_t* __restrict__ sizes, uint64_t members, uint64_t* __restrict__ offsets, uint64_t* __restrict__ total)
{
    __shared__ unsigned long long wave_sums[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < members; base += 1024u) {
        const uint64_t i = base + threadIdx.x;
        const unsigned long long v = i < members ? sizes[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
        if (lane == 63u) wave_sums[wave] = inc;
        __syncthreads();
        unsigned long long before = carry, all = 0;
        for (uint32_t w = 0; w < 16u; ++w) { const unsigned long long sw = wave_sums[w]; all += sw; if (w < wave) before += sw; }
        if (i < members) offsets[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256)
void bgzf_compact_kernel(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes, const uint64_t* __restrict__ offsets,
                         uint64_t members, uint8_t* __restrict__ dst)
{
    for (uint64_t m = blockIdx.x; m < members; m += gridDim.x) {
        const uint8_t* __restrict__ from = slots + m * uint64_t(kSlot);
        uint8_t* __restrict__ to = dst + offsets[m];
        const uint32_t size = sizes[m];
        // the slot is aligned; the destination is wherever the sum put it: words in, bytes out
        const uint32_t words = size / 4u;
        for (uint32_t i = threadIdx.x; i < words; i += 256u) {
            const uint32_t v = reinterpret_cast<const uint32_t*>(from)[i];
            uint8_t* q = to + 4u * i;
            if ((reinterpret_cast<uintptr_t>(q) & 3u) == 0) *reinterpret_cast<uint32_t*>(q) = v;
            else { q[0] = uint8_t(v); q[1] = uint8_t(v >> 8); q[2] = uint8_t(v >> 16); q[3] = uint8_t(v >> 24); }
        }
        for (uint32_t i = words * 4u + threadIdx.x; i < size; i += 256u) to[i] = from[i];
    }
}

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

extern "C" {

uint64_t fqd_bgzf_bound(uint64_t n)
{
    const uint64_t members = (n + kMember - 1) / kMember;
    return n + members * uint64_t(kHeadBytes + 5u + kTailBytes);
}

int fqd_bgzf_deflate(fqd_engine* e, const uint8_t* src, uint64_t n, uint32_t lines_per_record,
                     uint8_t* dst, uint64_t dst_capacity, uint64_t* out_bytes)
{
    if (!e) return FQD_ERR_ARG;
    if (!out_bytes || (n && (!src || !dst)) || lines_per_record == 0 || lines_per_record > 64)
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_bgzf_deflate: bad arguments");
    *out_bytes = 0;
    if (n == 0) return FQD_OK;
    if (dst_capacity < fqd_bgzf_bound(n))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_bgzf_deflate: dst_capacity below fqd_bgzf_bound(n)");
    BGZF_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint64_t members = (n + kMember - 1) / kMember;

    const size_t hist_bytes = round_up((kLitLen + kDist) * sizeof(uint64_t), 256);
    const size_t codes_bytes = round_up(sizeof(Codes), 256);
    const uint64_t roomy = (members + 255) / 256 * 256;        // the scratch is sized in steps: calls of about the same size do not regrow it
    const size_t sizes_bytes = round_up(roomy * sizeof(uint32_t), 256);
    const size_t offs_bytes = round_up((roomy + 1) * sizeof(uint64_t), 256);
    const size_t slots_bytes = members * size_t(kSlot);
    void* base = nullptr;
    const int rc = fqd_internal_scratch(e, 1, hist_bytes + codes_bytes + sizes_bytes + offs_bytes + roomy * size_t(kSlot), &base);
    if (rc != FQD_OK) return rc;
    uint8_t* at = static_cast<uint8_t*>(base);
    unsigned long long* d_hist = reinterpret_cast<unsigned long long*>(at); at += hist_bytes;
    Codes* d_codes = reinterpret_cast<Codes*>(at); at += codes_bytes;
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(at); at += sizes_bytes;
    uint64_t* d_offs = reinterpret_cast<uint64_t*>(at); at += offs_bytes;
    uint8_t* d_slots = at;

    const uint32_t grid = uint32_t(std::min<uint64_t>(members, 2048));
    BGZF_TRY(e, hipMemsetAsync(d_hist, 0, hist_bytes, stream));
    BGZF_TRY(e, hipMemsetAsync(d_slots, 0, slots_bytes, stream));
    hipLaunchKernelGGL(bgzf_count_kernel, dim3(grid), dim3(kThreads), 0, stream, src, n, members, lines_per_record, d_hist);
    BGZF_TRY(e, hipGetLastError());
    uint64_t hist[kLitLen + kDist];
    BGZF_TRY(e, hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, stream));
    BGZF_TRY(e, hipStreamSynchronize(stream));
    static thread_local Codes codes;
    build_codes(hist, members, codes);
    BGZF_TRY(e, hipMemcpyAsync(d_codes, &codes, sizeof codes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bgzf_emit_kernel, dim3(grid), dim3(kThreads), 0, stream, src, n, members, lines_per_record,
                       static_cast<const Codes*>(d_codes), d_slots, d_sizes);
    BGZF_TRY(e, hipGetLastError());
    hipLaunchKernelGGL(bgzf_offsets_kernel, dim3(1), dim3(1024), 0, stream, static_cast<const uint32_t*>(d_sizes), members, d_offs, d_offs + members);
    BGZF_TRY(e, hipGetLastError());
    hipLaunchKernelGGL(bgzf_compact_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const uint8_t*>(d_slots),
                       static_cast<const uint32_t*>(d_sizes), static_cast<const uint64_t*>(d_offs), members, dst);
    BGZF_TRY(e, hipGetLastError());
    uint64_t total = 0;
    BGZF_TRY(e, hipMemcpyAsync(&total, d_offs + members, sizeof total, hipMemcpyDeviceToHost, stream));
    BGZF_TRY(e, hipStreamSynchronize(stream));
    *out_bytes = total;
    return FQD_OK;
}

} // extern "C"
