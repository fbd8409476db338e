// fqd_inflate.hip — BGZF inputs inflated and cut into records on the GPU (same library as fqd_engine.hip).
//
//   bgzf_inflate_kernel   one WAVE per member (fqd_inflate_wave.hpp): the lanes decode a block's bits side by side from
//                         guessed code boundaries, fall into step, count, decode again to write; tables in LDS (10 KB)
//   bgzf_check_crc_kernel one workgroup per member: CRC-32 of what came out against the member's trailer
//                         (chunk registers + pairwise combine, as the writer: fqd_bgzf_core.hpp)
//   count_newlines / newline_positions / records kernels: the inflated text cut into FASTQ/FASTA records
//                         by the rule of the host scanner (host/records.cpp scan_core; fastqview.cpp:92-138):
//                         a record is `lines_per_record` lines, starts with '@' ('>'), and in FASTQ its
//                         sequence and quality lines have the same length.  Anything else — and a text
//                         that does not end a record with its last byte — is only REPORTED here: the caller
//                         then reads the file the host way, which reproduces the reference's diagnostics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "../../include/fqdupaway.h"
#include "fqd_bgzf_core.hpp"
#include "fqd_inflate_wave.hpp"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out);

namespace {

#define INF_TRY(e, expr)                                                                    \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess) { (void)hipGetLastError();       \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } } while (0)

constexpr uint32_t kWave = 64;

// The wave as fqd_inflate_wave.hpp sees it: a workgroup IS one wave, so the workgroup barrier costs nothing and makes
// the LDS and global writes of a phase visible to the next.
struct WaveCtx {
    static constexpr uint32_t kLanes = kWave;
    uint32_t lane;
    template <class F> __device__ __forceinline__ void lanes(F f) { f(lane); __syncthreads(); }
    template <class F> __device__ __forceinline__ void lanes_open(F f) { f(lane); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    template <class F> __device__ __forceinline__ uint64_t ballot(F f) { return __ballot(f(lane) ? 1 : 0); }
    __device__ __forceinline__ uint32_t same(uint32_t v) const { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    __device__ __forceinline__ void add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
#ifdef FQD_STAMPS
    unsigned long long last = 0, spent[8] = {};
    __device__ __forceinline__ void mark(int k) { const unsigned long long now = __builtin_readcyclecounter(); spent[k] += now - last; last = now; }
#else
    __device__ __forceinline__ void mark(int) {}
#endif
};
#ifdef FQD_STAMPS
__device__ unsigned long long g_inflate_cycles[8];
#endif

constexpr uint32_t kInflateWavesPerCu = 16;            // 10 KB of LDS each; four per SIMD need <= 128 VGPRs

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 4)))
void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off, const uint32_t* __restrict__ comp_len,
                         const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ out_len, uint64_t members,
                         uint8_t* __restrict__ text, fqd::winf::Token* __restrict__ tokens, unsigned long long* __restrict__ n_bad)
{
    __shared__ fqd::winf::Shared<kWave> sh;
    WaveCtx ctx{threadIdx.x};
#ifdef FQD_STAMPS
    ctx.last = __builtin_readcyclecounter();
#endif
    fqd::winf::Token* tok = tokens + size_t(blockIdx.x) * fqd::winf::kTokenRoom;
    uint32_t bad = 0;
    for (uint64_t m = blockIdx.x; m < members; m += gridDim.x) {
        const uint32_t len = out_len[m];
        if (len > 65536u) { ++bad; continue; }                // no BGZF member holds more (the CLI's header walk says so too; an ABI caller may not)
        const uint32_t st = fqd::winf::inflate_member(ctx, sh, comp + comp_off[m], comp_len[m], text + out_off[m], len, tok);
        bad += st != fqd::winf::kOk ? 1u : 0u;
    }
    if (threadIdx.x == 0 && bad) atomicAdd(n_bad, static_cast<unsigned long long>(bad));
#ifdef FQD_STAMPS
    if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_inflate_cycles[k], ctx.spent[k]);
#endif
}

// Byte table and "advance by 128 * 2^k zero bytes" matrices of CRC-32, worked out by the compiler: they live in the
// code object, so that queueing a batch (fqd_bgzf_inflate_async) never copies from pageable host memory — such a
// copy makes the caller wait for everything queued on the stream before it, i.e. for the batch before.
struct CrcTables { uint32_t table[256]; uint32_t shift[fqd::bgzf::kLevels][32]; };
constexpr CrcTables make_crc_tables()
{
    CrcTables c{};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (r & 1u ? 0xEDB88320u : 0u);
        c.table[i] = r;
    }
    for (uint32_t b = 0; b < 32; ++b) {
        uint32_t reg = 1u << b;
        for (uint32_t k = 0; k < fqd::bgzf::kChunk; ++k) reg = c.table[reg & 0xFFu] ^ (reg >> 8);
        c.shift[0][b] = reg;
    }
    for (uint32_t k = 1; k < fqd::bgzf::kLevels; ++k)
        for (uint32_t b = 0; b < 32; ++b) {
            uint32_t out = 0;                                 // the matrix applied to its own column: twice as far
            for (uint32_t j = 0; j < 32u; ++j) out ^= (c.shift[k - 1][b] >> j) & 1u ? c.shift[k - 1][j] : 0u;
            c.shift[k][b] = out;
        }
    return c;
}
__device__ const CrcTables g_crc_tables = make_crc_tables();

__global__ __launch_bounds__(fqd::bgzf::kThreads)
void bgzf_check_crc_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ out_len,
                           const uint32_t* __restrict__ want, uint64_t members, unsigned long long* __restrict__ n_bad)
{
    using namespace fqd::bgzf;
    __shared__ alignas(16) uint8_t data[kThreads * kChunk + 16];
    __shared__ uint32_t table[256];
    __shared__ uint32_t crc[kThreads];
    const uint32_t t = threadIdx.x;
    const CrcTables* __restrict__ tabs = &g_crc_tables;
    for (uint32_t i = t; i < 256u; i += kThreads) table[i] = tabs->table[i];
    __syncthreads();
    for (uint64_t m = blockIdx.x; m < members; m += gridDim.x) {
        const uint32_t L = out_len[m];
        if (L > 65536u) continue;                             // counted as bad by the inflater; more would not fit `data`
        const uint8_t* __restrict__ p = text + out_off[m];
        // aligned 16-byte loads from the first aligned address on; the ragged head byte by byte
        const uint32_t head = uint32_t((16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u);
        const uint32_t h = head < L ? head : L;
        if (t < h) data[t] = p[t];
        const uint32_t whole = (L - h) / 16u;
        for (uint32_t i = t; i < whole; i += kThreads) {
            const uint4 v = reinterpret_cast<const uint4*>(p + h)[i];
            uint8_t* d = data + h + 16u * i;                          // LDS side unaligned by h: bytes out of the registers
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = uint8_t(w[k >> 2] >> (8 * (k & 3)));
        }
        for (uint32_t i = h + whole * 16u + t; i < L; i += kThreads) data[i] = p[i];
        __syncthreads();
        uint32_t lo, hi;
        chunk_of(t, L, lo, hi);
        crc[t] = crc_chunk(table, Aligned{data}, lo, hi);
        __syncthreads();
        for (uint32_t k = 0; k < kLevels; ++k) {
            if ((t & ((2u << k) - 1u)) == 0u) crc[t] = crc_advance(tabs->shift[k], crc[t]) ^ crc[t + (1u << k)];
            __syncthreads();
        }
        if (t == 0 && L != 0 && (crc[0] ^ 0xFFFFFFFFu) != want[m]) atomicAdd(n_bad, 1ull);
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------
constexpr uint32_t kScanThreads = 256, kScanPer = 32, kScanTile = kScanThreads * kScanPer;     // 8 KiB of text per tile

__device__ __forceinline__ uint32_t newline_mask(const uint8_t* __restrict__ text, uint64_t n, uint64_t at)
{
    // bit k set iff text[at + k] == '\n', k < 32 (bytes beyond n do not count)
    uint32_t m = 0;
    if (at + kScanPer <= n && (reinterpret_cast<uintptr_t>(text + at) & 15u) == 0) {
        const uint4 a = reinterpret_cast<const uint4*>(text + at)[0], b = reinterpret_cast<const uint4*>(text + at)[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 32; ++k) m |= (((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) == 10u ? 1u : 0u) << k;
    } else {
        for (uint32_t k = 0; k < kScanPer && at + k < n; ++k) m |= (text[at + k] == uint8_t('\n') ? 1u : 0u) << k;
    }
    return m;
}

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* sums /* LDS, 4 */, uint32_t& before)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
    __syncthreads();
    if (lane == 63u) sums[wave] = inc;
    __syncthreads();
    uint32_t all = 0, prev = 0;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / 64u; ++w) { all += sums[w]; if (w < wave) prev += sums[w]; }
    before = prev + inc - v;
    return all;
}

__global__ __launch_bounds__(kScanThreads)
void count_newlines_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t tiles, uint32_t* __restrict__ tile_count)
{
    __shared__ uint32_t sums[4];
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t at = tile * kScanTile + uint64_t(threadIdx.x) * kScanPer;
        const uint32_t mine = at < n ? uint32_t(__popc(newline_mask(text, n, at))) : 0u;
        uint32_t before;
        const uint32_t all = block_sum_256(mine, sums, before);
        if (threadIdx.x == 0) tile_count[tile] = all;
    }
}

// Exclusive running sum of the tile counts, total to offsets[tiles]: every workgroup sums its stretch of the counts,
// one workgroup turns the sums into where each stretch starts, every workgroup scans its stretch from there.  (One
// workgroup walking all four million tiles of a 32 GB text took 6 ms, as long as counting the newlines themselves.)
constexpr uint32_t kOffsetParts = 1024;
__global__ __launch_bounds__(1024)
void tile_sums_kernel(const uint32_t* __restrict__ counts, uint64_t tiles, uint64_t per_part, unsigned long long* __restrict__ parts)
{
    __shared__ unsigned long long wave_sums[16];
    const uint64_t lo = blockIdx.x * per_part, hi = lo + per_part < tiles ? lo + per_part : tiles;
    unsigned long long v = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024u) v += counts[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63u) == 0u) wave_sums[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long all = 0; for (uint32_t w = 0; w < 16u; ++w) all += wave_sums[w]; parts[blockIdx.x] = all; }
}

__global__ __launch_bounds__(1024)
void part_starts_kernel(unsigned long long* __restrict__ parts, uint32_t n_parts, uint64_t tiles, uint64_t* __restrict__ offsets)
{
    __shared__ unsigned long long wave_sums[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long v = threadIdx.x < n_parts ? parts[threadIdx.x] : 0ull;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
    if (lane == 63u) wave_sums[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (uint32_t w = 0; w < 16u; ++w) { const unsigned long long sw = wave_sums[w]; all += sw; if (w < wave) before += sw; }
    if (threadIdx.x < n_parts) parts[threadIdx.x] = before + inc - v;
    if (threadIdx.x == 0) offsets[tiles] = all;
}

__global__ __launch_bounds__(1024)
void tile_offsets_kernel(const uint32_t* __restrict__ counts, uint64_t tiles, uint64_t per_part, const unsigned long long* __restrict__ parts,
                         uint64_t* __restrict__ offsets)
{
    __shared__ unsigned long long wave_sums[16];
    __shared__ unsigned long long carry;
    const uint64_t lo = blockIdx.x * per_part, hi = lo + per_part < tiles ? lo + per_part : tiles;
    if (threadIdx.x == 0) carry = parts[blockIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t base = lo; base < hi; base += 1024u) {
        const uint64_t i = base + threadIdx.x;
        const unsigned long long v = i < hi ? counts[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(inc, d, 64); if (int(lane) >= d) inc += up; }
        if (lane == 63u) wave_sums[wave] = inc;
        __syncthreads();
        unsigned long long before = carry, all = 0;
        for (uint32_t w = 0; w < 16u; ++w) { const unsigned long long sw = wave_sums[w]; all += sw; if (w < wave) before += sw; }
        if (i < hi) offsets[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += all;
        __syncthreads();
    }
}

// counts[0..tiles) -> offsets[0..tiles]; `parts` is room for kOffsetParts numbers
inline void launch_tile_offsets(hipStream_t stream, const uint32_t* counts, uint64_t tiles, uint64_t* offsets, unsigned long long* parts)
{
    const uint64_t per_part = std::max<uint64_t>(1024, (tiles + kOffsetParts - 1) / kOffsetParts);
    const uint32_t n_parts = uint32_t((tiles + per_part - 1) / per_part);
    hipLaunchKernelGGL(tile_sums_kernel, dim3(n_parts), dim3(1024), 0, stream, counts, tiles, per_part, parts);
    hipLaunchKernelGGL(part_starts_kernel, dim3(1), dim3(1024), 0, stream, parts, n_parts, tiles, offsets);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(n_parts), dim3(1024), 0, stream, counts, tiles, per_part, static_cast<const unsigned long long*>(parts), offsets);
}

__global__ __launch_bounds__(kScanThreads)
void newline_positions_kernel(const uint8_t* __restrict__ text, uint64_t n, uint64_t tiles, const uint64_t* __restrict__ tile_offset,
                              uint64_t* __restrict__ nl_pos)
{
    __shared__ uint32_t sums[4];
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t at = tile * kScanTile + uint64_t(threadIdx.x) * kScanPer;
        uint32_t m = at < n ? newline_mask(text, n, at) : 0u;
        uint32_t before;
        block_sum_256(uint32_t(__popc(m)), sums, before);
        uint64_t k = tile_offset[tile] + before;
        while (m) { const uint32_t b = uint32_t(__ffs(int(m))) - 1u; nl_pos[k++] = at + b; m &= m - 1u; }
    }
}

// Record r = lines [r * K, r * K + K).  flags: bit 0 a record does not start with `lead`, bit 1 FASTQ
// sequence/quality lengths differ, bit 2 a record of 4 GiB or more.
__global__ __launch_bounds__(256)
void records_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ nl_pos, uint64_t n_records, uint32_t K, uint8_t lead,
                    uint64_t* __restrict__ start, uint64_t* __restrict__ seq_off, uint32_t* __restrict__ id_len,
                    uint32_t* __restrict__ seq_len, uint32_t* __restrict__ size, uint32_t* __restrict__ flags)
{
    uint32_t bad = 0;
    for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < n_records; r += uint64_t(gridDim.x) * 256ull) {
        const uint64_t s = r ? nl_pos[r * K - 1] + 1u : 0u;
        const uint64_t e0 = nl_pos[r * K], e1 = nl_pos[r * K + 1], last = nl_pos[r * K + K - 1];
        if (text[s] != lead) bad |= 1u;
        if (K == 4u) { const uint64_t e2 = nl_pos[r * K + 2]; if (e1 - e0 != last - e2) bad |= 2u; }
        const uint64_t sz = last - s + 1u;
        if (sz > 0xFFFFFFFFull) bad |= 4u;
        start[r] = s; seq_off[r] = e0 + 1u; id_len[r] = uint32_t(e0 - s + 1u); seq_len[r] = uint32_t(e1 - e0 - 1u); size[r] = uint32_t(sz);
    }
    if (bad) atomicOr(flags, bad);
}

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

extern "C" {

// One batch of members queued on the engine's stream, nothing waited for: bad members are ADDED to the two counters at
// bad_counters (device; the caller zeroes them once and reads them when it likes).
int fqd_bgzf_inflate_async(fqd_engine* e, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len,
                           const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint64_t n_members,
                           uint8_t* text, uint64_t* bad_counters)
{
    if (!e) return FQD_ERR_ARG;
    if (!bad_counters || (n_members && (!comp || !comp_off || !comp_len || !out_off || !out_len || !crc || !text)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_bgzf_inflate_async: bad arguments");
    if (n_members == 0) return FQD_OK;
    INF_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint32_t grid = uint32_t(std::min<uint64_t>(n_members, 256u * kInflateWavesPerCu));
    const size_t lens_bytes = round_up(sizeof(fqd::winf::Token) * fqd::winf::kTokenRoom * grid, 256);
    void* base = nullptr;
    const int rc = fqd_internal_scratch(e, 1, 256 + lens_bytes, &base);     // batches of one stream share it: they run one after the other
    if (rc != FQD_OK) return rc;
    fqd::winf::Token* d_tokens = reinterpret_cast<fqd::winf::Token*>(static_cast<uint8_t*>(base) + 256);
    unsigned long long* d_bad = reinterpret_cast<unsigned long long*>(bad_counters);
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(grid), dim3(kWave), 0, stream, comp, comp_off, comp_len, out_off, out_len, n_members,
                       text, d_tokens, d_bad);
    INF_TRY(e, hipGetLastError());
    hipLaunchKernelGGL(bgzf_check_crc_kernel, dim3(uint32_t(std::min<uint64_t>(n_members, 2048))), dim3(fqd::bgzf::kThreads), 0, stream,
                       static_cast<const uint8_t*>(text), out_off, out_len, crc, n_members, d_bad + 1);
    INF_TRY(e, hipGetLastError());
    return FQD_OK;
}

int fqd_bgzf_inflate(fqd_engine* e, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len,
                     const uint64_t* out_off, const uint32_t* out_len, const uint32_t* crc, uint64_t n_members,
                     uint8_t* text, uint64_t* n_bad)
{
    if (!e) return FQD_ERR_ARG;
    if (!n_bad || (n_members && (!comp || !comp_off || !comp_len || !out_off || !out_len || !crc || !text)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_bgzf_inflate: bad arguments");
    *n_bad = 0;
    if (n_members == 0) return FQD_OK;
    INF_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint32_t grid = uint32_t(std::min<uint64_t>(n_members, 256u * kInflateWavesPerCu));
    const size_t lens_bytes = round_up(sizeof(fqd::winf::Token) * fqd::winf::kTokenRoom * grid, 256);
    void* base = nullptr;
    const int rc = fqd_internal_scratch(e, 1, 256 + lens_bytes, &base);
    if (rc != FQD_OK) return rc;
    unsigned long long* d_bad = static_cast<unsigned long long*>(base);
    fqd::winf::Token* d_tokens = reinterpret_cast<fqd::winf::Token*>(static_cast<uint8_t*>(base) + 256);
    INF_TRY(e, hipMemsetAsync(d_bad, 0, 256, stream));
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(grid), dim3(kWave), 0, stream, comp, comp_off, comp_len, out_off, out_len, n_members,
                       text, d_tokens, d_bad);
    INF_TRY(e, hipGetLastError());
    hipLaunchKernelGGL(bgzf_check_crc_kernel, dim3(uint32_t(std::min<uint64_t>(n_members, 2048))), dim3(fqd::bgzf::kThreads), 0, stream,
                       static_cast<const uint8_t*>(text), out_off, out_len, crc, n_members, d_bad + 1);
    INF_TRY(e, hipGetLastError());
    unsigned long long bad[2] = {0, 0};
    INF_TRY(e, hipMemcpyAsync(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, stream));
    INF_TRY(e, hipStreamSynchronize(stream));
#ifdef FQD_STAMPS
    {
        unsigned long long c[8] = {}, zero[8] = {};
        INF_TRY(e, hipMemcpyFromSymbol(c, HIP_SYMBOL(g_inflate_cycles), sizeof c));
        INF_TRY(e, hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_cycles), zero, sizeof zero));
        unsigned long long all = 0; for (int k = 0; k < 8; ++k) all += c[k];
        const char* name[8] = {"headers", "tables", "count rounds", "prefix", "write pass", "match order", "stored", "match copies"};
        std::fprintf(stderr, "[inflate cycles, summed over waves] total %.3e:", double(all));
        for (int k = 0; k < 8; ++k) std::fprintf(stderr, "  %s %.1f%%", name[k], all ? 100.0 * double(c[k]) / double(all) : 0.0);
        std::fprintf(stderr, "\n");
    }
#endif
    *n_bad = bad[0] + bad[1];
    return FQD_OK;
}

int fqd_count_lines(fqd_engine* e, const uint8_t* text, uint64_t n, uint64_t* n_lines)
{
    if (!e) return FQD_ERR_ARG;
    if (!n_lines || (n && !text)) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_count_lines: bad arguments");
    *n_lines = 0;
    if (n == 0) return FQD_OK;
    INF_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
    void* base = nullptr;
    const size_t counts_bytes = round_up(tiles * sizeof(uint32_t), 256);
    const int rc = fqd_internal_scratch(e, 0, counts_bytes + (tiles + 1 + kOffsetParts) * sizeof(uint64_t), &base);
    if (rc != FQD_OK) return rc;
    uint32_t* counts = static_cast<uint32_t*>(base);
    uint64_t* offs = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(base) + counts_bytes);
    hipLaunchKernelGGL(count_newlines_kernel, dim3(uint32_t(std::min<uint64_t>(tiles, 8192))), dim3(kScanThreads), 0, stream, text, n, tiles, counts);
    INF_TRY(e, hipGetLastError());
    launch_tile_offsets(stream, counts, tiles, offs, reinterpret_cast<unsigned long long*>(offs + tiles + 1));
    INF_TRY(e, hipGetLastError());
    INF_TRY(e, hipMemcpyAsync(n_lines, offs + tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    INF_TRY(e, hipStreamSynchronize(stream));
    return FQD_OK;
}

int fqd_scan_records(fqd_engine* e, const uint8_t* text, uint64_t n, uint32_t lines_per_record, uint64_t n_records,
                     uint64_t* start, uint64_t* seq_off, uint32_t* id_len, uint32_t* seq_len, uint32_t* size, int* well_formed)
{
    if (!e) return FQD_ERR_ARG;
    if (!well_formed || (lines_per_record != 4 && lines_per_record != 2) || (n && !text) ||
        (n_records && (!start || !seq_off || !id_len || !seq_len || !size)))
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_scan_records: bad arguments");
    *well_formed = 0;
    if (n == 0) { *well_formed = n_records == 0; return FQD_OK; }
    INF_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
    // the line count of the text decides how much scratch the newline positions need: counted first
    uint64_t n_lines = 0;
    int rc = fqd_count_lines(e, text, n, &n_lines);
    if (rc != FQD_OK) return rc;
    uint8_t last_byte = 0;
    INF_TRY(e, hipMemcpyAsync(&last_byte, text + n - 1, 1, hipMemcpyDeviceToHost, stream));
    INF_TRY(e, hipStreamSynchronize(stream));
    if (last_byte != uint8_t('\n') || n_lines != n_records * lines_per_record) return FQD_OK;       // not well formed: nothing written
    if (n_records == 0) { *well_formed = 1; return FQD_OK; }
    const size_t counts_bytes = round_up(tiles * sizeof(uint32_t), 256), offs_bytes = round_up((tiles + 1 + kOffsetParts) * sizeof(uint64_t), 256);
    void* base = nullptr;
    rc = fqd_internal_scratch(e, 0, counts_bytes + offs_bytes + 256 + n_lines * sizeof(uint64_t), &base);
    if (rc != FQD_OK) return rc;
    uint32_t* counts = static_cast<uint32_t*>(base);
    uint64_t* offs = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(base) + counts_bytes);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(base) + counts_bytes + offs_bytes);
    uint64_t* nl_pos = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(base) + counts_bytes + offs_bytes + 256);
    // (the scratch may have moved when it grew: the counts are made again, it takes microseconds)
    hipLaunchKernelGGL(count_newlines_kernel, dim3(uint32_t(std::min<uint64_t>(tiles, 8192))), dim3(kScanThreads), 0, stream, text, n, tiles, counts);
    launch_tile_offsets(stream, counts, tiles, offs, reinterpret_cast<unsigned long long*>(offs + tiles + 1));
    INF_TRY(e, hipGetLastError());
    INF_TRY(e, hipMemsetAsync(d_flags, 0, 256, stream));
    hipLaunchKernelGGL(newline_positions_kernel, dim3(uint32_t(std::min<uint64_t>(tiles, 8192))), dim3(kScanThreads), 0, stream, text, n, tiles,
                       static_cast<const uint64_t*>(offs), nl_pos);
    INF_TRY(e, hipGetLastError());
    hipLaunchKernelGGL(records_kernel, dim3(uint32_t(std::min<uint64_t>((n_records + 255) / 256, 8192))), dim3(256), 0, stream, text,
                       static_cast<const uint64_t*>(nl_pos), n_records, lines_per_record, uint8_t(lines_per_record == 4 ? '@' : '>'),
                       start, seq_off, id_len, seq_len, size, d_flags);
    INF_TRY(e, hipGetLastError());
    uint32_t flags = 0;
    INF_TRY(e, hipMemcpyAsync(&flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, stream));
    INF_TRY(e, hipStreamSynchronize(stream));
    *well_formed = flags == 0;
    return FQD_OK;
}

} // extern "C"
