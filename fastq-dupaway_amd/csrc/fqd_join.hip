// fqd_join.hip — the `--unordered` read-ID join on the GPU (same library as fqd_engine.hip).
//
// The reference sorts both FASTQ files by ID tag with an on-disk merge sort
// (ExternalSorter<T>, external_sort.hpp:66-215; order = FastqViewWithId::cmp,
// fastqview.cpp:168-178) and merge-joins the sorted files (hash_dup_remover.hpp:279-340).
// Here the tags live in HBM and are ordered by an LSD radix sort over 8-byte big-endian
// chunks (rocPRIM's device radix sort does the per-chunk key/value passes; chunks in which
// every tag agrees — instrument / run / flow-cell prefixes — are skipped), and the
// equality branch of the merge-join is a binary search of every tag of file A in sorted B.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "../../include/fqdupaway.h"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out);

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;

#define JOIN_TRY(e, expr)                                                                   \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess)                                  \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } while (0)

__global__ __launch_bounds__(kBlock)
void iota_and_maxlen_kernel(uint32_t* __restrict__ perm, const uint32_t* __restrict__ len, uint64_t n, unsigned int* __restrict__ maxlen)
{
    unsigned int m = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlock) {
        perm[i] = uint32_t(i);
        m = max(m, len[i]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_down(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxlen, m);
}

// keys[i] = bytes [8c, 8c+8) of the tag of record perm[i], big-endian, zero padded
// (c < 0: the tag length, the least significant key).  lo/hi accumulate min/max so the host
// can skip a pass in which every key is the same.
__global__ __launch_bounds__(kBlock)
void chunk_keys_kernel(fqd_tags t, const uint32_t* __restrict__ perm, int chunk, uint64_t* __restrict__ keys,
                       unsigned long long* __restrict__ lo_hi)
{
    unsigned long long lo = ~0ull, hi = 0;
    for (uint64_t i = blockIdx.x * uint64_t(kBlock) + threadIdx.x; i < t.n; i += uint64_t(gridDim.x) * kBlock) {
        const uint32_t r = perm[i];
        const uint32_t L = t.lengths[r];
        uint64_t k = 0;
        if (chunk < 0) k = L;
        else {
            const uint32_t from = uint32_t(chunk) * 8u;
            const uint8_t* p = t.bytes + t.offsets[r];
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) k = (k << 8) | (from + b < L ? p[from + b] : 0u);
        }
        keys[i] = k;
        lo = min(lo, (unsigned long long)k); hi = max(hi, (unsigned long long)k);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        lo = min(lo, (unsigned long long)__shfl_down(lo, d, 64));
        hi = max(hi, (unsigned long long)__shfl_down(hi, d, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&lo_hi[0], lo); atomicMax(&lo_hi[1], hi); }
}

// FastqViewWithId::cmp (fastqview.cpp:168-178) on device: bytes over the shorter length, then
// shorter first.  (strncmp would also stop at a NUL byte; tags are text and hold none.)
__device__ __forceinline__ int compare_tags(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen)
{
    const uint32_t m = alen < blen ? alen : blen;
    for (uint32_t k = 0; k < m; ++k) {
        const int d = int(a[k]) - int(b[k]);
        if (d) return d;
    }
    return alen == blen ? 0 : (alen < blen ? -1 : 1);
}

__global__ __launch_bounds__(kBlock)
void match_kernel(fqd_tags a, const uint32_t* __restrict__ perm_a, fqd_tags b, const uint32_t* __restrict__ perm_b,
                  uint32_t* __restrict__ match)
{
    for (uint64_t k = blockIdx.x * uint64_t(kBlock) + threadIdx.x; k < a.n; k += uint64_t(gridDim.x) * kBlock) {
        const uint32_t ra = perm_a[k];
        const uint8_t* ta = a.bytes + a.offsets[ra];
        const uint32_t la = a.lengths[ra];
        uint64_t lo = 0, hi = b.n;
        while (lo < hi) {                                    // lower bound of ta in sorted b
            const uint64_t mid = (lo + hi) >> 1;
            const uint32_t rb = perm_b[mid];
            if (compare_tags(b.bytes + b.offsets[rb], b.lengths[rb], ta, la) < 0) lo = mid + 1; else hi = mid;
        }
        uint32_t out = kNoMatch;
        if (lo < b.n) {
            const uint32_t rb = perm_b[lo];
            if (compare_tags(b.bytes + b.offsets[rb], b.lengths[rb], ta, la) == 0) out = uint32_t(lo);
        }
        match[k] = out;
    }
}

uint32_t grid_for(uint64_t n) { return uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, 2048))); }

} // namespace

extern "C" {

int fqd_sort_tags(fqd_engine* e, const fqd_tags* t, uint32_t* perm)
{
    if (!e) return FQD_ERR_ARG;
    if (!t || (t->n && (!perm || !t->offsets || !t->lengths)) || t->n >= 0xFFFFFFFFull)
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_sort_tags: bad arguments");
    if (t->n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    const uint64_t n = t->n;

    // scratch 0: [state: maxlen | lo | hi][keys_a][keys_b][perm_b][rocPRIM temp]
    size_t temp_bytes = 0;
    JOIN_TRY(e, rocprim::radix_sort_pairs(nullptr, temp_bytes, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                          static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 64, stream));
    const size_t state_bytes = 256;
    const size_t keys_bytes = ((n * sizeof(uint64_t)) + 255) & ~size_t(255);
    const size_t perm_bytes = ((n * sizeof(uint32_t)) + 255) & ~size_t(255);
    void* base = nullptr;
    int rc = fqd_internal_scratch(e, 0, state_bytes + 2 * keys_bytes + perm_bytes + temp_bytes + 256, &base);
    if (rc) return rc;
    char* p = static_cast<char*>(base);
    unsigned long long* state = reinterpret_cast<unsigned long long*>(p);      p += state_bytes;
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(p);                         p += keys_bytes;
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(p);                         p += keys_bytes;
    uint32_t* perm_alt = reinterpret_cast<uint32_t*>(p);                       p += perm_bytes;
    void* temp = p;

    JOIN_TRY(e, hipMemsetAsync(state, 0, state_bytes, stream));
    hipLaunchKernelGGL(iota_and_maxlen_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream,
                       perm, t->lengths, n, reinterpret_cast<unsigned int*>(state));
    unsigned long long host_state[3] = {0, 0, 0};
    JOIN_TRY(e, hipMemcpyAsync(host_state, state, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    const uint32_t maxlen = uint32_t(host_state[0] & 0xFFFFFFFFu);
    const int n_chunks = int((maxlen + 7) / 8);

    uint32_t* cur = perm; uint32_t* alt = perm_alt;
    for (int c = -1; c < n_chunks; ++c) {                    // least significant first: length, then last chunk .. first
        const int chunk = c < 0 ? -1 : n_chunks - 1 - c;
        host_state[1] = ~0ull; host_state[2] = 0;
        JOIN_TRY(e, hipMemcpyAsync(state + 1, host_state + 1, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(chunk_keys_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, *t, cur, chunk, keys_a, state + 1);
        JOIN_TRY(e, hipMemcpyAsync(host_state + 1, state + 1, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        JOIN_TRY(e, hipStreamSynchronize(stream));
        if (host_state[1] == host_state[2]) continue;        // every tag agrees on this key: order unchanged
        // only the bits that vary need sorting
        const unsigned long long diff = host_state[1] ^ host_state[2];
        const unsigned end_bit = 64u - unsigned(__builtin_clzll(diff));
        JOIN_TRY(e, rocprim::radix_sort_pairs(temp, temp_bytes, keys_a, keys_b, cur, alt, n, 0, end_bit, stream));
        std::swap(cur, alt);
    }
    if (cur != perm) JOIN_TRY(e, hipMemcpyAsync(perm, cur, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    JOIN_TRY(e, hipStreamSynchronize(stream));
    return FQD_OK;
}

int fqd_match_sorted_tags(fqd_engine* e, const fqd_tags* a, const uint32_t* perm_a,
                          const fqd_tags* b, const uint32_t* perm_b, uint32_t* match)
{
    if (!e) return FQD_ERR_ARG;
    if (!a || !b || (a->n && (!perm_a || !match)) || (b->n && !perm_b) || a->n >= 0xFFFFFFFFull || b->n >= 0xFFFFFFFFull)
        return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_match_sorted_tags: bad arguments");
    if (a->n == 0) return FQD_OK;
    JOIN_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    hipLaunchKernelGGL(match_kernel, dim3(grid_for(a->n)), dim3(kBlock), 0, stream, *a, perm_a, *b, perm_b, match);
    JOIN_TRY(e, hipGetLastError());
    return FQD_OK;
}

} // extern "C"
