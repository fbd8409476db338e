// fqd_gunzip.hip — an ORDINARY gzip member inflated on the GPU (same library as fqd_engine.hip): the kernels around
// fqd_gunzip_core.hpp and the entry point that strings them together.
//
// Replaces, for regular files read by the GPU-resident runs, the gzip decompressor the reference pushes onto its input
// stream (file_utils.cpp:59-66) where the file is NOT BGZF — one long deflate stream, what gzip, pigz and sequencers
// write (BGZF: fqd_inflate.hip).  The scheme is the one of host/pgzip.hpp (pugz, rapidgzip), with the chip's waves in the
// place of a handful of threads:
//   gz_find_starts_kernel   one wave per unit of the compressed bytes: 64 bit offsets at a time are given a first look
//                           (three header bits, counts, a complete code-length code: registers only); the few that pass
//                           get the second look (all code lengths, both codes complete) one after the other;
//   gz_decode_kernel        one wave per unit with a start: lane 0 decodes serially (the tables of the block in LDS, the
//                           bits in a register) into a ring of 16-bit symbols in LDS, from which matches are copied;
//                           the whole wave moves the ring's new symbols out to HBM, 16 bytes a lane;
//   gz_windows_kernel       one workgroup, unit after unit: the 32 KiB of text that end with the unit, from its symbols and
//                           the window before it;
//   gz_resolve_kernel       every symbol becomes a byte: one workgroup per unit, eight symbols a lane and step;
//   gz_crc_kernel           CRC-32 of every 64 KiB of the text; the host folds them (one 32 x 32 bit matrix per fold).
// Anything irregular — a chain of unit ends and starts that does not close, damaged data, a unit that outgrows its room,
// CRC or length that differ from the trailer — is only REPORTED (ok = 0): the caller reads the file the host way, whose
// diagnostics are the reference's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

#include "../../include/fqdupaway.h"
#include "fqd_gunzip_core.hpp"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);

namespace {

using namespace fqd::gunz;

#define GZ_TRY(e, expr)                                                                     \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess) { (void)hipGetLastError();       \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } } while (0)

constexpr uint32_t kWave = 64;
constexpr uint32_t kRing = 4096;                  // symbols of a decoder's LDS ring (8 KiB): what it put last, matches copy out of it
constexpr uint32_t kStretch = kRing / 2 - 512;    // symbols lane 0 decodes before the wave moves them out (a code adds up to 258 more)

// ---- 1. where units start --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave)
void gz_find_starts_kernel(BitIn in, uint64_t unit_bits, uint64_t n_units, uint64_t* __restrict__ start)
{
    __shared__ uint8_t lens[320];
    __shared__ uint32_t found;
    const uint32_t lane = threadIdx.x;
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        if (u == 0) { if (lane == 0) start[0] = 0; continue; }
        const uint64_t lo = u * unit_bits, hi = lo + unit_bits < in.nbits ? lo + unit_bits : in.nbits;
        uint64_t at = ~0ull;
        for (uint64_t p0 = lo; p0 < hi && at == ~0ull; p0 += kWave) {
            const uint64_t p = p0 + lane;
            unsigned long long cand = __ballot(p < hi && block_start_first_look(in, p));
            while (cand) {                                                // the few that pass, lowest offset first
                const uint32_t l = uint32_t(__ffsll(static_cast<long long>(cand))) - 1u;
                if (lane == 0) found = block_start_second_look(in, p0 + l, lens) ? 1u : 0u;
                __syncthreads();
                const bool yes = found != 0u;
                __syncthreads();
                if (yes) { at = p0 + l; break; }
                cand &= cand - 1ull;
            }
        }
        if (lane == 0) start[u] = at;
    }
}

// ---- 2. a unit decoded into symbols ------------------------------------------------------------------------------------------
struct UnitIn  { uint64_t start_bit, stop_bit, sym_at, sym_cap; };           // where to start, the next unit's nominal start, room in the symbol scratch
struct UnitOut { uint64_t end_bit, n_sym; uint32_t status, deepest; };

// Lane 0's sink: the ring in LDS; what has left the ring is read back from HBM (matches deeper than the ring keeps: rare in
// FASTQ, where a match reaches a record or two back).
struct RingSink {
    uint16_t* ring;                      // LDS, kRing symbols
    const uint16_t* out;                 // HBM: the unit's symbols moved out so far ([0, flushed))
    uint64_t cap, n, flushed;
    uint32_t lane;
    __device__ __forceinline__ bool room(uint32_t need) const { return n + need <= cap; }
    __device__ __forceinline__ void put(uint16_t s) { ring[n & (kRing - 1u)] = s; ++n; }       // (every lane the same symbol to the same place)
    // the ring still holds place i unless i + kRing has been put; n never gets further than flushed + kStretch + 258 + 7
    __device__ __forceinline__ uint16_t at(uint64_t i) const { return i + kRing > flushed + kStretch + 512u ? ring[i & (kRing - 1u)] : out[i]; }
    // a match: the places it reads all hold their symbols before it starts (k mod d), so the lanes copy 64 symbols a turn
    __device__ __forceinline__ void copy(uint32_t d, uint32_t len)
    {
        const int64_t from = int64_t(n) - int64_t(d);
        for (uint32_t k = lane; k < len; k += kWave) {
            const int64_t i = from + int64_t(k < d ? k : k % d);
            ring[(n + k) & (kRing - 1u)] = i < 0 ? uint16_t(256 + int64_t(kWindow) + i) : at(uint64_t(i));
        }
        n += len;
    }
    __device__ __forceinline__ uint64_t count() const { return n; }
};

__global__ __launch_bounds__(kWave)
void gz_decode_kernel(BitIn in, const UnitIn* __restrict__ units, uint32_t n_units, uint16_t* __restrict__ sym, UnitOut* __restrict__ result,
                      uint32_t* __restrict__ next_unit, unsigned long long* __restrict__ debug /* 8 words, FQD_GUNZIP_TRACE */)
{
    __shared__ Tables tables;
    __shared__ __attribute__((aligned(16))) uint16_t ring[kRing];
    __shared__ uint8_t lens[320];
    __shared__ uint32_t my_unit;
    __shared__ uint64_t sh_n;
    __shared__ uint32_t sh_status;
    const uint32_t lane = threadIdx.x;
    for (;;) {
        if (lane == 0) my_unit = atomicAdd(next_unit, 1u);
        __syncthreads();
        const uint32_t u = my_unit;
        __syncthreads();
        if (u >= n_units) break;
        const UnitIn ui = units[u];
        uint16_t* out = sym + ui.sym_at;
        State st;
        st.pos = st.start_bit = ui.start_bit;
        RingSink sink{ring, out, ui.sym_cap, 0, 0, lane};
        uint64_t flushed = 0;
        for (uint64_t stretch = 0;; ++stretch) {
            // EVERY lane runs the decoder, on the same bits to the same end: the control flow of 6700 instructions stays uniform
            // (a first version ran it under `if (lane == 0)` and never came back from the GPU, while the very same code, ring and
            // all, runs to its end on the CPU); what the lanes write — tables, ring — they all write alike
            sink.flushed = flushed;
            decode_some(in, tables, lens, st, ui.stop_bit, sink, kStretch);
            if (stretch > (ui.sym_cap / kStretch) + 16u && st.status == kOk) st.status = kBadData;          // (cannot happen: every stretch adds symbols)
            if (lane == 0) {
                sh_n = sink.n; sh_status = st.status;
                if (debug) { debug[0] = stretch; debug[1] = sink.n; debug[2] = st.pos; debug[3] = st.status; debug[4] = st.in_block; debug[5] = u; }
            }
            __syncthreads();
            const uint64_t n = sh_n;
            const uint32_t status = sh_status;
            // whole groups of eight symbols leave the ring, 16 bytes a lane (the unit's room starts 16-byte aligned); the last few wait
            const uint64_t upto = status == kOk ? n & ~7ull : n;
            for (uint64_t g = (flushed >> 3) + lane; g < (upto >> 3); g += kWave)
                reinterpret_cast<uint4*>(out)[g] = reinterpret_cast<const uint4*>(ring)[g & (kRing / 8u - 1u)];
            if (status != kOk) for (uint64_t i = (upto & ~7ull) + lane; i < upto; i += kWave) out[i] = ring[i & (kRing - 1u)];
            flushed = upto & ~7ull;
            __syncthreads();                                              // (the stores are visible to lane 0's later loads: one workgroup)
            if (status != kOk) break;
        }
        if (lane == 0) result[u] = UnitOut{st.pos, sink.n, st.status, st.deepest};
    }
}

// ---- 4. windows, unit after unit -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void gz_windows_kernel(const UnitIn* __restrict__ units, const UnitOut* __restrict__ result, uint32_t n_units, const uint16_t* __restrict__ sym,
                       uint8_t* __restrict__ windows /* (n_units + 1) x kWindow; [0] is given */)
{
    for (uint32_t u = 0; u < n_units; ++u) {
        const uint8_t* prev = windows + uint64_t(u) * kWindow;
        uint8_t* next = windows + uint64_t(u + 1u) * kWindow;
        const uint16_t* s = sym + units[u].sym_at;
        const uint64_t n = result[u].n_sym;
        for (uint32_t k = threadIdx.x; k < kWindow; k += 1024u) next[k] = window_byte(prev, s, n, k);
        __syncthreads();
    }
}

// ---- 5. bytes ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void gz_resolve_kernel(const UnitIn* __restrict__ units, const UnitOut* __restrict__ result, const uint64_t* __restrict__ text_at, uint32_t n_units,
                       const uint16_t* __restrict__ sym, const uint8_t* __restrict__ windows, uint8_t* __restrict__ text)
{
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint16_t* s = sym + units[u].sym_at;                        // 16-byte aligned
        const uint8_t* win = windows + uint64_t(u) * kWindow;
        uint8_t* dst = text + text_at[u];
        const uint64_t n = result[u].n_sym;
        const bool plain = result[u].deepest == 0u;
        for (uint64_t g = threadIdx.x; g < (n >> 3); g += 256u) {
            const uint4 v = reinterpret_cast<const uint4*>(s)[g];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint8_t b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t x = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                b[k] = (plain || x < 256u) ? uint8_t(x) : win[x - 256u];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[8u * g + k] = b[k];          // (dst has whatever alignment the units before left it)
        }
        for (uint64_t i = (n & ~7ull) + threadIdx.x; i < n; i += 256u) { const uint32_t x = s[i]; dst[i] = x < 256u ? uint8_t(x) : win[x - 256u]; }
    }
}

// ---- CRC-32 of every 64 KiB slice ----------------------------------------------------------------------------------------------
struct CrcTable { uint32_t t[256]; };
constexpr CrcTable make_crc_table()
{
    CrcTable c{};
    for (uint32_t i = 0; i < 256; ++i) { uint32_t r = i; for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (r & 1u ? 0xEDB88320u : 0u); c.t[i] = r; }
    return c;
}
__device__ const CrcTable g_gz_crc_table = make_crc_table();
constexpr uint32_t kSlice = 65536, kCrcThreads = 256, kCrcChunk = kSlice / kCrcThreads;       // 256 bytes a lane

// The raw CRC register after each lane's 256 bytes (started from 0), then folded pairwise: reg(A|B) = advance(reg(A), |B|) ^ reg(B)
// with advance = "as many zero bytes", a 32 x 32 bit matrix per level given by the host (shift[k]: 256 * 2^k bytes).  Raw
// registers started from 0 make the combination linear; the slice's CRC-32 as zlib has it follows from advance(0xFFFFFFFF, |slice|).
__global__ __launch_bounds__(kCrcThreads)
void gz_crc_kernel(const uint8_t* __restrict__ text, uint64_t total, uint64_t n_slices, const uint32_t* __restrict__ shift /* [8][32] */,
                   uint32_t* __restrict__ raw_out)
{
    __shared__ uint32_t table[256];
    __shared__ uint32_t reg[kCrcThreads];
    const uint32_t t = threadIdx.x;
    table[t] = g_gz_crc_table.t[t];
    __syncthreads();
    for (uint64_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint64_t lo = sl * kSlice + uint64_t(t) * kCrcChunk;
        const uint64_t end = (sl + 1) * kSlice < total ? (sl + 1) * kSlice : total;
        // a short last slice: its bytes are moved to the END of the 64 KiB frame (leading zero bytes do not change a raw register started from 0)
        const uint64_t len = end - sl * kSlice, pad = kSlice - len;
        uint32_t r = 0;
        if (pad == 0 && (reinterpret_cast<uintptr_t>(text + lo) & 15u) == 0) {       // a whole slice: 16 bytes a load
            const uint4* p16 = reinterpret_cast<const uint4*>(text + lo);
            for (uint32_t k = 0; k < kCrcChunk / 16u; ++k) {
                const uint4 v = p16[k];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 16; ++j) r = table[(r ^ (w[j >> 2] >> (8 * (j & 3)))) & 0xFFu] ^ (r >> 8);
            }
        } else {
            for (uint32_t k = 0; k < kCrcChunk; ++k) {
                const uint64_t frame = uint64_t(t) * kCrcChunk + k;             // position in the frame
                const uint8_t b = frame >= pad ? text[sl * kSlice + (frame - pad)] : uint8_t(0);
                r = table[(r ^ b) & 0xFFu] ^ (r >> 8);
            }
        }
        reg[t] = r;
        __syncthreads();
        for (uint32_t k = 0; k < 8u; ++k) {
            if ((t & ((2u << k) - 1u)) == 0u) {
                uint32_t a = reg[t], adv = 0;
                const uint32_t* m = shift + 32u * k;
#pragma unroll 4
                for (uint32_t b = 0; b < 32u; ++b) adv ^= (a >> b) & 1u ? m[b] : 0u;
                reg[t] = adv ^ reg[t + (1u << k)];
            }
            __syncthreads();
        }
        if (t == 0) raw_out[sl] = reg[0];
        __syncthreads();
    }
}

// GF(2) helpers of the host side: the register after n zero bytes, as a matrix on the register's bits.
struct Mat { uint32_t col[32]; };
uint32_t mat_apply(const Mat& m, uint32_t v) { uint32_t r = 0; for (uint32_t b = 0; b < 32; ++b) if ((v >> b) & 1u) r ^= m.col[b]; return r; }
Mat mat_square(const Mat& m) { Mat r; for (uint32_t b = 0; b < 32; ++b) r.col[b] = mat_apply(m, m.col[b]); return r; }
Mat mat_one_zero_byte()
{
    Mat m;
    for (uint32_t b = 0; b < 32; ++b) {
        uint32_t r = 1u << b;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (r & 1u ? 0xEDB88320u : 0u);
        m.col[b] = r;
    }
    return m;
}
uint32_t advance_zero_bytes(uint32_t reg, uint64_t n)
{
    Mat m = mat_one_zero_byte();
    while (n) { if (n & 1u) reg = mat_apply(m, reg); n >>= 1; if (n) m = mat_square(m); }
    return reg;
}

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    hipError_t get(size_t bytes) { if (p) { (void)hipFree(p); p = nullptr; } return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

} // namespace

extern "C" {

int fqd_gunzip(fqd_engine* e, const uint8_t* deflate, uint64_t avail_bytes, uint8_t* text, uint64_t text_cap,
               uint64_t* text_bytes, uint64_t* deflate_bytes, uint32_t* crc32, int32_t* ok)
{
    if (!e) return FQD_ERR_ARG;
    if (!deflate || !text || !text_bytes || !deflate_bytes || !crc32 || !ok) return fqd_internal_fail(e, FQD_ERR_ARG, "fqd_gunzip: bad arguments");
    *ok = 0; *text_bytes = 0; *deflate_bytes = 0; *crc32 = 0;
    if (avail_bytes < 2) return FQD_OK;
    GZ_TRY(e, hipSetDevice(fqd_internal_device(e)));
    hipStream_t stream = fqd_internal_stream(e);
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, fqd_internal_device(e)) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }

    static const bool trace = std::getenv("FQD_GUNZIP_TRACE") != nullptr;      // stages to stderr as they are reached
#define GZ_TRACE(...) do { if (trace) { std::fprintf(stderr, "[gunzip] " __VA_ARGS__); std::fputc('\n', stderr); std::fflush(stderr); } } while (0)
    BitIn in;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(deflate);
    in.words = reinterpret_cast<const uint64_t*>(addr & ~uintptr_t(7));
    in.lead = uint64_t(addr & 7u) * 8u;
    in.nbits = avail_bytes * 8u;

    // units: small enough that there are several per wave slot of the chip, large enough to hold a block start more often than
    // not (zlib ends a block every 16 K codes: some 20-50 KB packed).  (One decoder per LANE instead of per wave — tables in HBM,
    // 16 KiB units, tens of thousands of lanes waiting on memory at once — was measured: 2.6 GB/s of text against 4.6; lanes of
    // a wave that copy matches of different lengths wait for the longest.)
    uint64_t unit_bytes = std::min<uint64_t>(512u << 10, std::max<uint64_t>(64u << 10, (avail_bytes / 8192u + 4095u) & ~uint64_t(4095)));
    if (const char* v = std::getenv("FQD_GUNZIP_UNIT_KB")) { const long kb = std::atol(v); if (kb > 0) unit_bytes = uint64_t(kb) << 10; }
    uint64_t ratio = 8;                                                      // symbols of room per compressed byte (FASTQ packs 3-6 fold)
    if (const char* v = std::getenv("FQD_GUNZIP_RATIO")) { const long r = std::atol(v); if (r > 0) ratio = uint64_t(r); }
    const uint64_t n_nominal = (avail_bytes + unit_bytes - 1) / unit_bytes;

    // ---- 1. starts
    DevMem d_start;
    GZ_TRY(e, d_start.get(n_nominal * 8));
    hipLaunchKernelGGL(gz_find_starts_kernel, dim3(uint32_t(std::min<uint64_t>(n_nominal, uint64_t(n_cu) * 16u))), dim3(kWave), 0, stream,
                       in, unit_bytes * 8u, n_nominal, d_start.as<uint64_t>());
    GZ_TRY(e, hipGetLastError());
    GZ_TRACE("%llu bytes, %llu units of %llu bytes: looking for block starts", (unsigned long long)avail_bytes, (unsigned long long)n_nominal, (unsigned long long)unit_bytes);
    std::vector<uint64_t> start(n_nominal);
    GZ_TRY(e, hipMemcpyAsync(start.data(), d_start.p, n_nominal * 8, hipMemcpyDeviceToHost, stream));
    GZ_TRY(e, hipStreamSynchronize(stream));
    std::vector<UnitIn> units;
    for (uint64_t u = 0; u < n_nominal; ++u) {
        if (start[u] == ~0ull) continue;
        uint64_t next = u + 1;
        while (next < n_nominal && start[next] == ~0ull) ++next;             // a unit without a start belongs to the one before it
        UnitIn x;
        x.start_bit = start[u];
        x.stop_bit = next < n_nominal ? next * unit_bytes * 8u : ~0ull;
        x.sym_cap = ((next - u) * unit_bytes * ratio + 1024u) & ~uint64_t(7);
        x.sym_at = 0;
        units.push_back(x);
    }

    GZ_TRACE("%zu units have a start", units.size());
    // ---- 2-5 in batches of units whose symbols fit the scratch
    uint64_t scratch_syms = uint64_t(4) << 30;                               // 8 GiB of symbols
    if (const char* v = std::getenv("FQD_GUNZIP_SCRATCH_MB")) { const long mb = std::atol(v); if (mb > 0) scratch_syms = (uint64_t(mb) << 20) / 2; }
    uint64_t largest = 0;
    for (const UnitIn& x : units) largest = std::max(largest, x.sym_cap);
    scratch_syms = std::max(scratch_syms, largest);
    {
        uint64_t all = 0;
        for (const UnitIn& x : units) all += x.sym_cap;
        scratch_syms = std::min(scratch_syms, all);
    }
    DevMem d_sym, d_units, d_result, d_text_at, d_windows, d_counter, d_debug;
    unsigned long long* h_debug = nullptr;
    if (trace) {                                                              // host-visible, so that a kernel that never ends can still be read
        GZ_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&h_debug), 64, hipHostMallocMapped));
        std::memset(h_debug, 0xFF, 64);
        d_debug.p = nullptr;
    }
    GZ_TRY(e, d_sym.get(scratch_syms * 2 + 64));
    GZ_TRY(e, d_counter.get(64));
    std::vector<uint8_t> carry(kWindow, 0);                                   // the window before the next batch's first unit
    uint64_t total = 0, expect_start = 0, repairs = 0;
    bool final_seen = false, good = true;
    std::vector<UnitOut> result;
    std::vector<uint64_t> text_at;
    size_t at = 0;
    while (at < units.size() && good && !final_seen) {
        size_t hi = at; uint64_t used = 0;
        while (hi < units.size() && used + units[hi].sym_cap <= scratch_syms) { units[hi].sym_at = used; used += units[hi].sym_cap; ++hi; }
        const uint32_t nb = uint32_t(hi - at);
        if (nb == 0) { good = false; break; }
        GZ_TRY(e, d_units.get(nb * sizeof(UnitIn)));
        GZ_TRY(e, d_result.get(nb * sizeof(UnitOut)));
        GZ_TRY(e, d_text_at.get(nb * 8));
        GZ_TRY(e, d_windows.get(uint64_t(nb + 1u) * kWindow));
        GZ_TRY(e, hipMemcpyAsync(d_units.p, units.data() + at, nb * sizeof(UnitIn), hipMemcpyHostToDevice, stream));
        GZ_TRY(e, hipMemsetAsync(d_counter.p, 0, 64, stream));
        hipLaunchKernelGGL(gz_decode_kernel, dim3(std::min<uint32_t>(nb, uint32_t(n_cu) * 10u)), dim3(kWave), 0, stream,
                           in, d_units.as<const UnitIn>(), nb, d_sym.as<uint16_t>(), d_result.as<UnitOut>(), d_counter.as<uint32_t>(),
                           h_debug);
        GZ_TRY(e, hipGetLastError());
        GZ_TRACE("batch of %u units queued for decoding (%llu symbols of room)", nb, (unsigned long long)used);
        if (trace) for (int tick = 0; tick < 20 && hipStreamQuery(stream) == hipErrorNotReady; ++tick) {
            usleep(200000);
            GZ_TRACE("  ... stretch %llu symbols %llu pos %llu status %llu in_block %llu unit %llu", h_debug[0], h_debug[1], h_debug[2], h_debug[3], h_debug[4], h_debug[5]);
        }
        result.resize(nb);
        GZ_TRY(e, hipMemcpyAsync(result.data(), d_result.p, nb * sizeof(UnitOut), hipMemcpyDeviceToHost, stream));
        GZ_TRY(e, hipStreamSynchronize(stream));
        GZ_TRACE("decoded: first unit status %u, %llu symbols, end bit %llu", result[0].status, (unsigned long long)result[0].n_sym, (unsigned long long)result[0].end_bit);
        // ---- 3. the chain
        text_at.resize(nb);
        uint32_t live = 0;
        for (uint32_t k = 0; k < nb && good && !final_seen; ++k) {
            UnitIn& x = units[at + k]; UnitOut& r = result[k];
            if (x.start_bit != expect_start) {
                // a guess that did not hold (a header-like stretch of bits inside a block: a few per gigabyte): the unit before ended
                // at the true boundary, so this one is decoded again from there — one small launch — and the chain goes on
                if (++repairs > 64u + units.size() / 16u) { good = false; break; }       // (damage, not bad luck)
                x.start_bit = expect_start;
                GZ_TRY(e, hipMemcpyAsync(d_units.as<UnitIn>() + k, &x, sizeof(UnitIn), hipMemcpyHostToDevice, stream));
                GZ_TRY(e, hipMemsetAsync(d_counter.p, 0, 64, stream));
                hipLaunchKernelGGL(gz_decode_kernel, dim3(1), dim3(kWave), 0, stream, in, d_units.as<const UnitIn>() + k, 1u, d_sym.as<uint16_t>(),
                                   d_result.as<UnitOut>() + k, d_counter.as<uint32_t>(), static_cast<unsigned long long*>(nullptr));
                GZ_TRY(e, hipGetLastError());
                GZ_TRY(e, hipMemcpyAsync(&r, d_result.as<UnitOut>() + k, sizeof(UnitOut), hipMemcpyDeviceToHost, stream));
                GZ_TRY(e, hipStreamSynchronize(stream));
                GZ_TRACE("unit %zu decoded again from bit %llu: status %u, %llu symbols", at + k, (unsigned long long)expect_start, r.status, (unsigned long long)r.n_sym);
            }
            if (r.status != kBoundary && r.status != kFinal) { good = false; break; }
            if (total + r.n_sym > text_cap) { good = false; break; }
            text_at[k] = total; total += r.n_sym;
            expect_start = r.end_bit;
            live = k + 1;
            if (r.status == kFinal) final_seen = true;
        }
        if (!good) break;
        // ---- 4, 5
        GZ_TRY(e, hipMemcpyAsync(d_text_at.p, text_at.data(), live * 8, hipMemcpyHostToDevice, stream));
        GZ_TRY(e, hipMemcpyAsync(d_windows.p, carry.data(), kWindow, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gz_windows_kernel, dim3(1), dim3(1024), 0, stream, d_units.as<const UnitIn>(), d_result.as<const UnitOut>(), live,
                           d_sym.as<const uint16_t>(), d_windows.as<uint8_t>());
        hipLaunchKernelGGL(gz_resolve_kernel, dim3(std::min<uint32_t>(live, uint32_t(n_cu) * 8u)), dim3(256), 0, stream,
                           d_units.as<const UnitIn>(), d_result.as<const UnitOut>(), d_text_at.as<const uint64_t>(), live,
                           d_sym.as<const uint16_t>(), d_windows.as<const uint8_t>(), text);
        GZ_TRY(e, hipGetLastError());
        GZ_TRY(e, hipMemcpyAsync(carry.data(), d_windows.as<uint8_t>() + uint64_t(live) * kWindow, kWindow, hipMemcpyDeviceToHost, stream));
        GZ_TRY(e, hipStreamSynchronize(stream));
        GZ_TRACE("windows and bytes of %u units done, %llu bytes of text so far", live, (unsigned long long)total);
        at = hi;
    }
    if (!good || !final_seen) return FQD_OK;                                  // *ok stays 0: the caller reads the file the host way

    // ---- CRC-32 of the text
    uint32_t crc = 0;
    if (total) {
        const uint64_t n_slices = (total + kSlice - 1) / kSlice;
        DevMem d_shift, d_raw;
        GZ_TRY(e, d_shift.get(8 * 32 * 4));
        GZ_TRY(e, d_raw.get(n_slices * 4));
        uint32_t shift[8][32];
        Mat m = mat_one_zero_byte();
        for (int k = 0; k < 8; ++k) m = mat_square(m);                        // 256 zero bytes
        for (int k = 0; k < 8; ++k) { std::memcpy(shift[k], m.col, sizeof m.col); m = mat_square(m); }
        const Mat slice_mat = m;                                              // 65536 zero bytes
        GZ_TRY(e, hipMemcpyAsync(d_shift.p, shift, sizeof shift, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gz_crc_kernel, dim3(uint32_t(std::min<uint64_t>(n_slices, uint64_t(n_cu) * 8u))), dim3(kCrcThreads), 0, stream,
                           static_cast<const uint8_t*>(text), total, n_slices, d_shift.as<const uint32_t>(), d_raw.as<uint32_t>());
        GZ_TRY(e, hipGetLastError());
        GZ_TRACE("CRC of %llu slices queued", (unsigned long long)n_slices);
        std::vector<uint32_t> raw(n_slices);
        GZ_TRY(e, hipMemcpyAsync(raw.data(), d_raw.p, n_slices * 4, hipMemcpyDeviceToHost, stream));
        GZ_TRY(e, hipStreamSynchronize(stream));
        // raw register of the whole text started from 0: full slices fold with the 64 KiB matrix, the short last one with its own length
        uint32_t reg = 0;
        for (uint64_t s = 0; s < n_slices; ++s) {
            const uint64_t len = s + 1 < n_slices ? kSlice : total - s * kSlice;
            reg = (len == kSlice ? mat_apply(slice_mat, reg) : advance_zero_bytes(reg, len)) ^ raw[s];
        }
        crc = reg ^ advance_zero_bytes(0xFFFFFFFFu, total) ^ 0xFFFFFFFFu;     // the same register started from all ones, then inverted: zlib's CRC-32
    }
    *text_bytes = total;
    *deflate_bytes = (expect_start + 7) / 8;
    *crc32 = crc;
    *ok = 1;
    return FQD_OK;
}

} // extern "C"
