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
//   gz_decode_planes_kernel one wave per unit: the wave decoder of the BGZF reader (fqd_inflate_wave.hpp) writing TWO texts of the
//                           unit behind two made-up windows — what comes out the same in both planes is a byte of the stream,
//                           what differs says which byte of the unknown window it is;
//   gz_window_maps_kernel,  the 32 KiB of text that end with a unit: which place of the window before every place copies (all units at
//   gz_windows_chain_kernel once), then the windows themselves unit after unit, the running window in LDS;
//   gz_resolve_kernel       every place of every unit becomes a byte: one workgroup per unit, sixteen places a lane and step;
//   gz_crc_kernel           CRC-32 of every 64 KiB of the text; the host folds them (one 32 x 32 bit matrix per fold).
// Anything irregular — a chain of unit ends and starts that does not close, damaged data, a unit that outgrows its room,
// CRC or length that differ from the trailer — is only REPORTED (ok = 0): the caller reads the file the host way, whose
// diagnostics are the reference's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <unordered_map>
#include <vector>
#include <unistd.h>

#include "../../include/fqdupaway.h"
#include "fqd_gunzip_core.hpp"
#include "fqd_inflate_wave.hpp"

#define FQD_HIDDEN __attribute__((visibility("hidden")))
FQD_HIDDEN hipStream_t fqd_internal_stream(fqd_engine* e);
FQD_HIDDEN int fqd_internal_device(fqd_engine* e);
FQD_HIDDEN int fqd_internal_fail(fqd_engine* e, int code, const char* msg);
FQD_HIDDEN int fqd_internal_scratch(fqd_engine* e, int which, size_t bytes, void** out);

namespace {

using namespace fqd::gunz;

#define GZ_TRY(e, expr)                                                                     \
    do { hipError_t err_ = (expr); if (err_ != hipSuccess) { (void)hipGetLastError();       \
        return fqd_internal_fail(e, FQD_ERR_HIP, hipGetErrorString(err_)); } } while (0)

constexpr uint32_t kWave = 64;

// ---- 1. where units start --------------------------------------------------------------------------------------------------
// Every bit offset of a unit is sifted, 64 a turn, with what 13 bits can tell (final-block bit, block type, code counts: one
// offset in nine passes) out of two words the wave slides along the stream; what passes WAITS until 64 offsets do and is then
// given the first look proper (a complete code-length code: one in a hundred) by 64 lanes side by side; what passes that
// waits for the second look (some 300 code lengths read, two codes checked: 15 000 instructions), again 64 side by side.
// The first offset of the unit that passes everything is its start.  (First version: every look for every offset as it came,
// the second one lane at a time — 25 of 168 ms per gigabyte of text; the looks in stages: 5.)
__global__ __launch_bounds__(kWave)
void gz_find_starts_kernel(BitIn in, uint64_t unit_bits, uint64_t first_unit, uint64_t n_units /* one past the last */, uint64_t* __restrict__ start)
{
    __shared__ uint8_t lens[kWave][320];
    __shared__ uint64_t sifted[2 * kWave], looked[2 * kWave];
    const uint32_t lane = threadIdx.x;
    const uint64_t last_word = ((in.nbits + in.lead) >> 6) + 3u;              // 32 readable bytes lie behind the stream: no word beyond them is asked for
    auto word = [&](uint64_t i) { return in.words[i <= last_word ? i : last_word]; };
    for (uint64_t u = first_unit + blockIdx.x; u < n_units; u += gridDim.x) {
        if (u == 0) { if (lane == 0) start[0] = 0; continue; }
        const uint64_t lo = u * unit_bits, hi = lo + unit_bits < in.nbits ? lo + unit_bits : in.nbits;
        uint64_t at = ~0ull;
        uint32_t n_sifted = 0, n_looked = 0;
        // takes the first n (<= 64) of a queue, keeps the rest; `test` runs on every lane that has an entry
        auto drain = [&](uint64_t* q, uint32_t& have, uint32_t n, auto&& test) -> unsigned long long {
            __syncthreads();
            const uint64_t mine = lane < n ? q[lane] : 0ull;
            const bool yes = lane < n && test(mine);
            const unsigned long long hit = __ballot(yes);
            const uint64_t keep = (lane < have - n) ? q[n + lane] : 0ull;      // (have - n <= 64)
            __syncthreads();
            if (lane < have - n) q[lane] = keep;
            have -= n;
            __syncthreads();
            return hit;
        };
        auto first_looks = [&](uint32_t n) {                                  // sifted -> looked, order kept
            const uint64_t mine = lane < n ? sifted[lane] : 0ull;             // (read before drain moves the queue)
            const unsigned long long hit = drain(sifted, n_sifted, n, [&](uint64_t p) { return block_start_first_look(in, p); });
            if ((hit >> lane) & 1ull) looked[n_looked + uint32_t(__popcll(hit & ((1ull << lane) - 1ull)))] = mine;
            n_looked += uint32_t(__popcll(hit));
        };
        auto second_looks = [&](uint32_t n) {                                 // true: one of the first n looked-at offsets is a block start
            const uint64_t mine = lane < n ? looked[lane] : 0ull;
            const unsigned long long hit = drain(looked, n_looked, n, [&](uint64_t p) { return block_start_second_look(in, p, lens[lane]); });
            if (hit) at = __shfl(mine, int(__ffsll(static_cast<long long>(hit))) - 1, 64);
            return hit != 0ull;
        };
        uint64_t wi = (lo + in.lead) >> 6;
        uint64_t A = word(wi), B = word(wi + 1), C = word(wi + 2);           // a turn is 64 offsets = one word: [A B] hold a lane's 13 bits, C is a turn ahead
        bool found = false;
        for (uint64_t p0 = lo; p0 < hi && !found; p0 += kWave) {
            const uint64_t p = p0 + lane;
            const uint64_t ahead = word(wi + 3);
            const uint32_t s = uint32_t((p0 + in.lead) & 63u) + lane;        // the lane's offset from the first bit of A: 0 .. 126
            const uint64_t x = s < 64u ? A : B, y = s < 64u ? B : C;
            const uint32_t r = s & 63u;
            const uint32_t w13 = uint32_t(r ? (x >> r) | (y << (64u - r)) : x) & 0x1FFFu;
            const bool pass = p < hi && p + 300 <= in.nbits && block_start_header_bits(w13);
            A = B; B = C; C = ahead; ++wi;
            const unsigned long long cand = __ballot(pass);
            if (pass) sifted[n_sifted + uint32_t(__popcll(cand & ((1ull << lane) - 1ull)))] = p;
            n_sifted += uint32_t(__popcll(cand));
            if (n_sifted >= kWave) {
                first_looks(kWave);
                if (n_looked >= kWave) found = second_looks(kWave);
            }
        }
        if (!found) {
            if (n_sifted) first_looks(n_sifted);
            while (!found && n_looked) found = second_looks(n_looked < kWave ? n_looked : kWave);
        }
        if (lane == 0) start[u] = at;
    }
}

// ---- 2. a unit decoded — into two texts, as BYTES -------------------------------------------------------------------------------------
// The wave decoder of the BGZF reader (fqd_inflate_wave.hpp: the lanes decode a block's bits side by side from guessed code
// boundaries, matches are resolved 64 at a time) writes bytes and copies bytes; what a unit needs are SYMBOLS — "a byte", or
// "the byte at place w of the 32 KiB before me", which nobody knows yet.  It gets them from that decoder as it is, given a
// second output: the unit's text is written twice — every literal stored, every match copied in both — each behind a MADE-UP window,
//     plane P: window[w] = w & 255          plane Q: window[w] = (w & 255) ^ (1 + (w >> 8))
// Copies are the identity on bytes, so a byte that was copied (however often) from place w of the window comes out as P[w] in
// one plane and Q[w] in the other — two values that differ for every w and give w back — and a literal of the stream comes
// out the same in both.  P == Q: the byte; otherwise w = P | ((P ^ Q) - 1) << 8.  Twice the stores and copies at 50 times the speed
// of the one-lane decoder this kernel replaced (csrc/fqd_gunzip_core.hpp, kept as the CPU reference form: 4.6 GB/s of text).
struct UnitIn  { uint64_t start_bit, stop_bit, at, cap; };                   // where to start, the next unit's nominal start, the unit's place in a plane, room behind the window
struct UnitOut { uint64_t end_bit, n; uint32_t status, how; };               // how: 1 a boundary, 2 the final block's end

struct WaveCtx {                                                             // the wave as fqd_inflate_wave.hpp sees it (as in fqd_inflate.hip)
    static constexpr uint32_t kLanes = kWave;
    uint32_t lane;
    template <class F> __device__ __forceinline__ void lanes(F f) { f(lane); __syncthreads(); }
    template <class F> __device__ __forceinline__ void lanes_open(F f) { f(lane); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    template <class F> __device__ __forceinline__ uint64_t ballot(F f) { return __ballot(f(lane) ? 1 : 0); }
    __device__ __forceinline__ uint32_t same(uint32_t v) const { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    __device__ __forceinline__ void add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
    __device__ __forceinline__ void mark(int) {}
};

__device__ __forceinline__ uint32_t made_up(uint32_t plane, uint32_t w) { return plane ? ((w & 255u) ^ (1u + (w >> 8))) : (w & 255u); }

__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gz_decode_planes_kernel(const uint8_t* __restrict__ deflate, uint64_t avail_bytes, const UnitIn* __restrict__ units, uint32_t n_units,
                             uint8_t* __restrict__ plane_p, uint8_t* __restrict__ plane_q, fqd::winf::Token* __restrict__ tokens,
                             UnitOut* __restrict__ result /* [2 * n_units]: unit u, plane p at 2u + p */, uint32_t* __restrict__ next_item)
{
    __shared__ fqd::winf::Shared<kWave> sh;
    __shared__ uint32_t my_item;
    __shared__ uint32_t info[3];
    WaveCtx ctx{threadIdx.x};
    fqd::winf::Token* tok = tokens + size_t(blockIdx.x) * fqd::winf::kTokenRoom;
    for (;;) {
        if (threadIdx.x == 0) my_item = atomicAdd(next_item, 1u);
        __syncthreads();
        const uint32_t u = my_item;
        __syncthreads();
        if (u >= n_units) break;
        const UnitIn ui = units[u];
        uint8_t* out_p = plane_p + ui.at;
        uint8_t* out_q = plane_q + ui.at;
        // the two made-up windows: 32 KiB each, sixteen bytes a lane and store
        for (uint32_t w0 = threadIdx.x * 16u; w0 < kWindow; w0 += kWave * 16u) {
#pragma unroll
            for (uint32_t plane = 0; plane < 2u; ++plane) {
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w = w0 + 4u * k;
                    v[k] = made_up(plane, w) | (made_up(plane, w + 1u) << 8) | (made_up(plane, w + 2u) << 16) | (made_up(plane, w + 3u) << 24);
                }
                reinterpret_cast<uint4*>((plane ? out_q : out_p) + w0)[0] = uint4{v[0], v[1], v[2], v[3]};
            }
        }
        __syncthreads();
        const uint64_t byte0 = ui.start_bit >> 3;
        const uint64_t left = avail_bytes - byte0;
        const uint32_t comp_len = uint32_t(left < (1ull << 28) ? left : (1ull << 28));
        const uint64_t rel = ui.stop_bit == ~0ull ? 0xFFFFFFFFull : ui.stop_bit - byte0 * 8u;
        // ONE decode, two texts: the codes are read once, every literal is stored and every match copied in both planes (round 4's
        // first version decoded the unit once per plane: 210 of the 360 ms of a 9.5 GB file)
        const uint32_t st = fqd::winf::inflate_stretch(ctx, sh, deflate + byte0, comp_len, uint32_t(ui.start_bit & 7u), uint32_t(rel < 0xFFFFFFFFull ? rel : 0xFFFFFFFFull),
                                                       out_p, kWindow, uint32_t(kWindow + ui.cap), tok, info, out_q);
        __syncthreads();
        if (threadIdx.x == 0) {
            UnitOut r;
            r.status = st;
            r.end_bit = st == fqd::winf::kOk ? byte0 * 8u + info[0] : 0;
            r.n = st == fqd::winf::kOk ? info[1] - kWindow : 0;
            r.how = st == fqd::winf::kOk ? info[2] : 0;
            result[2u * u] = r; result[2u * u + 1u] = r;                      // (one entry per plane, as the host has known them)
        }
        __syncthreads();
    }
}

// The symbol at place i of a unit, from its two planes.
__device__ __forceinline__ uint32_t symbol_of(const uint8_t* __restrict__ p, const uint8_t* __restrict__ q, uint64_t i)
{
    const uint32_t a = p[i], b = q[i];
    return a == b ? a : 256u + (a | (((a ^ b) - 1u) << 8));
}

// ---- 4. windows ------------------------------------------------------------------------------------------------------------------
// The 32 KiB of text that end with unit u, from its own last places and — where those were copied out of the window before it,
// or the unit is shorter than 32 KiB — from the window of unit u - 1.  Nearly every unit has such places (the instrument and
// flow-cell part of a FASTQ ID line is copied from record to record through the whole file), so the windows are a chain; but
// what a unit's window takes from the one before — WHICH place for every place — depends on the unit alone:
//   gz_window_maps_kernel   all units at once: map[u][k] = a byte, or 256 + the place of the window before that place k copies
//   gz_windows_chain_kernel one workgroup, the running window in LDS: next[k] = map byte, or prev[place]; the maps of the next
//                           unit are on their way while this one is looked up
// (One workgroup reading symbols and windows from HBM unit after unit took 25 us a unit: 120 of the first version's 168 ms per
// gigabyte of text.)
__global__ __launch_bounds__(256)
void gz_window_maps_kernel(const UnitIn* __restrict__ units, const UnitOut* __restrict__ result, uint32_t n_units,
                           const uint8_t* __restrict__ plane_p, const uint8_t* __restrict__ plane_q, uint16_t* __restrict__ maps /* n_units x kWindow */)
{
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint8_t* p = plane_p + units[u].at + kWindow;
        const uint8_t* q = plane_q + units[u].at + kWindow;
        const uint64_t n = result[2u * u].n;
        uint16_t* m = maps + uint64_t(u) * kWindow;
        for (uint32_t k = threadIdx.x; k < kWindow; k += 256u)
            // place k of the new window is text place n - kWindow + k of the unit, or, before the unit, place k + n of the old window
            m[k] = (n >= kWindow || k + n >= kWindow) ? uint16_t(symbol_of(p, q, n - kWindow + k)) : uint16_t(256u + k + uint32_t(n));
    }
}
__global__ __launch_bounds__(1024)
void gz_windows_chain_kernel(const uint16_t* __restrict__ maps, uint32_t n_units, uint8_t* __restrict__ windows /* (n_units + 1) x kWindow; [0] is given */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t win[];          // two windows: the one before, the one being made
    constexpr uint32_t kPer = kWindow / 1024u / 8u;                        // 4 x eight places a lane
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t * 16u; k < kWindow; k += 1024u * 16u) reinterpret_cast<uint4*>(win)[k / 16u] = reinterpret_cast<const uint4*>(windows)[k / 16u];
    uint4 m[kPer], ahead[kPer];
    if (n_units) for (uint32_t i = 0; i < kPer; ++i) ahead[i] = reinterpret_cast<const uint4*>(maps)[i * 1024u + t];
    __syncthreads();
    for (uint32_t u = 0; u < n_units; ++u) {
        const uint8_t* prev = win + (u & 1u) * kWindow;
        uint8_t* next = win + ((u + 1u) & 1u) * kWindow;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) m[i] = ahead[i];
        if (u + 1u < n_units) {
            const uint4* nm = reinterpret_cast<const uint4*>(maps + uint64_t(u + 1u) * kWindow);
#pragma unroll
            for (uint32_t i = 0; i < kPer; ++i) ahead[i] = nm[i * 1024u + t];
        }
        uint8_t* gnext = windows + uint64_t(u + 1u) * kWindow;
#pragma unroll
        for (uint32_t i = 0; i < kPer; ++i) {
            const uint32_t w[4] = {m[i].x, m[i].y, m[i].z, m[i].w};
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t sy = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t byte = sy < 256u ? sy : prev[sy - 256u];
                if (k < 4) lo |= byte << (8 * k); else hi |= byte << (8 * (k - 4));
            }
            const uint32_t at = (i * 1024u + t) * 8u;
            reinterpret_cast<uint2*>(next)[at / 8u] = uint2{lo, hi};
            reinterpret_cast<uint2*>(gnext)[at / 8u] = uint2{lo, hi};
        }
        __syncthreads();
    }
}

// ---- 5. bytes ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void gz_resolve_kernel(const UnitIn* __restrict__ units, const UnitOut* __restrict__ result, const uint64_t* __restrict__ text_at, uint32_t n_units,
                       const uint8_t* __restrict__ plane_p, const uint8_t* __restrict__ plane_q, const uint8_t* __restrict__ windows, uint8_t* __restrict__ text)
{
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint8_t* p = plane_p + units[u].at + kWindow;                   // 16-byte aligned
        const uint8_t* q = plane_q + units[u].at + kWindow;
        const uint8_t* win = windows + uint64_t(u) * kWindow;
        uint8_t* dst = text + text_at[u];
        const uint64_t n = result[2u * u].n;
        for (uint64_t g = threadIdx.x; g < (n >> 4); g += 256u) {
            const uint4 a = reinterpret_cast<const uint4*>(p)[g], b = reinterpret_cast<const uint4*>(q)[g];
            uint32_t wa[4] = {a.x, a.y, a.z, a.w};
            const uint32_t wb[4] = {b.x, b.y, b.z, b.w};
            if ((wa[0] ^ wb[0]) | (wa[1] ^ wb[1]) | (wa[2] ^ wb[2]) | (wa[3] ^ wb[3])) {       // some byte came out of the window
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t x = (wa[k >> 2] >> (8 * (k & 3))) & 0xFFu, y = (wb[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    if (x != y) wa[k >> 2] = (wa[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | (uint32_t(win[x | (((x ^ y) - 1u) << 8)]) << (8 * (k & 3)));
                }
            }
            uint8_t* d = dst + 16u * g;                                       // (dst has whatever alignment the units before left it)
            if ((reinterpret_cast<uintptr_t>(d) & 3u) == 0) { uint32_t* d4 = reinterpret_cast<uint32_t*>(d); d4[0] = wa[0]; d4[1] = wa[1]; d4[2] = wa[2]; d4[3] = wa[3]; }
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k) d[k] = uint8_t(wa[k >> 2] >> (8 * (k & 3)));
            }
        }
        for (uint64_t i = (n & ~15ull) + threadIdx.x; i < n; i += 256u) { const uint32_t s = symbol_of(p, q, i); dst[i] = s < 256u ? uint8_t(s) : win[s - 256u]; }
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

struct View {                                                                // a piece of the engine's scratch (below)
    void* p = nullptr;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

} // namespace

extern "C" {

int fqd_gunzip(fqd_engine* e, const uint8_t* deflate, uint64_t avail_bytes, uint8_t* text, uint64_t text_cap,
               uint64_t* text_bytes, uint64_t* deflate_bytes, uint32_t* crc32, int32_t* ok)
{
    return fqd_gunzip_arriving(e, deflate, avail_bytes, nullptr, text, text_cap, text_bytes, deflate_bytes, crc32, ok);
}

int fqd_gunzip_arriving(fqd_engine* e, const uint8_t* deflate, uint64_t avail_bytes, const volatile uint64_t* arrived, uint8_t* text, uint64_t text_cap,
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
    const auto t_begin = std::chrono::steady_clock::now();
#define GZ_TRACE(...) do { if (trace) { (void)hipStreamSynchronize(stream); std::fprintf(stderr, "[gunzip %7.2f ms] ", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); \
    std::fprintf(stderr, __VA_ARGS__); std::fputc('\n', stderr); std::fflush(stderr); } } while (0)
    BitIn in;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(deflate);
    in.words = reinterpret_cast<const uint64_t*>(addr & ~uintptr_t(7));
    in.lead = uint64_t(addr & 7u) * 8u;
    in.nbits = avail_bytes * 8u;

    // units: small enough that there are several per wave slot of the chip, large enough to hold a block start more often than
    // not (zlib ends a block every 16 K codes: some 20-50 KB packed).  (Measured and replaced on the way here, DESIGN §3b: one
    // serial decoder per wave writing 16-bit symbols, all lanes running it alike — 4.6 GB/s of text; the same, one decoder per
    // LANE with its tables in HBM — 2.6 GB/s: lanes of a wave that copy matches of different lengths wait for the longest.)
    // Large files: some 16 K units of at most 256 KiB — a unit that has to be decoded again is ONE wave's work (11 MB/s of packed bytes
    // when it runs alone), the chip's 4096 decoders are filled in whole rounds, and the windows' chain costs 2 us a unit.
    uint64_t unit_bytes = std::min<uint64_t>(256u << 10, std::max<uint64_t>(64u << 10, (avail_bytes / 16384u + 4095u) & ~uint64_t(4095)));
    if (const char* v = std::getenv("FQD_GUNZIP_UNIT_KB")) { const long kb = std::atol(v); if (kb > 0) unit_bytes = uint64_t(kb) << 10; }
    uint64_t ratio = 8;                                                      // symbols of room per compressed byte (FASTQ packs 3-6 fold)
    if (const char* v = std::getenv("FQD_GUNZIP_RATIO")) { const long r = std::atol(v); if (r > 0) ratio = uint64_t(r); }
    const uint64_t n_nominal = (avail_bytes + unit_bytes - 1) / unit_bytes;

    // ---- 2-5 in batches of units whose two planes fit the scratch: by default up to two rounds of the chip's decoders (a wave a unit)
    const uint64_t one_more = uint64_t(kWindow) + ((2u * unit_bytes * ratio + 1024u + 15u) & ~uint64_t(15)) + 64u;     // the room of a unit of one nominal unit
    const uint32_t decoders = uint32_t(std::min<uint64_t>(n_nominal, uint64_t(n_cu) * 16u));           // a wave a unit; 10 KB of LDS, <= 128 VGPRs: sixteen waves a CU
    uint64_t plane_bytes = std::min<uint64_t>(uint64_t(8) << 30, (2u * decoders + 64u) * one_more);   // per plane: two rounds if 8 GiB hold them (two back to back take less than twice one)
    if (const char* v = std::getenv("FQD_GUNZIP_SCRATCH_MB")) { const long mb = std::atol(v); if (mb > 0) plane_bytes = (uint64_t(mb) << 20) / 2; }
    auto room_of = [](const UnitIn& x) { return uint64_t(kWindow) + x.cap + 64u; };          // made-up window, text, slack (a multiple of 16)
    plane_bytes = std::min(plane_bytes, n_nominal * one_more + 64u * one_more);                       // (never more than all units need)
    // ONE piece of the engine's scratch holds everything the call needs on the device, and stays with the engine: a call that
    // frees gigabytes hands them to the driver for clearing, and the process's next hipMalloc — the dedup engine's key store, the
    // other file's planes — waits for that (configs[4] shape, two files at once: a second lost in whatever stage came next).
    View d_start, d_p, d_q, d_units, d_result, d_text_at, d_windows, d_counter, d_tokens, d_maps, d_ru, d_rr, d_shift, d_raw;
    uint64_t nb_cap = 0;                                                      // units a batch can hold
    auto place = [&]() -> int {
        nb_cap = plane_bytes / one_more + 32u;
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at_o = o; o = (o + bytes + 255u) & ~size_t(255); return at_o; };
        const size_t o_p = take(plane_bytes + 64), o_q = take(plane_bytes + 64);
        const size_t o_tok = take(size_t(decoders) * fqd::winf::kTokenRoom * sizeof(fqd::winf::Token));
        const size_t o_maps = take(nb_cap * kWindow * 2u), o_win = take((nb_cap + 1u) * kWindow);
        const size_t o_units = take(nb_cap * sizeof(UnitIn)), o_res = take(2u * nb_cap * sizeof(UnitOut)), o_at = take(nb_cap * 8u);
        const size_t o_ru = take(nb_cap * sizeof(UnitIn)), o_rr = take(2u * nb_cap * sizeof(UnitOut));
        const size_t o_cnt = take(64), o_start = take(n_nominal * 8u), o_shift = take(8 * 32 * 4), o_raw = take((text_cap / kSlice + 2u) * 4u);
        void* base = nullptr;
        const int rc_s = fqd_internal_scratch(e, 1, o, &base);
        if (rc_s != FQD_OK) return rc_s;
        uint8_t* b8 = static_cast<uint8_t*>(base);
        d_p.p = b8 + o_p; d_q.p = b8 + o_q; d_tokens.p = b8 + o_tok; d_maps.p = b8 + o_maps; d_windows.p = b8 + o_win;
        d_units.p = b8 + o_units; d_result.p = b8 + o_res; d_text_at.p = b8 + o_at; d_ru.p = b8 + o_ru; d_rr.p = b8 + o_rr;
        d_counter.p = b8 + o_cnt; d_start.p = b8 + o_start; d_shift.p = b8 + o_shift; d_raw.p = b8 + o_raw;
        return FQD_OK;
    };
    { const int rc_p = place(); if (rc_p != FQD_OK) return rc_p; }
    // ---- 1. starts, and the units they make — of what has ARRIVED (fqd_gunzip_arriving: the file is still being copied to HBM
    // by another thread of the caller, which raises *arrived as its copies complete; everything below works on what is there
    // and waits for the rest, so that block starts are found and units decoded under the read)
    auto have_now = [&]() -> uint64_t { if (!arrived) return avail_bytes; const uint64_t a = *arrived; return a < avail_bytes ? a : (a == ~0ull ? ~0ull : avail_bytes); };
    std::vector<uint64_t> start(n_nominal, ~0ull);
    std::vector<UnitIn> units;
    uint64_t searched = 0, built = 0;                                         // nominal units looked at for a start / turned into units
    bool hopeless = false;
    auto take_in = [&](uint64_t have) -> int {
        // a unit is looked at when it and what the looks read beyond a position (a block header: some hundred bytes) have arrived
        const uint64_t can = have == avail_bytes ? n_nominal : (have > 4096u ? (have - 4096u) / unit_bytes : 0u);
        if (can > searched) {
            hipLaunchKernelGGL(gz_find_starts_kernel, dim3(uint32_t(std::min<uint64_t>(can - searched, uint64_t(n_cu) * 16u))), dim3(kWave), 0, stream,
                               in, unit_bytes * 8u, searched, can, d_start.as<uint64_t>());
            GZ_TRY(e, hipGetLastError());
            GZ_TRY(e, hipMemcpyAsync(start.data() + searched, d_start.as<uint64_t>() + searched, (can - searched) * 8, hipMemcpyDeviceToHost, stream));
            GZ_TRY(e, hipStreamSynchronize(stream));
            GZ_TRACE("%llu of %llu bytes there: block starts of units %llu to %llu looked for", (unsigned long long)have, (unsigned long long)avail_bytes,
                     (unsigned long long)searched, (unsigned long long)can);
            searched = can;
        }
        while (built < searched) {
            if (start[built] == ~0ull) { ++built; continue; }
            uint64_t next = built + 1;
            while (next < searched && start[next] == ~0ull) ++next;          // a unit without a start belongs to the one before it
            // a long stretch without a dynamic block's start (stored or fixed blocks only: nothing a sequencer or gzip writes for FASTQ)
            // would be ONE wave's work at some 30 MB/s: beyond 32 MiB the host reader is the faster way, and the caller takes it
            if ((next - built) * unit_bytes > (uint64_t(32) << 20)) {
                GZ_TRACE("%llu bytes without a block start that can be guessed: left to the host reader", (unsigned long long)((next - built) * unit_bytes));
                hopeless = true; return FQD_OK;
            }
            if (next == searched && searched < n_nominal) break;             // where this unit stops is not known yet
            UnitIn x;
            x.start_bit = start[built];
            x.stop_bit = next < n_nominal ? next * unit_bytes * 8u : ~0ull;
            // room: `ratio` bytes of text per compressed byte of the unit's stretch — and of one unit more: a unit that is decoded again
            // from the boundary the unit before it REALLY ended at (a member that ends early in that unit) starts that much earlier
            x.cap = ((next - built + 1u) * unit_bytes * ratio + 1024u + 15u) & ~uint64_t(15);
            x.at = 0;
            units.push_back(x);
            built = next;
        }
        return FQD_OK;
    };

    GZ_TRACE("%llu bytes, %llu units of %llu bytes; scratch: 2 planes of %llu bytes, %u decoders' tokens", (unsigned long long)avail_bytes, (unsigned long long)n_nominal,
             (unsigned long long)unit_bytes, (unsigned long long)plane_bytes, decoders);
    struct Member { uint64_t text_from, text_to, deflate_end; uint32_t crc, isize; };
    std::vector<Member> members;
    uint64_t member_from = 0;
    std::vector<uint8_t> carry(kWindow, 0);                                   // the window before the next batch's first unit
    uint64_t total = 0, expect_start = 0, repairs = 0;
    bool final_seen = false, good = true;
    std::vector<UnitOut> result;
    std::vector<uint64_t> text_at;
    size_t at = 0;
    uint64_t have = 0;                                                        // bytes of the stream a batch may count on
    bool member_end_before = false;                                           // the unit the chain took last ended a member
    bool all_there = false;
    // What follows a member's final block: its trailer (CRC-32, ISIZE) and the end of the file, or the header of another member (RFC 1952).
    struct Trailer { uint32_t crc = 0, isize = 0; int kind = 0; /* 0 neither, 1 the file ends, 2 another member */ uint64_t next_start = 0; };
    std::unordered_map<uint64_t, Trailer> trailers;                          // by byte offset: looked at twice (below), fetched once
    auto trailer_at = [&](uint64_t end_bit, Trailer& t) -> int {
        const uint64_t trailer = (end_bit + 7u) / 8u;
        const auto it = trailers.find(trailer);
        if (it != trailers.end()) { t = it->second; return FQD_OK; }
        t = Trailer();
        if (trailer + 8u <= avail_bytes) {
            uint8_t head[8 + 1024];
            const size_t got = size_t(std::min<uint64_t>(sizeof head, avail_bytes - trailer));
            GZ_TRY(e, hipMemcpyAsync(head, deflate + trailer, got, hipMemcpyDeviceToHost, stream));
            GZ_TRY(e, hipStreamSynchronize(stream));
            t.crc = head[0] | (uint32_t(head[1]) << 8) | (uint32_t(head[2]) << 16) | (uint32_t(head[3]) << 24);
            t.isize = head[4] | (uint32_t(head[5]) << 8) | (uint32_t(head[6]) << 16) | (uint32_t(head[7]) << 24);
            if (trailer + 8u == avail_bytes) t.kind = 1;
            else {
                const uint8_t* h = head + 8; const size_t hn = got - 8;
                size_t at_h = 10;
                bool fine = hn >= 18 && h[0] == 31 && h[1] == 139 && h[2] == 8 && !(h[3] & 0xE0);
                if (fine && (h[3] & 4)) { if (at_h + 2 > hn) fine = false; else at_h += 2 + (h[at_h] | (size_t(h[at_h + 1]) << 8)); }
                if (fine && (h[3] & 8)) { while (at_h < hn && h[at_h]) ++at_h; ++at_h; }
                if (fine && (h[3] & 16)) { while (at_h < hn && h[at_h]) ++at_h; ++at_h; }
                if (fine && (h[3] & 2)) at_h += 2;
                if (fine && at_h + 2 <= hn) { t.kind = 2; t.next_start = (trailer + 8u + at_h) * 8u; }
            }
        }
        trailers[trailer] = t;
        return FQD_OK;
    };
    auto decode = [&](uint32_t first, uint32_t count) -> int {              // units [first, first + count) of the batch, both planes
        GZ_TRY(e, hipMemsetAsync(d_counter.p, 0, 64, stream));
        hipLaunchKernelGGL(gz_decode_planes_kernel, dim3(std::min<uint32_t>(count, decoders)), dim3(kWave), 0, stream,
                           deflate, have, d_units.as<const UnitIn>() + first, count, d_p.as<uint8_t>(), d_q.as<uint8_t>(),
                           d_tokens.as<fqd::winf::Token>(), d_result.as<UnitOut>() + 2u * first, d_counter.as<uint32_t>());
        GZ_TRY(e, hipGetLastError());
        GZ_TRY(e, hipMemcpyAsync(result.data() + 2u * first, d_result.as<UnitOut>() + 2u * first, 2u * count * sizeof(UnitOut), hipMemcpyDeviceToHost, stream));
        GZ_TRY(e, hipStreamSynchronize(stream));
        return FQD_OK;
    };
    // what lies behind a final block (trailer, perhaps a header of up to a kilobyte) can be looked at
    auto trailer_there = [&](uint64_t end_bit) { return all_there || (end_bit + 7u) / 8u + 8u + 1024u <= have; };
    uint64_t wait_beyond = 0;                                                // nothing to do until more than this has arrived
    while (good && !final_seen) {
        have = have_now();
        if (have == ~0ull) { GZ_TRACE("the caller says the rest of the file will not come"); good = false; break; }
        all_there = have == avail_bytes;
        if (!all_there && have <= wait_beyond) { ::usleep(100); continue; }
        int rc = take_in(have);
        if (rc) return rc;
        if (hopeless) return FQD_OK;
        // units that can be decoded with what is there: a unit reads on to the first block boundary behind its stop — a unit of slack
        size_t ready = at;
        while (ready < units.size() && (all_there || (units[ready].stop_bit != ~0ull && units[ready].stop_bit / 8u + unit_bytes <= have))) ++ready;
        if (ready == at) {
            if (!all_there) { wait_beyond = have; continue; }
            if (expect_start / 8u + 2u >= avail_bytes) break;                // (no final block's end was seen: not ok)
            // members go on behind the last unit that had a start: the rest of the file as one more unit, from where the chain stands
            UnitIn x;
            x.start_bit = expect_start; x.stop_bit = ~0ull; x.at = 0;
            x.cap = ((avail_bytes - expect_start / 8u) * ratio + 1024u + 15u) & ~uint64_t(15);
            if ((avail_bytes - expect_start / 8u) > (uint64_t(32) << 20)) return FQD_OK;
            units.push_back(x);
            ready = units.size();
        }
        // a batch: what the planes hold — less some room for units put in between (below) — and, when there are that many, whole
        // rounds of the decoders (a wave a unit): a round that fills a sixth of the chip takes as long as one that fills it
        const uint64_t spare = std::min<uint64_t>(plane_bytes / 8u, 24u * one_more);
        size_t hi = at; uint64_t used = 0;
        while (hi < ready && used + room_of(units[hi]) <= plane_bytes - spare) { units[hi].at = used; used += room_of(units[hi]); ++hi; }
        if (const size_t round = decoders; hi - at > round && (hi < ready || !all_there)) {
            hi = at + (hi - at) / round * round;
            used = units[hi - 1].at + room_of(units[hi - 1]);
        }
        if (hi == at) {
            // one unit that needs more than a plane holds (FQD_GUNZIP_SCRATCH_MB set small, a long stretch): the planes grow
            plane_bytes = room_of(units[at]) + room_of(units[at]) / 4u;
            GZ_TRY(e, hipStreamSynchronize(stream));
            if ((rc = place())) return rc;
            continue;
        }
        uint32_t nb = uint32_t(hi - at);                                      // (up to 24 units may be put in between, below)
        if (nb + 24u > nb_cap) { hi = at + size_t(nb_cap - 24u); nb = uint32_t(hi - at); used = units[hi - 1].at + room_of(units[hi - 1]); }   // (cannot be: a unit's room is at least one_more)
        GZ_TRY(e, hipMemcpyAsync(d_units.p, units.data() + at, nb * sizeof(UnitIn), hipMemcpyHostToDevice, stream));
        result.resize(2u * nb);
        GZ_TRACE("batch of %u units queued for decoding (%llu bytes a plane)", nb, (unsigned long long)used);
        rc = decode(0, nb);
        if (rc) return rc;
        GZ_TRACE("decoded: first unit status %u, %llu bytes, end bit %llu", result[0].status, (unsigned long long)result[0].n, (unsigned long long)result[0].end_bit);
        // ---- 3. the chain
        // Units that do not start where the unit before them ended — a guess that did not hold (a header-like stretch of bits inside a
        // block: a few per gigabyte), or the unit behind a member's end, whose next member starts where the trailer and a header
        // say — are decoded again from there.  One wave decodes some 30 MB/s of packed bytes, so such units are first collected over
        // the whole batch, on the assumption that where a unit ENDS does not change when it is decoded again (both starts are block
        // boundaries of one stream), and decoded again in ONE launch; the chain below then checks every link and mends, one unit
        // at a time, what that assumption missed.  (One launch per unit: 13 ms each, 90 of the 188 ms of a 12-member file.)
        {
            // A unit behind a member's end is not decoded again as a whole: the stretch from the new member's first block to the unit's
            // nominal start becomes a unit of its own, PUT IN before it (room for it at the planes' end), and the unit keeps what it
            // decoded from its own guess — on average a quarter of the work, and it grows with the unit no more.
            std::vector<uint32_t> again;                                     // positions in the batch
            std::vector<UnitIn> put_in;                                      // .cap == 0: decoded again where it is
            uint64_t expect = expect_start, spare_at = used;
            bool behind_member_end = member_end_before;
            for (uint32_t k = 0; k < nb; ++k) {
                UnitIn& x = units[at + k];
                if (x.start_bit != expect) {
                    const uint64_t nominal = at + k > 0 ? units[at + k - 1].stop_bit : ~0ull;      // where this unit's share of the stream begins
                    UnitIn y{0, 0, 0, 0};
                    if (behind_member_end && nominal != ~0ull && expect < nominal && x.start_bit >= nominal && put_in.size() < 24u &&
                        spare_at + one_more <= plane_bytes && std::count_if(put_in.begin(), put_in.end(), [](const UnitIn& u) { return u.cap != 0; }) < 24) {
                        y.start_bit = expect; y.stop_bit = nominal; y.at = spare_at; y.cap = one_more - kWindow - 64u;
                        spare_at += one_more;
                    } else x.start_bit = expect;
                    again.push_back(k); put_in.push_back(y);
                }
                const UnitOut& r = result[2u * k]; const UnitOut& r2 = result[2u * k + 1u];
                if (r.status != fqd::winf::kOk || r2.status != fqd::winf::kOk || r.end_bit != r2.end_bit || r.how != r2.how) break;   // garbage from a wrong guess: the chain takes over here
                if (r.how == 2u) {
                    if (!trailer_there(r.end_bit)) break;
                    Trailer t;
                    if ((rc = trailer_at(r.end_bit, t))) return rc;
                    if (t.kind != 2) break;
                    expect = t.next_start; behind_member_end = true;
                } else { expect = r.end_bit; behind_member_end = false; }
            }
            if (!again.empty()) {
                repairs += again.size();
                if (repairs > 64u + units.size() / 16u) { GZ_TRACE("%llu units do not start where the one before them ended: giving up", (unsigned long long)repairs); good = false; break; }
                std::vector<UnitIn> ru(again.size());
                for (size_t i = 0; i < again.size(); ++i) ru[i] = put_in[i].cap ? put_in[i] : units[at + again[i]];
                std::vector<UnitOut> rr(2u * again.size());
                GZ_TRY(e, hipMemcpyAsync(d_ru.p, ru.data(), ru.size() * sizeof(UnitIn), hipMemcpyHostToDevice, stream));
                GZ_TRY(e, hipMemsetAsync(d_counter.p, 0, 64, stream));
                hipLaunchKernelGGL(gz_decode_planes_kernel, dim3(std::min<uint32_t>(uint32_t(ru.size()), decoders)), dim3(kWave), 0, stream,
                                   deflate, have, d_ru.as<const UnitIn>(), uint32_t(ru.size()), d_p.as<uint8_t>(), d_q.as<uint8_t>(),
                                   d_tokens.as<fqd::winf::Token>(), d_rr.as<UnitOut>(), d_counter.as<uint32_t>());
                GZ_TRY(e, hipGetLastError());
                GZ_TRY(e, hipMemcpyAsync(rr.data(), d_rr.p, rr.size() * sizeof(UnitOut), hipMemcpyDeviceToHost, stream));
                GZ_TRY(e, hipStreamSynchronize(stream));
                uint32_t more = 0;
                for (size_t i = again.size(); i-- > 0;) {                     // from the back: positions before it stay what they are
                    const uint32_t k = again[i];
                    if (put_in[i].cap) {
                        units.insert(units.begin() + std::ptrdiff_t(at + k), put_in[i]);
                        result.insert(result.begin() + std::ptrdiff_t(2u * k), {rr[2u * i], rr[2u * i + 1u]});
                        ++more;
                    } else { result[2u * k] = rr[2u * i]; result[2u * k + 1u] = rr[2u * i + 1u]; }
                }
                nb += more; hi += more;
                GZ_TRY(e, hipMemcpyAsync(d_units.p, units.data() + at, nb * sizeof(UnitIn), hipMemcpyHostToDevice, stream));
                GZ_TRY(e, hipMemcpyAsync(d_result.p, result.data(), 2u * nb * sizeof(UnitOut), hipMemcpyHostToDevice, stream));
                GZ_TRACE("%zu units decoded again in one launch, each from where the unit before it ended (%u of them the short stretch behind a member's end, put in as a unit)",
                         again.size(), more);
            }
        }
        text_at.resize(nb);
        uint32_t live = 0;
        bool put_off = false;                                                 // a unit that needs bytes that have not arrived: it and what follows wait
        for (uint32_t k = 0; k < nb && good && !final_seen; ++k) {
            UnitIn& x = units[at + k];
            if (x.start_bit != expect_start) {
                // what the pass above could not know: the unit before this one ended elsewhere when it was decoded again
                if (++repairs > 64u + units.size() / 16u) { good = false; break; }       // (damage, not bad luck)
                x.start_bit = expect_start;
                GZ_TRY(e, hipMemcpyAsync(d_units.as<UnitIn>() + k, &x, sizeof(UnitIn), hipMemcpyHostToDevice, stream));
                if ((rc = decode(k, 1u))) return rc;
                GZ_TRACE("unit %zu decoded again from bit %llu: status %u, %llu bytes", at + k, (unsigned long long)expect_start, result[2u * k].status, (unsigned long long)result[2u * k].n);
            }
            const UnitOut& r = result[2u * k]; const UnitOut& r2 = result[2u * k + 1u];
            if (r.status != fqd::winf::kOk || r2.status != fqd::winf::kOk || r.end_bit != r2.end_bit || r.n != r2.n || r.how != r2.how) {
                if (!all_there) { put_off = true; break; }                    // (it may have run into what has not arrived: once more when there is more)
                GZ_TRACE("unit %zu (from bit %llu, room %llu): status %u / %u, %llu / %llu bytes: giving up", at + k, (unsigned long long)x.start_bit, (unsigned long long)x.cap,
                         r.status, r2.status, (unsigned long long)r.n, (unsigned long long)r2.n);
                good = false; break;
            }
            if (r.how == 2u && !trailer_there(r.end_bit)) { put_off = true; break; }
            if (total + r.n > text_cap) { GZ_TRACE("the text outgrows the room given (%llu bytes)", (unsigned long long)text_cap); good = false; break; }
            text_at[k] = total; total += r.n;
            expect_start = r.end_bit;
            live = k + 1;
            member_end_before = r.how == 2u;
            if (r.how == 2u) {
                // the member's final block ended here: its trailer follows at the next byte, and — `cat a.gz b.gz`, or a writer that
                // starts a member every so often — perhaps another member, whose first block the next unit is then decoded from
                Trailer t;
                if ((rc = trailer_at(r.end_bit, t))) return rc;
                if (t.kind == 0) { GZ_TRACE("no trailer, or no member header, behind the final block that ends at bit %llu", (unsigned long long)r.end_bit); good = false; break; }
                Member m;
                m.text_from = member_from; m.text_to = total; m.deflate_end = (r.end_bit + 7u) / 8u;
                m.crc = t.crc; m.isize = t.isize;
                members.push_back(m);
                member_from = total;
                if (t.kind == 1) { final_seen = true; break; }
                expect_start = t.next_start;
            }
        }
        if (!good) break;
        if (put_off) { GZ_TRACE("unit %zu waits for bytes that have not arrived (%llu there)", at + live, (unsigned long long)have); wait_beyond = have; }
        if (live == 0) continue;
        // ---- 4, 5
        GZ_TRY(e, hipMemcpyAsync(d_text_at.p, text_at.data(), live * 8, hipMemcpyHostToDevice, stream));
        GZ_TRY(e, hipMemcpyAsync(d_windows.p, carry.data(), kWindow, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gz_window_maps_kernel, dim3(std::min<uint32_t>(live, uint32_t(n_cu) * 8u)), dim3(256), 0, stream, d_units.as<const UnitIn>(),
                           d_result.as<const UnitOut>(), live, d_p.as<const uint8_t>(), d_q.as<const uint8_t>(), d_maps.as<uint16_t>());
        GZ_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(gz_windows_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(2u * kWindow)));
        hipLaunchKernelGGL(gz_windows_chain_kernel, dim3(1), dim3(1024), 2u * kWindow, stream, d_maps.as<const uint16_t>(), live, d_windows.as<uint8_t>());
        GZ_TRY(e, hipGetLastError());
        GZ_TRACE("windows of %u units made", live);
        hipLaunchKernelGGL(gz_resolve_kernel, dim3(std::min<uint32_t>(live, uint32_t(n_cu) * 8u)), dim3(256), 0, stream,
                           d_units.as<const UnitIn>(), d_result.as<const UnitOut>(), d_text_at.as<const uint64_t>(), live,
                           d_p.as<const uint8_t>(), d_q.as<const uint8_t>(), d_windows.as<const uint8_t>(), text);
        GZ_TRY(e, hipGetLastError());
        GZ_TRY(e, hipMemcpyAsync(carry.data(), d_windows.as<uint8_t>() + uint64_t(live) * kWindow, kWindow, hipMemcpyDeviceToHost, stream));
        GZ_TRY(e, hipStreamSynchronize(stream));
        GZ_TRACE("windows and bytes of %u units done, %llu bytes of text so far", live, (unsigned long long)total);
        at = at + live < hi && final_seen ? hi : at + live;                  // (units of the batch behind a repair that were not reached are decoded with the next batch)
    }
    if (!good || !final_seen) return FQD_OK;                                  // *ok stays 0: the caller reads the file the host way

    // ---- CRC-32 and length of every member's text against its trailer
    if (members.empty()) return FQD_OK;
    uint32_t crc = 0;
    {
        uint32_t shift[8][32];
        Mat m = mat_one_zero_byte();
        for (int k = 0; k < 8; ++k) m = mat_square(m);                        // 256 zero bytes
        for (int k = 0; k < 8; ++k) { std::memcpy(shift[k], m.col, sizeof m.col); m = mat_square(m); }
        const Mat slice_mat = m;                                              // 65536 zero bytes
        GZ_TRY(e, hipMemcpyAsync(d_shift.p, shift, sizeof shift, hipMemcpyHostToDevice, stream));
        std::vector<uint32_t> raw;
        for (const Member& mb : members) {
            const uint64_t len = mb.text_to - mb.text_from;
            crc = 0;
            if (len) {
                const uint64_t n_slices = (len + kSlice - 1) / kSlice;
                hipLaunchKernelGGL(gz_crc_kernel, dim3(uint32_t(std::min<uint64_t>(n_slices, uint64_t(n_cu) * 8u))), dim3(kCrcThreads), 0, stream,
                                   static_cast<const uint8_t*>(text) + mb.text_from, len, n_slices, d_shift.as<const uint32_t>(), d_raw.as<uint32_t>());
                GZ_TRY(e, hipGetLastError());
                raw.resize(n_slices);
                GZ_TRY(e, hipMemcpyAsync(raw.data(), d_raw.p, n_slices * 4, hipMemcpyDeviceToHost, stream));
                GZ_TRY(e, hipStreamSynchronize(stream));
                // raw register of the member's text started from 0: full slices fold with the 64 KiB matrix, the short last one with its own length
                uint32_t reg = 0;
                for (uint64_t sl = 0; sl < n_slices; ++sl) {
                    const uint64_t l = sl + 1 < n_slices ? kSlice : len - sl * kSlice;
                    reg = (l == kSlice ? mat_apply(slice_mat, reg) : advance_zero_bytes(reg, l)) ^ raw[sl];
                }
                crc = reg ^ advance_zero_bytes(0xFFFFFFFFu, len) ^ 0xFFFFFFFFu;   // the same register started from all ones, then inverted: zlib's CRC-32
            }
            if (crc != mb.crc || uint32_t(len) != mb.isize) { GZ_TRACE("a member's CRC-32 or length is not its trailer's"); return FQD_OK; }   // *ok stays 0
        }
    }
    GZ_TRACE("%zu member(s): CRC-32 and length as the trailers say", members.size());
    *text_bytes = total;
    *deflate_bytes = members.back().deflate_end;
    *crc32 = crc;
    *ok = 1;
    return FQD_OK;
}

} // extern "C"
