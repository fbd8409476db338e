// enc_shape_probe — does any feature of the staged encoder's STRUCTURE cost it bandwidth?
// tools/mix_probe.hip (the fixed one, profiles/r04_mix_probe.jsonl): a bare grid-stride kernel moves the encoder's traffic mix
// (150 B in, 72 B out per read) at 4.9-5.05 TB/s; the encoder itself reaches 5.3-5.7, and round 4's A/Bs say its arithmetic is
// not what bounds it (no packing at all: 0.15 ms).  (This header first quoted 7.3-7.5 TB/s for the bare kernel: the
// mix probe's first run, whose compiler had removed half the loads.)  The
// variants below move exactly the encoder's bytes — a workgroup takes tiles of 256 reads = 38400 contiguous bytes, writes
// 16 KiB of "keys" and 2 KiB of "hashes" per tile — and add its structure one piece at a time:
//   0  registers only: 16-byte loads (BATCH in flight per lane), xor, 16-byte key stores, 8-byte hash stores; no LDS
//   1  + staging: loads -> LDS -> barrier -> key stores out of LDS -> barrier              (the encoder's skeleton)
//   2  = 1 with every load of the tile in flight at once (BATCH 10)
//   3  = 1 with the NEXT tile's loads issued before this tile's stores (registers held across the barriers)
//   4  two tiles per workgroup in LDS (half the workgroups per CU), loads of one under the stores of the other
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/enc_shape_probe tools/enc_shape_probe.hip ; run: tools/enc_shape_probe [Mreads]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kR = 256, kL = 150, kTile = kR * kL, kN16 = kTile / 16;      // 38400 B = 2400 chunks of 16 B
constexpr uint32_t kPer = (kN16 + kR - 1) / kR;                                 // 10 chunks per lane (the last partly)

template <int BATCH>
__device__ __forceinline__ void stage(const u32x4* __restrict__ src, u32x4* dst)
{
    for (uint32_t c0 = threadIdx.x; c0 < kN16; c0 += uint32_t(BATCH) * kR) {
        u32x4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const uint32_t c = c0 + uint32_t(k) * kR; v[k] = __builtin_nontemporal_load(&src[c < kN16 ? c : kN16 - 1u]); }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const uint32_t c = c0 + uint32_t(k) * kR; if (c < kN16) dst[c] = v[k]; }
    }
}
__device__ __forceinline__ void keys_out(const uint64_t* lds64, u64x2* __restrict__ gout, uint64_t* __restrict__ hout)
{
    // a wave streams its 64 rows of 8 words: 4 x (two adjacent words out of LDS, one 16-byte store); rows start where the reads do
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t y = lane; y < 256u; y += 64u) {
        const uint32_t rr = y >> 2, k2 = (y & 3u) << 1;
        const uint64_t* row = lds64 + (((wave * 64u + rr) * kL + 11u) >> 3) + k2;
        const u64x2 v = {row[0], row[1]};
        __builtin_nontemporal_store(v, &gout[wave * 256u + y]);
    }
    hout[threadIdx.x] = lds64[threadIdx.x];
}

template <int VARIANT>
__global__ __launch_bounds__(256) void enc_shape(const uint8_t* __restrict__ in, uint64_t n_tiles, u64x2* __restrict__ keys, uint64_t* __restrict__ hashes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    u32x4* lds16 = reinterpret_cast<u32x4*>(lds);
    const uint64_t* lds64 = reinterpret_cast<const uint64_t*>(lds);
    if (VARIANT == 0) {
        for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const u32x4* src = reinterpret_cast<const u32x4*>(in + tile * kTile);
            u32x4 acc = {0, 0, 0, 0};
            for (uint32_t c0 = threadIdx.x; c0 < kN16; c0 += 5u * kR) {
                u32x4 v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) { const uint32_t c = c0 + uint32_t(k) * kR; v[k] = __builtin_nontemporal_load(&src[c < kN16 ? c : kN16 - 1u]); }
#pragma unroll
                for (int k = 0; k < 5; ++k) acc ^= v[k];
            }
            u64x2* gout = keys + tile * 1024u;                               // 256 keys x 64 B = 1024 x 16 B
            const u64x2 kv = {uint64_t(acc.x) | (uint64_t(acc.y) << 32), uint64_t(acc.z) | (uint64_t(acc.w) << 32)};
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(kv, &gout[(threadIdx.x >> 6) * 256u + (threadIdx.x & 63u) + 64u * j]);
            hashes[tile * kR + threadIdx.x] = kv.x;
        }
    } else if (VARIANT == 1 || VARIANT == 2) {
        for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            const u32x4* src = reinterpret_cast<const u32x4*>(in + tile * kTile);
            if (VARIANT == 1) stage<5>(src, lds16); else stage<10>(src, lds16);
            __syncthreads();
            keys_out(lds64, keys + tile * 1024u, hashes + tile * kR);
            __syncthreads();
        }
    } else if (VARIANT == 3) {
        uint64_t tile = blockIdx.x;
        if (tile >= n_tiles) return;
        u32x4 v[kPer];
        auto request = [&](uint64_t t) {
            const u32x4* src = reinterpret_cast<const u32x4*>(in + t * kTile);
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) { const uint32_t c = threadIdx.x + k * kR; v[k] = __builtin_nontemporal_load(&src[c < kN16 ? c : kN16 - 1u]); }
        };
        request(tile);
        for (;;) {
#pragma unroll
            for (uint32_t k = 0; k < kPer; ++k) { const uint32_t c = threadIdx.x + k * kR; if (c < kN16) lds16[c] = v[k]; }
            __syncthreads();
            const uint64_t next = tile + gridDim.x;
            if (next < n_tiles) request(next);
            keys_out(lds64, keys + tile * 1024u, hashes + tile * kR);
            __syncthreads();
            if (next >= n_tiles) break;
            tile = next;
        }
    } else {
        // two tile buffers: stage tile b while the keys of tile 1-b leave
        u32x4* buf16[2] = {lds16, lds16 + (kTile + 64u) / 16u};
        uint64_t tile = blockIdx.x;
        if (tile >= n_tiles) return;
        stage<5>(reinterpret_cast<const u32x4*>(in + tile * kTile), buf16[0]);
        __syncthreads();
        for (uint32_t b = 0;; b ^= 1u) {
            const uint64_t next = tile + gridDim.x;
            if (next < n_tiles) {
                const u32x4* src = reinterpret_cast<const u32x4*>(in + next * kTile);
                u32x4 w[kPer];
#pragma unroll
                for (uint32_t k = 0; k < kPer; ++k) { const uint32_t c = threadIdx.x + k * kR; w[k] = __builtin_nontemporal_load(&src[c < kN16 ? c : kN16 - 1u]); }
                keys_out(reinterpret_cast<const uint64_t*>(buf16[b]), keys + tile * 1024u, hashes + tile * kR);
#pragma unroll
                for (uint32_t k = 0; k < kPer; ++k) { const uint32_t c = threadIdx.x + k * kR; if (c < kN16) buf16[b ^ 1u][c] = w[k]; }
            } else {
                keys_out(reinterpret_cast<const uint64_t*>(buf16[b]), keys + tile * 1024u, hashes + tile * kR);
            }
            __syncthreads();
            if (next >= n_tiles) break;
            tile = next;
        }
    }
}

// Registers only, as variant 0, but WHERE a workgroup's bytes lie is the parameter:
//   MAP 0  tile-contiguous, tiles dealt round-robin (tile = block + j * grid): the encoder's way, TILE_READS reads per tile
//   MAP 1  tile-contiguous, every workgroup walks its OWN run of consecutive tiles
//   MAP 2  grid-strided 4 KiB pieces: at step k workgroup b takes piece k * grid + b (the grid sweeps one dense window, like
//          tools/mix_probe); a workgroup's "tile" is then ten pieces 4 MiB apart (what an encoder would need: whole reads
//          per piece, i.e. pieces cut at read boundaries)
template <int MAP, int TILE_READS, int HASHMODE>
__global__ __launch_bounds__(256) void shape_map(const uint8_t* __restrict__ in, uint64_t n_reads, u64x2* __restrict__ keys, uint64_t* __restrict__ hashes)
{
    constexpr uint32_t tile_bytes = TILE_READS * kL, n16 = tile_bytes / 16, per = (n16 + kR - 1) / kR;
    const uint64_t n_tiles = n_reads / TILE_READS;
    if (MAP == 2) {
        const uint64_t n_pieces = n_reads * kL / 4096, n_kp = n_reads * 64 / 4096 * 4, n_hp = n_reads * 8 / 2048;
        const u32x4* src = reinterpret_cast<const u32x4*>(in);
        u32x4 acc = {0, 0, 0, 0};
        uint64_t kp = blockIdx.x, hp = blockIdx.x;                    // key pieces of 1 KiB x 4 waves, hash pieces of 2 KiB
        for (uint64_t p0 = blockIdx.x; p0 < n_pieces; p0 += 5ull * gridDim.x) {
            u32x4 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) { const uint64_t p = p0 + uint64_t(k) * gridDim.x; v[k] = __builtin_nontemporal_load(&src[(p < n_pieces ? p : n_pieces - 1) * 256u + threadIdx.x]); }
#pragma unroll
            for (int k = 0; k < 5; ++k) acc ^= v[k];
            const u64x2 kv = {uint64_t(acc.x) | (uint64_t(acc.y) << 32), uint64_t(acc.z) | (uint64_t(acc.w) << 32)};
            // 5 pieces in = 20480 B = 136.5 reads -> 8738 B of keys = 2.13 pieces of 4 KiB, 1092 B of hashes
#pragma unroll
            for (int j = 0; j < 2; ++j) { if (kp < n_kp / 4) __builtin_nontemporal_store(kv, &keys[kp * 256u + threadIdx.x]); kp += gridDim.x; }
            if ((p0 / gridDim.x / 5) % 8 == 0) { if (kp < n_kp / 4) __builtin_nontemporal_store(kv, &keys[kp * 256u + threadIdx.x]); kp += gridDim.x; }
            const uint64_t step = p0 / gridDim.x / 5;
            if (HASHMODE == 0 && step % 2 == 0) { if (hp < n_hp) hashes[hp * 256u + threadIdx.x] = kv.x; hp += gridDim.x; }
            if (HASHMODE == 3 && step % 2 == 0) { if (hp < n_hp) __builtin_nontemporal_store(kv.x, &hashes[hp * 256u + threadIdx.x]); hp += gridDim.x; }
            if (HASHMODE == 2 && step % 4 == 0) { if (hp < n_hp / 2) __builtin_nontemporal_store(kv, reinterpret_cast<u64x2*>(hashes) + hp * 256u + threadIdx.x); hp += gridDim.x; }
        }
        return;
    }
    const uint64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    for (uint64_t j = 0; j < per_wg; ++j) {
        const uint64_t tile = MAP == 0 ? blockIdx.x + j * gridDim.x : blockIdx.x * per_wg + j;
        if (tile >= n_tiles) break;
        const u32x4* src = reinterpret_cast<const u32x4*>(in + tile * tile_bytes);
        u32x4 acc = {0, 0, 0, 0};
        u32x4 v[per];
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) { const uint32_t c = threadIdx.x + k * kR; v[k] = __builtin_nontemporal_load(&src[c < n16 ? c : n16 - 1u]); }
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) acc ^= v[k];
        u64x2* gout = keys + tile * (TILE_READS * 4u);
        const u64x2 kv = {uint64_t(acc.x) | (uint64_t(acc.y) << 32), uint64_t(acc.z) | (uint64_t(acc.w) << 32)};
        for (uint32_t y = threadIdx.x; y < TILE_READS * 4u; y += 256u) __builtin_nontemporal_store(kv, &gout[y]);
        if (HASHMODE == 0 && threadIdx.x < TILE_READS) hashes[tile * TILE_READS + threadIdx.x] = kv.x;
        if (HASHMODE == 3 && threadIdx.x < TILE_READS) __builtin_nontemporal_store(kv.x, &hashes[tile * TILE_READS + threadIdx.x]);
        if (HASHMODE == 2 && threadIdx.x < TILE_READS / 2) __builtin_nontemporal_store(kv, reinterpret_cast<u64x2*>(hashes + tile * TILE_READS) + threadIdx.x);
    }
}

template <int MAP, int TILE_READS, int HASHMODE = 0>
static int run_map(const uint8_t* in, uint64_t n_reads, u64x2* keys, uint64_t* hashes, int blocks_per_cu, const char* label)
{
    const uint32_t grid = 256u * blocks_per_cu;
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    hipLaunchKernelGGL((shape_map<MAP, TILE_READS, HASHMODE>), dim3(grid), dim3(256), 0, 0, in, n_reads, keys, hashes);
    OK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((shape_map<MAP, TILE_READS, HASHMODE>), dim3(grid), dim3(256), 0, 0, in, n_reads, keys, hashes);
    OK(hipEventRecord(b, 0)); OK(hipEventSynchronize(b));
    float ms = 0; OK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double bytes = double(n_reads) * (HASHMODE == 1 ? 214.0 : 222.0);
    std::printf("{\"probe\": \"where a workgroup's bytes lie\", \"map\": %d, \"tile_reads\": %d, \"hash_stores\": \"%s\", \"what\": \"%s\", \"workgroups_per_cu\": %d, \"GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.2f}\n",
                MAP, TILE_READS, HASHMODE == 0 ? "8 B per lane, plain" : HASHMODE == 1 ? "none" : HASHMODE == 2 ? "16 B per lane, nt, half the lanes" : "8 B per lane, nt", label, blocks_per_cu, bytes / 1e9, ms, bytes / ms / 1e9);
    std::fflush(stdout);
    return 0;
}

template <int VARIANT>
static int run(const uint8_t* in, uint64_t n_tiles, u64x2* keys, uint64_t* hashes, int blocks_per_cu, const char* label)
{
    const size_t lds = VARIANT == 0 ? 0 : (VARIANT == 4 ? 2 * (kTile + 64) : kTile + 64);
    if (lds > 64 * 1024) OK(hipFuncSetAttribute(reinterpret_cast<const void*>(enc_shape<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const uint32_t grid = 256u * blocks_per_cu;
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    hipLaunchKernelGGL(enc_shape<VARIANT>, dim3(grid), dim3(256), lds, 0, in, n_tiles, keys, hashes);
    OK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(enc_shape<VARIANT>, dim3(grid), dim3(256), lds, 0, in, n_tiles, keys, hashes);
    OK(hipEventRecord(b, 0)); OK(hipEventSynchronize(b));
    float ms = 0; OK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double bytes = double(n_tiles) * (kTile + 256.0 * 72.0);
    std::printf("{\"probe\": \"encoder shape\", \"variant\": %d, \"what\": \"%s\", \"workgroups_per_cu\": %d, \"lds_bytes\": %zu, \"GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.2f}\n",
                VARIANT, label, blocks_per_cu, lds, bytes / 1e9, ms, bytes / ms / 1e9);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    const uint64_t reads = uint64_t((argc > 1 ? std::atof(argv[1]) : 100.0) * 1e6);
    const uint64_t n_tiles = reads / kR;
    uint8_t* in = nullptr; u64x2* keys = nullptr; uint64_t* hashes = nullptr;
    OK(hipMalloc(&in, n_tiles * kTile + 64)); OK(hipMalloc(&keys, n_tiles * 256ull * 64)); OK(hipMalloc(&hashes, n_tiles * 256ull * 8));
    OK(hipMemset(in, 'A', n_tiles * kTile + 64)); OK(hipDeviceSynchronize());
    int rc = 0;
    for (int rep = 0; rep < 2; ++rep) {
        rc |= run_map<0, 256>(in, reads, keys, hashes, 4, "tile-contiguous 38400 B, tiles round-robin (the encoder)");
        rc |= run_map<0, 128>(in, reads, keys, hashes, 4, "tile-contiguous 19200 B, tiles round-robin");
        rc |= run_map<0, 64>(in, reads, keys, hashes, 4, "tile-contiguous 9600 B, tiles round-robin");
        rc |= run_map<0, 64>(in, reads, keys, hashes, 8, "tile-contiguous 9600 B, tiles round-robin");
        rc |= run_map<0, 512>(in, reads, keys, hashes, 4, "tile-contiguous 76800 B, tiles round-robin");
        rc |= run_map<1, 256>(in, reads, keys, hashes, 4, "tile-contiguous 38400 B, a run of consecutive tiles per workgroup");
        rc |= run_map<2, 256>(in, reads, keys, hashes, 4, "grid-strided 4 KiB pieces, five in flight per lane");
        rc |= run_map<2, 256>(in, reads, keys, hashes, 8, "grid-strided 4 KiB pieces, five in flight per lane");
        rc |= run_map<2, 256, 1>(in, reads, keys, hashes, 4, "grid-strided 4 KiB pieces");
        rc |= run_map<2, 256, 2>(in, reads, keys, hashes, 4, "grid-strided 4 KiB pieces");
        rc |= run_map<2, 256, 3>(in, reads, keys, hashes, 4, "grid-strided 4 KiB pieces");
        rc |= run_map<0, 256, 1>(in, reads, keys, hashes, 4, "tile-contiguous 38400 B, tiles round-robin");
        rc |= run_map<0, 256, 2>(in, reads, keys, hashes, 4, "tile-contiguous 38400 B, tiles round-robin");
        rc |= run_map<0, 256, 3>(in, reads, keys, hashes, 4, "tile-contiguous 38400 B, tiles round-robin");
        rc |= run<0>(in, n_tiles, keys, hashes, 4, "registers only, tile-contiguous, 5 loads in flight per lane");
        rc |= run<0>(in, n_tiles, keys, hashes, 8, "registers only, tile-contiguous, 5 loads in flight per lane");
        rc |= run<1>(in, n_tiles, keys, hashes, 4, "staged through LDS, two barriers per tile (the encoder's skeleton)");
        rc |= run<2>(in, n_tiles, keys, hashes, 4, "staged, all 10 loads of a lane in flight at once");
        rc |= run<3>(in, n_tiles, keys, hashes, 4, "staged, next tile's loads issued before this tile's stores");
        rc |= run<4>(in, n_tiles, keys, hashes, 2, "two tiles per workgroup, loads of one under the stores of the other");
    }
    return rc;
}
