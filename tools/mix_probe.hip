// mix_probe — what does this chip's HBM sustain for the staged encoder's TRAFFIC SHAPE, with no work attached?
// The encoder reads 150 B and writes 72 B per read (64-byte key + 8-byte hash): 22.2 GB per 100 M reads in 3.7-4.2 ms =
// 5.3-6.0 TB/s.  The guide's 6.29 TB/s "copy ceiling" is a 1:1 copy.  This measures bare grid-stride kernels that move
// the same bytes in the same proportion — R 16-byte loads per W 16-byte stores per lane and step, non-temporal both —
// so that the encoder can be priced against ITS OWN ceiling.
// (The first run of this probe, gpurun_out/r4/mix_probe.jsonl, reported 7.3-7.5 TB/s for the 2:1 mix: its stores used
// only some of the loaded values and the compiler had removed the other loads.  Fixed; see below.)
//   R:W = 1:0 read only · 1:1 copy · 2:1 the encoder's mix (150:72 = 2.08:1) · 0:1 write only
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mix_probe tools/mix_probe.hip ; run: tools/mix_probe [GB read, default 15]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Every lane: per step R loads (16 B each, consecutive lanes consecutive chunks, R planes a grid-width apart) and W stores.
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void mix(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t steps, uint64_t* __restrict__ sink)
{
    const uint64_t lanes = gridDim.x * uint64_t(256), me = blockIdx.x * uint64_t(256) + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t s = 0; s < steps; ++s) {
        u32x4 v[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const u32x4* p = &src[(s * R + r) * lanes + me];
            v[r] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        // every store carries a value that depends on EVERY load of the step: the first version of this probe stored
        // v[w % R], the compiler dropped the loads nobody used, and its "2:1 mix" was a 1:1 copy of half the bytes
        u32x4 all = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r) all ^= v[r];
        acc ^= all;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            u32x4* q = &dst[(s * W + w) * lanes + me];
            const u32x4 x = R > 0 ? all + u32x4{uint32_t(w), 0, 0, 0} : u32x4{uint32_t(s), uint32_t(w), 0, 0};
            if (NT) __builtin_nontemporal_store(x, q); else *q = x;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

template <int R, int W, bool NT>
static int run(const u32x4* src, u32x4* dst, uint64_t* sink, uint64_t read_bytes, int blocks_per_cu, const char* label)
{
    const uint32_t grid = 256u * blocks_per_cu;
    const uint64_t lanes = grid * uint64_t(256);
    const uint64_t per_step = lanes * 16ull * (R > 0 ? R : W);
    const uint64_t steps = read_bytes / per_step;
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    hipLaunchKernelGGL((mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, src, dst, steps, sink);
    OK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((mix<R, W, NT>), dim3(grid), dim3(256), 0, 0, src, dst, steps, sink);
    OK(hipEventRecord(b, 0)); OK(hipEventSynchronize(b));
    float ms = 0; OK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double rd = double(steps) * lanes * 16.0 * R, wr = double(steps) * lanes * 16.0 * W;
    std::printf("{\"probe\": \"%s\", \"loads_per_step\": %d, \"stores_per_step\": %d, \"nontemporal\": %s, \"blocks_per_cu\": %d, \"read_GB\": %.2f, "
                "\"written_GB\": %.2f, \"ms\": %.3f, \"TB_per_s\": %.2f}\n",
                label, R, W, NT ? "true" : "false", blocks_per_cu, rd / 1e9, wr / 1e9, ms, (rd + wr) / ms / 1e9);
    return 0;
}

int main(int argc, char** argv)
{
    const double gb = argc > 1 ? std::atof(argv[1]) : 15.0;
    const uint64_t bytes = uint64_t(gb * 1e9) / 4096 * 4096;
    u32x4 *src = nullptr, *dst = nullptr; uint64_t* sink = nullptr;
    OK(hipMalloc(&src, bytes)); OK(hipMalloc(&dst, bytes)); OK(hipMalloc(&sink, 64));
    OK(hipMemset(src, 1, bytes)); OK(hipMemset(dst, 2, bytes)); OK(hipDeviceSynchronize());
    int rc = 0;
    for (int bpc : {4, 8}) {
        rc |= run<4, 0, true>(src, dst, sink, bytes, bpc, "read only");
        rc |= run<0, 4, true>(src, dst, sink, bytes, bpc, "write only");
        rc |= run<2, 2, true>(src, dst, sink, bytes, bpc, "copy 1:1");
        rc |= run<4, 2, true>(src, dst, sink, bytes, bpc, "encoder mix 2:1");
        rc |= run<6, 3, true>(src, dst, sink, bytes, bpc, "encoder mix 2:1, deeper");
        rc |= run<4, 2, false>(src, dst, sink, bytes, bpc, "encoder mix 2:1");
    }
    return rc;
}
