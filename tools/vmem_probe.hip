// vmem_probe — what does ONE vector memory instruction cost a CU when its lanes ask for scattered addresses, and does
// the cost depend on how many lanes are active?  (The wave-per-member inflater issues one load per decoded code for
// whichever few lanes need more bits)  Each wave issues `iters` loads (or 8-byte stores) of 4 or 8
// bytes per active lane, every lane in a line of its own inside a footprint that fits the L2; 16 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmem_probe tools/vmem_probe.hip ; run: tools/vmem_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BYTES, bool STORE>
__global__ __launch_bounds__(64) void probe(uint8_t* __restrict__ base, uint32_t active, uint32_t iters, uint32_t lines_per_wave, uint64_t* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x;
    uint8_t* mine = base + (uint64_t(blockIdx.x) * lines_per_wave) * 64ull;
    uint64_t acc = 0;
    if (lane < active) {
        for (uint32_t i = 0; i < iters; i += 8) {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t line = (lane * 17u + (i + k) * 5u) % lines_per_wave;            // scattered over the wave's lines
                uint8_t* p = mine + line * 64ull + ((i + k) & 7u) * (BYTES == 1 ? 1 : 8) + (BYTES == 8 ? 3 : 0);   // (8-byte accesses unaligned, as the copies are)
                if (STORE) { if (BYTES == 8) { uint64_t v = i + k; __builtin_memcpy(p, &v, 8); } else if (BYTES == 4) *reinterpret_cast<uint32_t*>(p) = i; else *p = uint8_t(i); }
                else { if (BYTES == 8) { uint64_t v; __builtin_memcpy(&v, p, 8); acc += v; } else if (BYTES == 4) acc += *reinterpret_cast<const uint32_t*>(p); else acc += *p; }
            }
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

template <int BYTES, bool STORE>
static int run(uint8_t* buf, uint64_t* sink, uint32_t active, const char* what)
{
    const uint32_t grid = 256u * 16u, iters = 4096, lines = 256;      // 16 KB per wave: 64 MB in all
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<BYTES, STORE>), dim3(grid), dim3(64), 0, 0, buf, active, iters, lines, sink);
    OK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<BYTES, STORE>), dim3(grid), dim3(64), 0, 0, buf, active, iters, lines, sink);
    OK(hipEventRecord(b, 0)); OK(hipEventSynchronize(b));
    float ms = 0; OK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double per_cu = double(iters) * 16.0;                                    // instructions a CU issued
    std::printf("{\"probe\": \"%s\", \"bytes\": %d, \"active_lanes\": %u, \"ms\": %.3f, \"ns_per_instruction_per_cu\": %.1f}\n", what, BYTES, active, ms, ms * 1e6 / per_cu);
    return 0;
}

int main()
{
    uint8_t* buf = nullptr; uint64_t* sink = nullptr;
    OK(hipMalloc(&buf, (256ull * 16 * 256 + 4) * 64)); OK(hipMalloc(&sink, 64));
    OK(hipMemset(buf, 1, (256ull * 16 * 256 + 4) * 64));
    for (uint32_t active : {1u, 4u, 16u, 64u}) {
        if (run<4, false>(buf, sink, active, "load")) return 1;
        if (run<8, false>(buf, sink, active, "load unaligned")) return 1;
        if (run<1, true>(buf, sink, active, "store")) return 1;
        if (run<8, true>(buf, sink, active, "store unaligned")) return 1;
    }
    return 0;
}
