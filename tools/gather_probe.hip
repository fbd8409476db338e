// gather_probe — how many random 64-byte lines per second does this chip's HBM serve?
// The dedup kernel's verify phase fetches two random 64-byte key lines per duplicate; DESIGN §7 puts its bound at
// the ~31 G lines/s that phase reaches.  This measures the rate a bare gather reaches over the same footprint:
// 4 lanes x 16 B per line, K lines in flight per 4-lane group, every line index a hash of a counter.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_probe tools/gather_probe.hip ; run: tools/gather_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

template <int K, int LANES>   // LANES per line: 4 x 16 B or 8 x 8 B
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ base, uint64_t n_lines, uint64_t per_group, uint64_t* __restrict__ out)
{
    const uint64_t group = (blockIdx.x * uint64_t(256) + threadIdx.x) / LANES;
    const uint32_t sub = threadIdx.x % LANES;
    const uint64_t groups = gridDim.x * uint64_t(256) / LANES;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < per_group; i += K) {
        uint64_t v[K][2];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t line = mix(group + (i + k) * groups) % n_lines;
            if (LANES == 4) { const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(base + line * 64 + sub * 16); v[k][0] = x.x; v[k][1] = x.y; }
            else            { v[k][0] = *reinterpret_cast<const uint64_t*>(base + line * 64 + sub * 8); v[k][1] = 0; }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k][0] ^ v[k][1];
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

template <int K, int LANES>
static int run(const uint8_t* buf, uint64_t n_lines, uint64_t* out, int blocks_per_cu, const char* label)
{
    const uint64_t total = 80ull << 20;                       // lines fetched per launch
    const uint32_t grid = 256u * blocks_per_cu;
    const uint64_t groups = grid * uint64_t(256) / LANES;
    const uint64_t per_group = (total / groups + K - 1) / K * K;
    hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<K, LANES>), dim3(grid), dim3(256), 0, 0, buf, n_lines, per_group, out);
    OK(hipEventRecord(a, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((gather<K, LANES>), dim3(grid), dim3(256), 0, 0, buf, n_lines, per_group, out);
    OK(hipEventRecord(b, 0)); OK(hipEventSynchronize(b));
    float ms = 0; OK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double lines = double(per_group) * groups;
    std::printf("{\"probe\": \"%s\", \"lanes_per_line\": %d, \"in_flight_per_group\": %d, \"blocks_per_cu\": %d, \"ms\": %.3f, \"Glines_per_s\": %.1f, \"TB_per_s\": %.2f}\n",
                label, LANES, K, blocks_per_cu, ms, lines / ms / 1e6, lines * 64 / ms / 1e9);
    return 0;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? std::atof(argv[1]) : 6.0;
    const uint64_t bytes = uint64_t(gib * (1ull << 30)) / 64 * 64;
    uint8_t* buf = nullptr; uint64_t* out = nullptr;
    OK(hipMalloc(&buf, bytes)); OK(hipMalloc(&out, 64)); OK(hipMemset(buf, 1, bytes)); OK(hipDeviceSynchronize());
    const uint64_t n_lines = bytes / 64;
    std::printf("{\"footprint_GiB\": %.1f}\n", gib);
    int rc = 0;
    rc |= run<2, 4>(buf, n_lines, out, 8, "random 64B lines");
    rc |= run<4, 4>(buf, n_lines, out, 8, "random 64B lines");
    rc |= run<8, 4>(buf, n_lines, out, 8, "random 64B lines");
    rc |= run<16, 4>(buf, n_lines, out, 8, "random 64B lines");
    rc |= run<8, 4>(buf, n_lines, out, 4, "random 64B lines");
    rc |= run<8, 8>(buf, n_lines, out, 8, "random 64B lines");
    rc |= run<16, 8>(buf, n_lines, out, 8, "random 64B lines");
    return rc;
}
