// atomic_probe — measures the primitive rates the insert kernel is built from, on the
// real chip: random 8-byte loads / CAS / atomicMin / no-return atomics into tables of
// different sizes (HBM-sized, Infinity-Cache-sized, L2-sized), and 64-B key gathers.
// Design aid only (not shipped, not part of the timed path).
//   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip && ./atomic_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

enum Op { LOAD8 = 0, CAS, CAS_UNIQUE, MIN_RET, MIN_NORET, STORE8, LOAD64B, CAS_ILP4, OPS };
static const char* kNames[OPS] = {"load8", "cas(ret)", "cas-claim", "umin(ret)", "umin(noret)", "store8", "gather64B", "cas ilp4"};

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long* tab, uint64_t mask, uint64_t n, uint64_t seed, unsigned long long* sink)
{
    unsigned long long acc = 0;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    if (OP == CAS_ILP4) {
        for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
            unsigned long long r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t h = mix(seed ^ (i + k * stride));
                r[k] = atomicCAS(&tab[h & mask], ~0ull, h | 1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += r[k];
        }
    } else {
        for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) {
            const uint64_t h = mix(seed ^ i);
            const uint64_t p = h & mask;
            if (OP == LOAD8)      acc += tab[p];
            if (OP == CAS)        acc += atomicCAS(&tab[p], ~0ull, h | 1);
            if (OP == CAS_UNIQUE) { uint64_t q = p; while (atomicCAS(&tab[q], ~0ull, h | 1) != ~0ull) q = (q + 1) & mask; }
            if (OP == MIN_RET)    acc += atomicMin(&tab[p], (unsigned long long)h);
            if (OP == MIN_NORET)  atomicMin(&tab[p], (unsigned long long)h);
            if (OP == STORE8)     tab[p] = h;
            if (OP == LOAD64B) {
                const ulonglong2* q = reinterpret_cast<const ulonglong2*>(tab + (p & ~7ull));
                ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
                acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
            }
        }
    }
    if (acc == 0x1234567) *sink = acc;
}

template <int OP>
float run(unsigned long long* tab, uint64_t slots, uint64_t n, unsigned long long* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(tab, 0xFF, slots * 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(probe<OP>, dim3(2048), dim3(256), 0, 0, tab, slots - 1, n, 77 + rep, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const uint64_t n = 100000000ull;
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    const uint64_t sizes[] = {1ull << 28, 1ull << 24, 1ull << 21, 1ull << 18};   // 2 GiB, 128 MiB, 16 MiB, 2 MiB
    printf("%-12s", "op \\ table");
    for (uint64_t s : sizes) printf("%14.0f MiB", s * 8 / 1048576.0);
    printf("   (G ops/s over %llu M ops)\n", (unsigned long long)(n / 1000000));
    unsigned long long* tab; CK(hipMalloc(&tab, sizes[0] * 8));
    for (int op = 0; op < OPS; ++op) {
        printf("%-12s", kNames[op]);
        for (uint64_t s : sizes) {
            if (op == CAS_UNIQUE && s < 2 * n) { printf("%18s", "-"); continue; }
            float ms = 0;
            switch (op) {
                case LOAD8: ms = run<LOAD8>(tab, s, n, sink); break;
                case CAS: ms = run<CAS>(tab, s, n, sink); break;
                case CAS_UNIQUE: ms = run<CAS_UNIQUE>(tab, s, n, sink); break;
                case MIN_RET: ms = run<MIN_RET>(tab, s, n, sink); break;
                case MIN_NORET: ms = run<MIN_NORET>(tab, s, n, sink); break;
                case STORE8: ms = run<STORE8>(tab, s, n, sink); break;
                case LOAD64B: ms = run<LOAD64B>(tab, s, n, sink); break;
                case CAS_ILP4: ms = run<CAS_ILP4>(tab, s, n, sink); break;
            }
            printf("%11.2f (%5.1fms)", n / (ms * 1e-3) / 1e9, ms);
        }
        printf("\n");
    }
    return 0;
}
