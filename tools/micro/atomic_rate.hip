// atomic_rate.hip -- what do a million fire-and-forget device-scope atomic adds cost, next to a million plain scattered stores?
// (Would a radix-sort scatter pass be able to histogram the NEXT digit per destination tile on the fly -- one atomic per key into
// tiles x 1 024 counters -- instead of a separate counting launch?)  One thread per key, keys hashed; counters = 256 x 1 024 words
// (1 MB: next digit by destination tile) or 16 x 1 024 words (64 KB: by group of sixteen tiles), or both.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>   // 0: scattered stores only, 1: + atomic into 1 MB, 2: + atomic into 64 KB, 3: + both
__global__ __launch_bounds__(512) void k(uint32_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ big, uint32_t *__restrict__ small)
{
    const uint32_t i = blockIdx.x * 512u + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint32_t k = i * 8u + r;
        if (k >= n) return;
        const uint32_t h = hash(k);
        const uint32_t at = h % n;                 // where the key goes
        out[at] = h;
        if (MODE & 1) atomicAdd(&big[(at >> 12) * 1024u + ((h >> 20) & 1023u)], 1u);
        if (MODE & 2) atomicAdd(&small[(at >> 16) * 1024u + ((h >> 20) & 1023u)], 1u);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
int run(uint32_t n, uint32_t *out, uint32_t *big, uint32_t *small_, const char *what)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float sum = 0.f, best = 1e9f;
    for (int rep = 0; rep < 30; ++rep) {
        hipExtLaunchKernelGGL((k<MODE>), dim3((n / 8 + 511) / 512), dim3(512), 0, 0, e0, e1, 0, n, out, big, small_);
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 5) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-44s %.2f us (min %.2f)\n", what, sum / 25 * 1e3, best * 1e3);
    return 0;
}

int main()
{
    const uint32_t n = 1000000;
    uint32_t *out, *big, *small_;
    CK(hipMalloc(&out, (size_t)n * 4));
    CK(hipMalloc(&big, 256 * 1024 * 4));
    CK(hipMalloc(&small_, 16 * 1024 * 4));
    CK(hipMemset(big, 0, 256 * 1024 * 4));
    CK(hipMemset(small_, 0, 16 * 1024 * 4));
    if (run<0>(n, out, big, small_, "1 M scattered stores") || run<1>(n, out, big, small_, "+ 1 M atomics into 262 144 counters") ||
        run<2>(n, out, big, small_, "+ 1 M atomics into 16 384 counters") || run<3>(n, out, big, small_, "+ both"))
        return 1;
    return 0;
}
