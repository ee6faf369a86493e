// lds_residency.hip -- how many workgroups with a given LDS allocation are resident per CU on gfx950?
// Each workgroup (64 threads) spins ~20 us; a launch of CUs x n workgroups takes ~20 us while n
// workgroups fit per CU, and steps up when they no longer do.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_residency lds_residency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_spin(unsigned long long *out, unsigned long long ticks)
{
    extern __shared__ float s[];
    s[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { s[threadIdx.x] += 1.0f; }
    if (s[threadIdx.x] == 12345.0f) out[0] = 1;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.multiProcessorCount,
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    unsigned long long *d;
    (void)hipMalloc(&d, 64);
    for (int threads : {64, 512}) {
        for (size_t kb : {8, 16, 20, 32, 40, 64}) {
            const size_t lds = kb * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                printf("  (set attribute failed for %zu KB)\n", kb);
            printf("threads %3d LDS %2zu KB:", threads, kb);
            for (int n = 1; n <= 10; ++n) {
                if (n * threads > 2048) break;
                hipLaunchKernelGGL(k_spin, dim3(p.multiProcessorCount * n), dim3(threads), lds, 0, d, 2000ull);
                (void)hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_spin, dim3(p.multiProcessorCount * n), dim3(threads), lds, 0, d, 2000ull);
                (void)hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf(" n=%d:%.0fus", n, us);
            }
            printf("\n");
        }
    }
    return 0;
}
