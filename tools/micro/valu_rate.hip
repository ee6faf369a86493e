// valu_rate.hip -- VALU issue rate of a gfx950 SIMD: every wave runs a chain of dependent
// v_fma_f32 (inline asm, nothing for the compiler to pack or drop) at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INDEP>
__global__ void k_chain(unsigned long long *out, int iters)
{
    float a[INDEP];
    for (int i = 0; i < INDEP; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const float b = 1.0001f, c = 0.5f;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < INDEP; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
    }
    const unsigned long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < INDEP; ++i) s += a[i];
    if (s == 12345.0f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = c1 - c0;
}

template <int INDEP>
void run(unsigned long long *d, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 2000;
    std::vector<unsigned long long> h(blocks * 4);
    hipLaunchKernelGGL(k_chain<INDEP>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipLaunchKernelGGL(k_chain<INDEP>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (auto v : h) c += v;
    c /= h.size();
    const double ops = 8.0 * INDEP * iters;
    printf("indep %d, %d waves/SIMD: %.2f cycles per v_fma per wave -> SIMD issues one wave64 op every %.2f cycles\n", INDEP,
           waves_per_simd, c / ops, c / ops / waves_per_simd);
}

int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 8 * 2048 * 4 * 8);
    for (int w : {1, 2, 4, 8}) run<1>(d, w);
    for (int w : {1, 2, 4, 8}) run<4>(d, w);
    return 0;
}
