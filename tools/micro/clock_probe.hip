// clock_probe.hip -- shader clock actually delivered during short kernels: s_memtime (core clock
// counter) against the 100 MHz wall clock, for a chip-filling VALU loop of a few tens of
// microseconds, launched back to back like the frames of bench.py.
// build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_busy(unsigned long long *out, int iters)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; a = a * b + 0.25f; b = b * 1.00001f - 1e-6f; }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (a == 12345.0f) out[0] = 1;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

int main()
{
    const int blocks = 2048, threads = 256;
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks);
    std::vector<unsigned long long> h(2 * blocks);
    for (int iters : {500, 2000, 8000}) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double c = 0, w = 0;
            for (int b = 0; b < blocks; ++b) { c += h[2 * b]; w += h[2 * b + 1]; }
            // 4 dependent VALU ops per iteration; 8 waves per SIMD share the pipe
            printf("iters %5d: wave time %.2f us, s_memtime/wall = %.1f MHz; VALU issue: %.2f cycles per op per wave\n", iters,
                   w / blocks / 100.0, c / w * 100.0, (c / blocks) / (4.0 * iters));
        }
    }
    return 0;
}
