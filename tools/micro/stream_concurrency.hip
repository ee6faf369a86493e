// How many streams of one process run kernels side by side on this device (the runtime multiplexes streams onto a few
// hardware queues): K streams each take one idle 200 us wave; the wall time is 200 us x ceil(K / concurrent queues).
// build: hipcc --offload-arch=gfx950 -O2 -o stream_concurrency stream_concurrency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main()
{
    for (int K = 1; K <= 8; ++K) {
        std::vector<hipStream_t> s(K);
        for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (auto &x : s) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, 1000ull);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; ++r) for (auto &x : s) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, x, 20000ull);   // 200 us at 100 MHz
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5.0;
        std::printf("%d streams: %.0f us per round of one 200 us wave each -> %.1f side by side\n", K, us, K * 200.0 / us);
        for (auto &x : s) hipStreamDestroy(x);
    }
    return 0;
}
