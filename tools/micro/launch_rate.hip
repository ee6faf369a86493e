// launch_rate.hip -- how fast does gfx950 start waves?  Every wave records its start time (100 MHz
// wall clock) and spins 10 us so that nothing retires during the ramp; the spread of the start times
// over a grid that just fills the chip is the dispatch time.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_start(unsigned long long *out, size_t lds_touch)
{
    extern __shared__ float s[];
    const unsigned long long t0 = wall_clock64();
    if (lds_touch) s[threadIdx.x] = 1.0f;
    while (wall_clock64() - t0 < 1000ull) {}
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t0;
}

int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 8 * 65536);
    std::vector<unsigned long long> h(65536);
    struct Cfg { int threads, waves_total; size_t lds; };
    const Cfg cfgs[] = {{64, 4096, 0}, {256, 4096, 0}, {512, 4096, 0}, {1024, 4096, 0}, {256, 8192, 0}, {1024, 8192, 0},
                        {256, 4096, 20480}, {256, 4096, 38912}, {512, 4096, 32768}, {256, 2048, 0}, {64, 2048, 0}};
    for (const Cfg &c : cfgs) {
        const int blocks = c.waves_total * 64 / c.threads;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_start), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_start, dim3(blocks), dim3(c.threads), c.lds, 0, d, c.lds);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d, 8 * c.waves_total, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> t(h.begin(), h.begin() + c.waves_total);
        std::sort(t.begin(), t.end());
        const double span = (t.back() - t.front()) / 100.0, p50 = (t[t.size() / 2] - t.front()) / 100.0, p90 = (t[t.size() * 9 / 10] - t.front()) / 100.0;
        printf("threads %4d waves %5d lds %5zu: start p50 %.2f us p90 %.2f us last %.2f us -> %.0f waves/us\n", c.threads, c.waves_total, c.lds, p50,
               p90, span, c.waves_total / span);
    }
    return 0;
}
