// first_launch.hip -- what the first launch on a stream costs the HOST after the stream has been waited for.  Seen in the frame
// loop: of a window of frames that starts on an idle device, the first frame on each of the three slot streams takes ~21 us
// to enqueue (three launches + two small calls), every later one ~10.5.  Three streams, K rounds of one small kernel per stream;
// before each window the device is made idle in one of several ways; the host time of every launch call is recorded.
// build: hipcc --offload-arch=gfx950 -O2 -o first_launch first_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_small(float *p, int n, volatile unsigned *flag, unsigned v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
    if (flag && i == 0) { __threadfence_system(); *flag = v; }
}
int main(int argc, char **argv)
{
    const unsigned long long busy_ticks = argc > 1 ? strtoull(argv[1], nullptr, 10) : 0ull;   // 100 MHz ticks the device stays busy behind every window
    printf("device kept busy for %.0f us behind each window\n", busy_ticks / 100.0);
    const int S = 3, ROUNDS = 6, N = 1 << 20, WINDOWS = 60;
    hipStream_t s[S];
    float *buf[S];
    hipEvent_t ev[S];
    for (int i = 0; i < S; ++i) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipMalloc(&buf[i], N * 4)); CK(hipMemset(buf[i], 0, N * 4)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    unsigned *flags;
    CK(hipHostMalloc(&flags, 64 * S, hipHostMallocDefault));
    memset(flags, 0, 64 * S);
    const char *names[] = {"hipDeviceSynchronize", "hipStreamSynchronize x3", "hipEventSynchronize x3", "hipStreamQuery polled", "host flag polled (no HIP wait at all)"};
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<std::vector<double>> t(S * ROUNDS);
        unsigned tag = 0;
        for (int w = 0; w < WINDOWS; ++w) {
            ++tag;
            // make the device idle
            if (mode == 0) CK(hipDeviceSynchronize());
            else if (mode == 1) for (int i = 0; i < S; ++i) CK(hipStreamSynchronize(s[i]));
            else if (mode == 2) for (int i = 0; i < S; ++i) { CK(hipEventRecord(ev[i], s[i])); CK(hipEventSynchronize(ev[i])); }
            else if (mode == 3) for (int i = 0; i < S; ++i) while (hipStreamQuery(s[i]) == hipErrorNotReady) {}
            else if (w) for (int i = 0; i < S; ++i) while (((volatile unsigned *)flags)[16 * i] != tag - 1) {}
            for (int r = 0; r < ROUNDS; ++r)
                for (int i = 0; i < S; ++i) {
                    const bool last = r == ROUNDS - 1;
                    const double t0 = now_us();
                    hipLaunchKernelGGL(k_small, dim3(N / 256), dim3(256), 0, s[i], buf[i], N, last ? flags + 16 * i : nullptr, tag);
                    t[r * S + i].push_back(now_us() - t0);
                }
            if (busy_ticks) for (int i = 0; i < S; ++i) {
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], busy_ticks);
                hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s[i], buf[i], 64, flags + 16 * i, tag);   // the flag follows the busy kernel
            }
        }
        CK(hipDeviceSynchronize());
        printf("%-40s launch call us (median), rounds x streams:", names[mode]);
        for (int k = 0; k < S * ROUNDS; ++k) {
            std::sort(t[k].begin(), t[k].end());
            printf("%s%.1f", k % S ? " " : " | ", t[k][t[k].size() / 2]);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
