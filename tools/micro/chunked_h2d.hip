// chunked_h2d.hip -- what does it cost to send 8 MB of PAGEABLE host memory to the device in C pieces instead of one
// (hipMemcpyAsync from pageable memory returns when the source has been read; an event recorded behind every piece), and
// what does page-locking the caller's buffer (hipHostRegister) change?  Host clock, the last piece's arrival included.
// build: hipcc --offload-arch=gfx950 -O3 -o chunked_h2d chunked_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 501501ull * 16;   // SYN-1M's vertices as pcl::PointXYZ records
    void *src = nullptr;
    if (posix_memalign(&src, 4096, bytes)) return 1;
    memset(src, 1, bytes);
    void *dst;
    CK(hipMalloc(&dst, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int pinned = 0; pinned < 2; ++pinned) {
        if (pinned) {
            const double t0 = now_us();
            CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
            printf("hipHostRegister of %.1f MB: %.0f us\n", bytes / 1e6, now_us() - t0);
        }
        for (int C : {1, 2, 4, 8, 16, 32}) {
            double best = 1e18, sum = 0, enq = 0;
            const int reps = 40;
            for (int rep = 0; rep < reps + 5; ++rep) {
                CK(hipStreamSynchronize(s));
                const double t0 = now_us();
                for (int c = 0; c < C; ++c) {
                    const size_t a = bytes * c / C / 4096 * 4096, b = c + 1 == C ? bytes : bytes * (c + 1) / C / 4096 * 4096;
                    CK(hipMemcpyAsync((char *)dst + a, (char *)src + a, b - a, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[c], s));
                }
                const double t1 = now_us();
                CK(hipStreamSynchronize(s));
                const double t2 = now_us();
                if (rep >= 5) { sum += t2 - t0; enq += t1 - t0; best = t2 - t0 < best ? t2 - t0 : best; }
            }
            printf("%s source, %2d pieces: %.1f us (min %.1f), host busy %.1f us  -> %.1f GB/s\n", pinned ? "registered" : "pageable  ", C, sum / reps, best,
                   enq / reps, bytes / (sum / reps) / 1e3);
        }
    }
    CK(hipHostUnregister(src));
    return 0;
}
