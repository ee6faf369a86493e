// strided_h2d.hip -- the adapter's vertices arrive as 16-byte pcl::PointXYZ; only 12 bytes of each are coordinates.  Does a
// pitched copy (hipMemcpy2DAsync: rows of 12 bytes, source pitch 16) move the 6 MB faster than a flat copy moves the 8 MB?
// build: hipcc --offload-arch=gfx950 -O3 -o strided_h2d strided_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 501501;
    void *src = nullptr;
    if (posix_memalign(&src, 4096, n * 16)) return 1;
    memset(src, 1, n * 16);
    void *pinned = nullptr;
    CK(hipHostMalloc(&pinned, n * 16, 0));
    memset(pinned, 1, n * 16);
    void *dst = nullptr;
    CK(hipMalloc(&dst, n * 16));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct { const char *what; int mode; } cases[] = {{"flat 8 MB, pageable", 0}, {"pitched 12-of-16, pageable", 1}, {"flat 8 MB, pinned", 2},
                                                      {"pitched 12-of-16, pinned", 3}, {"pitched, rows of 3072 x 12 B (pageable)", 4}};
    for (auto &c : cases) {
        double sum = 0, best = 1e9;
        for (int rep = 0; rep < 25; ++rep) {
            const void *from = (c.mode & 2) ? pinned : src;
            const double t0 = now();
            if (c.mode == 0 || c.mode == 2) CK(hipMemcpyAsync(dst, from, n * 16, hipMemcpyHostToDevice, s));
            else if (c.mode == 4) CK(hipMemcpy2DAsync(dst, 12, src, 16, 12, n, hipMemcpyHostToDevice, s));
            else CK(hipMemcpy2DAsync(dst, 12, from, 16, 12, n, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            const double dt = (now() - t0) * 1e6;
            if (rep >= 5) { sum += dt; best = dt < best ? dt : best; }
        }
        printf("%-44s %8.1f us (min %.1f)\n", c.what, sum / 20, best);
    }
    return 0;
}
