// host_read.hip -- can a KERNEL fetch the caller's (page-locked) vertices over PCIe as fast as the runtime's copy engine does?
// If it can, a frame's vertex upload could be pieces of a kernel (no per-piece runtime cost: chunked_h2d.hip found 15 - 20 us per
// hipMemcpyAsync piece) with the projection of each piece's triangles riding behind it.  8 MB (SYN-1M's vertices as 16-byte
// records): hipMemcpyAsync from the registered buffer; k_fetch with G workgroups of 256 lanes, 16 bytes per lane and load, U
// loads in flight per lane, reading the registered buffer's device alias and writing device memory; the same in P pieces
// (one launch per piece, same stream).  Host clock around enqueue + wait, and the kernels' own time by events.
// build: hipcc --offload-arch=gfx950 -O3 -o host_read host_read.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int U>
__global__ __launch_bounds__(256) void k_fetch(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) dst[i + u * stride] = v[u];
    }
}

int main()
{
    const size_t bytes = 501501ull * 16, n = bytes / 16;
    void *src = nullptr;
    if (posix_memalign(&src, 4096, bytes)) return 1;
    memset(src, 1, bytes);
    void *dst;
    CK(hipMalloc(&dst, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
    void *alias = nullptr;
    CK(hipHostGetDevicePointer(&alias, src, 0));
    void *pinned = nullptr;
    CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 2, bytes);
    auto run = [&](const char *what, auto &&enqueue) {
        double sum = 0, best = 1e18, dev = 0;
        const int reps = 30;
        for (int rep = 0; rep < reps + 5; ++rep) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            hipEventRecord(e0, s);
            enqueue();
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            const double t1 = now_us();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 5) { sum += t1 - t0; dev += ms * 1e3; best = t1 - t0 < best ? t1 - t0 : best; }
        }
        printf("%-64s host %.1f us (min %.1f), by events %.1f us -> %.1f GB/s\n", what, sum / reps, best, dev / reps, bytes / (dev / reps) / 1e3);
        fflush(stdout);
    };
    run("hipMemcpyAsync, registered source", [&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s); });
    run("hipMemcpyAsync, hipHostMalloc source", [&] { hipMemcpyAsync(dst, pinned, bytes, hipMemcpyHostToDevice, s); });
    char name[128];
    for (int which = 0; which < 2; ++which) {
        const uint4 *from = (const uint4 *)(which ? pinned : alias);
        for (int G : {64, 256, 1024, 4096}) {
            snprintf(name, sizeof name, "k_fetch<4>, %s, %d workgroups", which ? "hipHostMalloc" : "registered", G);
            run(name, [&] { hipLaunchKernelGGL(k_fetch<4>, dim3(G), dim3(256), 0, s, from, (uint4 *)dst, n); });
        }
        snprintf(name, sizeof name, "k_fetch<1>, %s, 2048 workgroups", which ? "hipHostMalloc" : "registered");
        run(name, [&] { hipLaunchKernelGGL(k_fetch<1>, dim3(2048), dim3(256), 0, s, from, (uint4 *)dst, n); });
        snprintf(name, sizeof name, "k_fetch<8>, %s, 256 workgroups", which ? "hipHostMalloc" : "registered");
        run(name, [&] { hipLaunchKernelGGL(k_fetch<8>, dim3(256), dim3(256), 0, s, from, (uint4 *)dst, n); });
        for (int P : {2, 4, 8}) {
            snprintf(name, sizeof name, "k_fetch<4> in %d pieces, %s, 256 workgroups each", P, which ? "hipHostMalloc" : "registered");
            run(name, [&] {
                for (int p = 0; p < P; ++p) {
                    const size_t a = n * p / P, b = n * (p + 1) / P;
                    hipLaunchKernelGGL(k_fetch<4>, dim3(256), dim3(256), 0, s, from + a, (uint4 *)dst + a, b - a);
                }
            });
        }
    }
    CK(hipHostUnregister(src));
    return 0;
}
