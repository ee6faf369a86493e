// Microbenchmark: cost of divergent per-lane gathers on gfx950 as a function of load width and
// of the number of active lanes (the trace kernel's node fetch pattern).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W, int STRIDE>  // W = dwords per lane per load (1,2,4); STRIDE loads of one 64-B record
__global__ void k_gather(const float* __restrict__ buf, const unsigned* __restrict__ idx, unsigned nrec, int iters,
                         unsigned active_mask_lo, unsigned active_mask_hi, float* out)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long mask = ((unsigned long long)active_mask_hi << 32) | active_mask_lo;
    const bool on = (mask >> lane) & 1ull;
    unsigned r = idx[(blockIdx.x * blockDim.x + threadIdx.x) % nrec];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (on) {
            const float* p = buf + (size_t)r * 16;
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) {
                if (W == 4) { float4 v = *reinterpret_cast<const float4*>(p + 4 * s); acc += v.x + v.y + v.z + v.w; r = __float_as_uint(v.w); }
                if (W == 2) { float2 v = *reinterpret_cast<const float2*>(p + 4 * s); acc += v.x + v.y; r = __float_as_uint(v.y); }
                if (W == 1) { float v = p[4 * s]; acc += v; r = __float_as_uint(v); }
            }
            r = r % nrec;  // dependent chain like a traversal
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const unsigned nrec = 1u << 20;  // 64 MB of 64-byte records
    std::vector<float> h((size_t)nrec * 16);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { unsigned v = (unsigned)rand() % nrec; h[i] = *reinterpret_cast<float*>(&v); }
    // make the values small-denormal-safe: they are only reinterpreted as integers
    std::vector<unsigned> hidx(nrec);
    for (unsigned i = 0; i < nrec; ++i) hidx[i] = (unsigned)rand() % nrec;
    float *d, *o; unsigned* di;
    hipMalloc(&d, h.size() * 4); hipMalloc(&di, nrec * 4);
    const int blocks = 256 * 5, threads = 256;
    hipMalloc(&o, blocks * threads * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(di, hidx.data(), nrec * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 64;
    struct M { const char* name; unsigned lo, hi; int n; } masks[] = {
        {"64 lanes", 0xFFFFFFFFu, 0xFFFFFFFFu, 64}, {"32 lanes (low half)", 0xFFFFFFFFu, 0u, 32},
        {"32 lanes (even)", 0x55555555u, 0x55555555u, 32}, {"8 lanes", 0x01010101u, 0x01010101u, 8}, {"1 lane", 1u, 0u, 1}};
    auto run = [&](auto kern, const char* kn, int loads_per_iter, int bytes) {
        for (auto& m : masks) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, di, nrec, iters, m.lo, m.hi, o);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, di, nrec, iters, m.lo, m.hi, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double wave_instr = (double)blocks * (threads / 64) * iters * loads_per_iter;
            const double cyc_per_instr_per_cu = ms * 1e-3 * 2.4e9 / (wave_instr / 256.0);
            printf("%-22s %-20s %8.3f ms  %7.1f cycles/VMEM-instr/CU  %7.2f GB/s useful\n", kn, m.name, ms, cyc_per_instr_per_cu,
                   (double)blocks * threads * (m.n / 64.0) * iters * loads_per_iter * bytes / (ms * 1e-3) / 1e9);
        }
    };
    run(k_gather<4, 4>, "x4, 4 loads/rec(64B)", 4, 16);
    run(k_gather<4, 2>, "x4, 2 loads/rec(32B)", 2, 16);
    run(k_gather<4, 1>, "x4, 1 load/rec(16B)", 1, 16);
    run(k_gather<2, 1>, "x2, 1 load/rec(8B)", 1, 8);
    run(k_gather<1, 1>, "x1, 1 load/rec(4B)", 1, 4);
    return 0;
}
