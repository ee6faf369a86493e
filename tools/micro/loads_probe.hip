// loads_probe.hip -- how long does the LOAD phase of k_project take (index load -> vertex gather of every triangle of a
// 1 000 000-triangle grid mesh, nothing else), and what does it respond to?  k_project's own "loads only" form
// (LS_PROJECT_DEBUG=1) takes 6.0 us: 50 MB from L2 / Infinity Cache = 8.3 TB/s.  Is that bandwidth, or latency x rounds
// (15 628 waves of 64 triangles over 8 192 wave slots, two dependent round trips each)?  SETS triangles per lane with all
// their loads issued up front = SETS x the bytes in flight per wave and 1 / SETS the waves.
// build: hipcc --offload-arch=gfx950 -O3 -o loads_probe loads_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

template <int SETS, bool SPREAD>
__global__ __launch_bounds__(256) void k_loads(const float *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t ntris,
                                               float *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t n_waves = (ntris + 64u * SETS - 1u) / (64u * SETS), rank = blockIdx.x * 4u + w;
    uint32_t a[SETS], b[SETS], c[SETS], k[SETS];
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        // set s of wave `rank`: 64 triangles; SPREAD: eight runs of eight, a stride of the wave count apart (k_project's order)
        const uint32_t wave = rank * SETS + s, nw = n_waves * SETS;
        k[s] = SPREAD ? ((lane >> 3) * nw + wave) * 8u + (lane & 7u) : wave * 64u + lane;
        if (rank >= n_waves || k[s] >= ntris) k[s] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int s = 0; s < SETS; ++s)
        if (k[s] != 0xFFFFFFFFu) { a[s] = idx[3 * (size_t)k[s]]; b[s] = idx[3 * (size_t)k[s] + 1]; c[s] = idx[3 * (size_t)k[s] + 2]; }
    float acc = 0.f;
    float r[SETS][9];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
        if (k[s] != 0xFFFFFFFFu) {
            const float *pa = verts + 3 * (size_t)a[s], *pb = verts + 3 * (size_t)b[s], *pc = verts + 3 * (size_t)c[s];
            r[s][0] = pa[0]; r[s][1] = pa[1]; r[s][2] = pa[2]; r[s][3] = pb[0]; r[s][4] = pb[1]; r[s][5] = pb[2];
            r[s][6] = pc[0]; r[s][7] = pc[1]; r[s][8] = pc[2];
        }
#pragma unroll
    for (int s = 0; s < SETS; ++s)
        if (k[s] != 0xFFFFFFFFu)
#pragma unroll
            for (int i = 0; i < 9; ++i) acc += r[s][i];
    if (acc == 12345.678f) out[0] = acc;   // (never: keeps the loads alive)
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SETS, bool SPREAD>
int run(const float *dv, const uint32_t *di, uint32_t ntris, float *dout)
{
    const uint32_t n_waves = (ntris + 64u * SETS - 1u) / (64u * SETS), blocks = (n_waves + 3u) / 4u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 30; ++rep) {
        hipExtLaunchKernelGGL((k_loads<SETS, SPREAD>), dim3(blocks), dim3(256), 0, 0, e0, e1, 0, dv, di, ntris, dout);
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 5) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("sets %d per lane, %s: %6u waves  kernel %.2f us (min %.2f)\n", SETS, SPREAD ? "spread runs" : "contiguous ", n_waves, sum / 25 * 1e3, best * 1e3);
    return 0;
}

int main()
{
    const int cx = 1000, cy = 500;   // SYN-1M's topology (lidarshooter_amd/synth.py: grid_mesh)
    std::vector<float> v((size_t)(cx + 1) * (cy + 1) * 3);
    for (int j = 0; j <= cy; ++j)
        for (int i = 0; i <= cx; ++i) { float *p = &v[((size_t)j * (cx + 1) + i) * 3]; p[0] = -50.f + 0.1f * i; p[1] = -50.f + 0.2f * j; p[2] = 0.01f * ((i * 7 + j * 13) % 17); }
    std::vector<uint32_t> t((size_t)cx * cy * 6);
    for (int j = 0; j < cy; ++j)
        for (int i = 0; i < cx; ++i) {
            const uint32_t v00 = j * (cx + 1) + i, v10 = v00 + 1, v01 = v00 + cx + 1, v11 = v01 + 1;
            uint32_t *q = &t[((size_t)j * cx + i) * 6];
            q[0] = v00; q[1] = v10; q[2] = v11; q[3] = v00; q[4] = v11; q[5] = v01;
        }
    const uint32_t ntris = cx * cy * 2;
    float *dv, *dout;
    uint32_t *di;
    CK(hipMalloc(&dv, v.size() * 4));
    CK(hipMalloc(&di, t.size() * 4));
    CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(di, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    if (run<1, false>(dv, di, ntris, dout) || run<1, true>(dv, di, ntris, dout) || run<2, true>(dv, di, ntris, dout) || run<4, true>(dv, di, ntris, dout) ||
        run<4, false>(dv, di, ntris, dout) || run<8, true>(dv, di, ntris, dout))
        return 1;
    return 0;
}
