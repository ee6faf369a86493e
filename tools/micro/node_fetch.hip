// node_fetch.hip -- what does the vector-memory pipe charge k_trace_inst for a node fetch, and which lane pattern is cheap?
// The wide walk issues 1.01 M dwordx4 wave-loads per frame = 3 957 per CU in 117.6 us (profiles/r06_bvh_sq.txt): one every
// ~68 cycles per CU, and with 12 waves per CU instead of 8 a trip takes 1.6x as long (r06_bvh_grid_sweep.txt): the signature of
// a throughput bound on the CU's address path, not of latency.  This probe walks a heap-ordered tree of 128-byte nodes
// (2^LEVELS - 1 of them), every trip's node index depending on the previous trip's data, in four lane patterns:
//   0  lane per ray, 8 x dwordx4 of its own node (the wide walk as shipped: 64 different lines per instruction)
//   1  lane per ray, 4 x dwordx4 of a 64-byte node (the binary walk)
//   2  FOUR lanes per ray: lane c of the quad loads lo[c] and hi[c] of the node (2 x dwordx4; a quad reads 64 contiguous bytes)
//   3  lane per ray, but the node comes in through LDS: eight lanes fetch one node's eight quarters (128 contiguous bytes per
//      eight lanes, eight instructions for the wave's 64 nodes), LDS transposes them to their owners
// COHERENT: the rays of a wave share their path down to level `split`, below it every ray goes its own way.
// Reported: wave-trips per us, RAY-trips per us (what counts), cycles per load instruction and CU.
// build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Node { float4 q[8]; };

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_walk(const Node *__restrict__ nodes, uint32_t levels, uint32_t split, uint32_t iters,
                                              uint32_t chain, float *__restrict__ out)
{
    __shared__ float4 s_t[MODE == 3 ? 4 : 1][MODE == 3 ? 64 * 9 : 1];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4u + w;
    const uint32_t ray_in_wave = MODE == 2 ? lane >> 2 : MODE == 4 ? lane >> 1 : lane, c = lane & 3u;
    const uint32_t wave_seed = mix(wave * 0x9E3779B9u + 1u), ray_seed = mix(wave_seed ^ (ray_in_wave * 0x85EBCA6Bu + 7u));
    uint32_t cur = 0, lev = 0, path = 0;
    // the path's turns: the wave's above `split`, the ray's own below (hashed once per walk from the root, not per trip)
    const uint32_t keep = split >= 32u ? 0xFFFFFFFFu : (1u << split) - 1u;
    uint32_t turns = (wave_seed & keep) | (ray_seed & ~keep);
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t r;   // the word of the node the next step depends on
        if (MODE == 0) {
            const float4 *nd = nodes[cur].q;
            const float4 a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3], a4 = nd[4], a5 = nd[5], a6 = nd[6], a7 = nd[7];
            acc += (a0.x + a1.y) + (a2.z + a3.x) + (a4.y + a5.z) + (a6.x + a7.y);
            r = __float_as_uint(a0.w);
        } else if (MODE == 1) {
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + (size_t)cur * 64u);
            const float4 a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3];
            acc += (a0.x + a1.y) + (a2.z + a3.x);
            r = __float_as_uint(a0.w);
        } else if (MODE == 2) {
            const float4 *nd = nodes[cur].q;
            const float4 lo = nd[c], hi = nd[4 + c];
            acc += lo.x + hi.y;
            r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(lo.w), 0x00 /* quad_perm 0,0,0,0 */, 0xf, 0xf, false);
        } else if (MODE == 4) {   // two lanes per ray: lane p of the pair takes the slots 2p, 2p + 1 (four loads)
            const float4 *nd = nodes[cur].q;
            const uint32_t p2 = 2u * (lane & 1u);
            const float4 l0 = nd[p2], l1 = nd[p2 + 1u], h0 = nd[4u + p2], h1 = nd[5u + p2];
            acc += (l0.x + h0.y) + (l1.z + h1.x);
            r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(l0.w), 0xA0 /* quad_perm 0,0,2,2 */, 0xf, 0xf, false);
        } else {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t n = (uint32_t)__shfl((int)cur, 8 * k + (int)(lane >> 3));
                v[k] = nodes[n].q[lane & 7u];
            }
            float4 *t = s_t[w];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[(8u * k + (lane >> 3)) * 9u + (lane & 7u)] = v[k];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float4 a0 = t[lane * 9u], a1 = t[lane * 9u + 1], a2 = t[lane * 9u + 2], a3 = t[lane * 9u + 3], a4 = t[lane * 9u + 4],
                         a5 = t[lane * 9u + 5], a6 = t[lane * 9u + 6], a7 = t[lane * 9u + 7];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            acc += (a0.x + a1.y) + (a2.z + a3.x) + (a4.y + a5.z) + (a6.x + a7.y);
            r = __float_as_uint(a0.w);
        }
        // down one level: the wave's path above `split`, the ray's own below; at the bottom back to the root on a new path
        // the trip's dependent arithmetic (the slab tests, the ordering, the stack): `chain` fused multiply-adds in a row
        float z = __uint_as_float((r & 0x007FFFFFu) | 0x3F800000u);
        for (uint32_t k = 0; k < chain; ++k) z = fmaf(z, 0.999f, 0.001f);
        acc += z;
        const uint32_t bit = ((turns >> lev) ^ (r >> 7) ^ (z > 3.0f ? 1u : 0u)) & 1u;
        ++lev;
        if (lev >= levels) { lev = 0; cur = 0; ++path; turns = mix(turns + path); turns = (mix(wave_seed + path) & keep) | (turns & ~keep); }
        else cur = 2u * cur + 1u + bit;
    }
    if (acc == 12345.678f) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
int run(const Node *d, uint32_t levels, uint32_t split, uint32_t bpc, uint32_t iters, uint32_t chain, float *dout)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t blocks = 256u * bpc;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipExtLaunchKernelGGL((k_walk<MODE>), dim3(blocks), dim3(256), 0, 0, e0, e1, 0, d, levels, split, iters, chain, dout);
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double us = best * 1e3, waves = 4.0 * blocks, rays = MODE == 2 ? 16.0 : MODE == 4 ? 32.0 : 64.0;
    const double loads_per_trip = MODE == 0 ? 8 : MODE == 1 ? 4 : MODE == 2 ? 2 : MODE == 4 ? 4 : 8;
    const double trip_us = us / iters;
    // cycles of the CU's memory pipe per load instruction if it were the only bound (2.3 GHz; waves per CU = 4 bpc)
    const double cyc_per_load = trip_us * 2300.0 / (4.0 * bpc * loads_per_trip);
    printf("mode %d chain %3u levels %2u split %2u waves/SIMD %u: %8.1f us, trip %.3f us, ray-trips/us %9.0f, cycles per load instr and CU %.1f\n", MODE, chain, levels,
           split, bpc, us, trip_us, waves * rays * iters / us, cyc_per_load);
    return 0;
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 400u;
    const uint32_t max_levels = 20;
    const size_t n = ((size_t)1 << max_levels) - 1;
    std::vector<Node> h(n);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) {
            s = s * 1664525u + 1013904223u;
            h[i].q[k] = make_float4((float)(s & 1023u), (float)((s >> 10) & 1023u), (float)((s >> 20) & 1023u), 0.f);
            uint32_t w = s ^ (uint32_t)i * 2654435761u;
            h[i].q[k].w = *reinterpret_cast<float *>(&w);
        }
    Node *d = nullptr;
    float *dout = nullptr;
    CK(hipMalloc(&d, n * sizeof(Node)));
    CK(hipMalloc(&dout, 64));
    CK(hipMemcpy(d, h.data(), n * sizeof(Node), hipMemcpyHostToDevice));
    // levels 20 = 128 MB of nodes (Infinity Cache), 14 = 2 MB (every XCD's L2); split: how far down a wave's rays agree
    // levels 20 = 128 MB of nodes (Infinity Cache); split: how far down a wave's rays agree.  chain: the dependent arithmetic
    // of a trip -- 200 for the lane-per-ray wide walk (four slab tests, three comparators, pushes), 100 binary, 60 / 110 for a
    // quad / pair of lanes per ray (one / two slab tests per lane, the ordering by DPP); 0 = the memory side alone
    const uint32_t cases[][2] = {{20, 16}, {20, 14}, {20, 12}};
    for (auto &cs : cases)
        for (uint32_t bpc : {2u, 4u, 6u, 8u}) {
            if (run<0>(d, cs[0], cs[1], bpc, iters, 0u, dout) || run<0>(d, cs[0], cs[1], bpc, iters, 200u, dout)) return 1;
            if (run<1>(d, cs[0], cs[1], bpc, iters, 0u, dout) || run<1>(d, cs[0], cs[1], bpc, iters, 100u, dout)) return 1;
            if (run<4>(d, cs[0], cs[1], bpc, iters, 0u, dout) || run<4>(d, cs[0], cs[1], bpc, iters, 110u, dout)) return 1;
            if (run<2>(d, cs[0], cs[1], bpc, iters, 0u, dout) || run<2>(d, cs[0], cs[1], bpc, iters, 60u, dout)) return 1;
        }
    return 0;
}
