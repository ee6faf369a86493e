// ls_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the LiDAR tracer hot path.
//
// Compiled with -ffp-contract=off: every float operation rounds once and fused multiply-adds are
// explicit fmaf() calls, so the arithmetic below is the same operation sequence as the CPU oracle
// (oracle/ls_oracle.c) and ids/t can be compared bit for bit.
//
// Reference functions replaced (paths relative to /root/reference/ros_ws/src/lidarshooter/src/):
//   k_transform ........ MeshTransformer::transformIntoBuffer (MeshTransformer.cpp:142-205) +
//                        LidarDevice::originToSensor (LidarDevice.cpp:383-391)
//   k_morton/k_leaves/k_range_*/k_hierarchy ... rtcCommitScene (EmbreeTracer.cpp:290-295) /
//                        OptixTracer::buildAccelStructure (OptixTracer.cpp:517-571)
//   k_trace ............ LidarDevice::nextRay16 (LidarDevice.cpp:294-342) + rtcIntersect16
//                        (EmbreeTracer.cpp:472-480) / allRaysGPUKernel + __raygen__rg
//                        (LidarDeviceKernels.cu:25-52, OptixTracerModules.cu:26-72)
//   k_pack ............. XYZIRBytes::addToCloud (XYZIRBytes.cpp:24-40) / addPointsToCloud
//                        (OptixTracer.cpp:895-942)
#include "ls_kernels.h"
#include "ls_device.h"
#include "ls_launch.h"
#include "ls_tuning.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>


namespace ls {

namespace {



// ------------------------------------------------------------------------------------------
// Vertex transform into the sensor frame:  p' = Rinv * ((A * v) - t).  One thread per vertex;
// also folds max |coordinate| of the scene into *maxabs (bits of a non-negative float).
// Algorithmic bytes per vertex: stride (read) + 12 (write).
// ------------------------------------------------------------------------------------------
constexpr int kTransformBlock = 1024;   // sixteen waves share one atomic
__global__ __launch_bounds__(kTransformBlock) void k_transform(const uint8_t *__restrict__ raw, uint32_t stride, uint32_t n,
                                                               Affine m, float *__restrict__ out,
                                                               uint32_t *__restrict__ maxabs)
{
    __shared__ float s_max[kTransformBlock / 64];
    float mx = 0.0f;
    for (uint32_t j = blockIdx.x * kTransformBlock + threadIdx.x; j < n; j += gridDim.x * kTransformBlock) {
        const V3 o = xform_vertex(m, raw + (size_t)j * stride);
        out[3 * (size_t)j + 0] = o.x;
        out[3 * (size_t)j + 1] = o.y;
        out[3 * (size_t)j + 2] = o.z;
        mx = fmaxf(mx, fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z))));
    }
    // one atomic per block of 1 024 threads (a single hot address sustains only ~90 atomics/us chip-wide: the 512
    // atomics of 512 blocks of 256 were 5.7 of this kernel's 9.5 us)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x < 64) {
        mx = threadIdx.x < kTransformBlock / 64 ? s_max[threadIdx.x] : 0.0f;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (threadIdx.x == 0 && mx > 0.0f) atomicMax(maxabs, __float_as_uint(mx));
    }
}

__global__ __launch_bounds__(kBlock) void k_quads_to_triangles(const uint32_t *__restrict__ q, uint32_t n, uint32_t *__restrict__ t)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t v0 = q[4 * (size_t)i], v1 = q[4 * (size_t)i + 1], v2 = q[4 * (size_t)i + 2], v3 = q[4 * (size_t)i + 3];
    uint32_t *o = t + 6 * (size_t)i;
    o[0] = v0; o[1] = v1; o[2] = v3;
    o[3] = v2; o[4] = v3; o[5] = v1;
}

// Largest vertex index of a geometry's triangles (atomicMax into *out, zeroed by the launcher): checked against the
// geometry's vertex count at the next commit.  The reference hands Embree shared buffers unchecked
// (EmbreeTracer.cpp:140-176: a bad index reads host memory out of bounds -- undefined, usually survivable); here every
// kernel that gathers vertices would take a device memory fault, and with it the process, so such a mesh is refused
// before anything is launched over it.  12 MB of indices at a million triangles: a few microseconds, once per index upload.
__global__ __launch_bounds__(kBlock) void k_index_max(const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ out)
{
    uint32_t m = 0;
    // sixteen bytes per load, four loads in flight per lane (a 256-workgroup grid: 256 atomics on the one word), when the
    // array is aligned for it -- the library's own copies are; a caller's array handed over in place need not be
    const uint32_t n4 = (reinterpret_cast<uintptr_t>(idx) & 15u) == 0 ? n / 4u : 0u;
    const uint4 *__restrict__ idx4 = reinterpret_cast<const uint4 *>(idx);
    const uint32_t stride = gridDim.x * kBlock;
    uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    for (; j + 3u * stride < n4; j += 4u * stride) {
        const uint4 a = idx4[j], b = idx4[j + stride], c = idx4[j + 2u * stride], d = idx4[j + 3u * stride];
        m = max(max(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w))), max(m, max(max(max(c.x, c.y), max(c.z, c.w)), max(max(d.x, d.y), max(d.z, d.w)))));
    }
    for (; j < n4; j += stride) { const uint4 a = idx4[j]; m = max(m, max(max(a.x, a.y), max(a.z, a.w))); }
    for (j = 4u * n4 + blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) m = max(m, idx[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    __shared__ uint32_t s_m[kBlock / 64];
    if ((threadIdx.x & 63u) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
}

__global__ __launch_bounds__(kBlock) void k_rebase(const uint32_t *__restrict__ idx, uint32_t n, uint32_t vbase,
                                                   uint32_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j < n) out[j] = idx[j] + vbase;
}

// ------------------------------------------------------------------------------------------
// Sensor-centred sign-magnitude Morton keys (30 bits): [sx sy sz | 9-bit |x|,|y|,|z| interleaved].
// Every ray starts at the sensor origin, so (a) a ray stays inside one sign octant and (b) inside
// an octant the Morton order of |coordinate| is front-to-back for EVERY ray: the left-to-right
// depth-first order of the radix tree is the traversal order, which is what lets the trace
// kernel run stackless on skip links.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMortonBlock = 1024;
__device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// A workgroup takes one tile of the sort (kSortTile keys, four per thread); COUNT: it also leaves the tile's histogram of
// the first digit where the sort's first pass looks for it (the pass's counting launch -- 5 us -- is not run).
template <bool COUNT>
__global__ __launch_bounds__(kMortonBlock) void k_morton(const float *__restrict__ verts, const uint32_t *__restrict__ tris,
                                                         uint32_t ntris, const uint32_t *__restrict__ maxabs,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                         uint32_t *__restrict__ counts)
{
    static_assert(kSortDigits == kMortonBlock && kSortTile % kMortonBlock == 0, "a thread per digit");
    __shared__ uint32_t s_cnt[kSortDigits];
    if (COUNT) {
        s_cnt[threadIdx.x] = 0u;
        __syncthreads();
    }
    const float m = __uint_as_float(*maxabs);
    const float scale = m > 0.0f ? 512.0f / m : 0.0f;
#pragma unroll
    for (uint32_t r = 0; r < kSortTile / kMortonBlock; ++r) {
        const uint32_t k = blockIdx.x * kSortTile + r * kMortonBlock + threadIdx.x;
        if (k < ntris) {
            const uint32_t i0 = tris[3 * (size_t)k + 0], i1 = tris[3 * (size_t)k + 1], i2 = tris[3 * (size_t)k + 2];
            uint32_t key = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float p0 = verts[3 * (size_t)i0 + a], p1 = verts[3 * (size_t)i1 + a], p2 = verts[3 * (size_t)i2 + a];
                const float c = 0.5f * (fminf(p0, fminf(p1, p2)) + fmaxf(p0, fmaxf(p1, p2)));
                const uint32_t q = min(511u, (uint32_t)(fabsf(c) * scale));
                key |= expand_bits(q) << (2 - a);
                key |= (c < 0.0f ? 1u : 0u) << (29 - a);
            }
            keys[k] = key;
            if (vals) vals[k] = k;
            if (COUNT) atomicAdd(&s_cnt[key & (kSortDigits - 1u)], 1u);
        }
    }
    if (COUNT) {
        __syncthreads();
        counts[(size_t)blockIdx.x * kSortDigits + threadIdx.x] = s_cnt[threadIdx.x];
    }
}

// ------------------------------------------------------------------------------------------
// Aligned-range tree: entry j of level l bounds leaves [j<<l, (j+1)<<l).  k_leaves_tree (below) builds
// levels 1..9 for 512 leaves per block in LDS; k_range_top finishes the (few) remaining levels
// in one block.  No atomics, no cross-workgroup hand-off.
// ------------------------------------------------------------------------------------------
struct Box { float lo[3], hi[3]; };

__device__ __forceinline__ Box box_empty() { return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; }
__device__ __forceinline__ void box_merge(Box &a, const Box &b)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) { a.lo[i] = fminf(a.lo[i], b.lo[i]); a.hi[i] = fmaxf(a.hi[i], b.hi[i]); }
}
__device__ __forceinline__ Box load_box(const float4 *boxes, uint32_t e)
{
    const float4 a = boxes[2 * (size_t)e], b = boxes[2 * (size_t)e + 1];
    return {{a.x, a.y, a.z}, {b.x, b.y, b.z}};
}
__device__ __forceinline__ void store_box(float4 *boxes, uint32_t e, const Box &b)
{
    boxes[2 * (size_t)e] = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.0f);
    boxes[2 * (size_t)e + 1] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.0f);
}

constexpr int kBottomLevels = 9;  // 512 leaves per block

// the levels above k_leaves_tree's nine, one workgroup: through global memory while a level has more than kTopLds entries
// (a round trip and a barrier per level), then -- 977 entries at a million leaves -- in LDS, the stores to the tree
// fire-and-forget (ten levels: 9.5 -> 3 us)
constexpr uint32_t kTopLds = 1024;
__global__ __launch_bounds__(kBlock) void k_range_top(RangeTree rt, float4 *__restrict__ boxes)
{
    __shared__ float s_a[6][kTopLds], s_b[6][kTopLds / 2];
    uint32_t l = kBottomLevels + 1;
    for (; l < rt.levels && rt.count[l] > kTopLds; ++l) {
        for (uint32_t j = threadIdx.x; j < rt.count[l]; j += kBlock) {
            Box b = load_box(boxes, rt.offset[l - 1] + 2 * j);
            if (2 * j + 1 < rt.count[l - 1]) { const Box c = load_box(boxes, rt.offset[l - 1] + 2 * j + 1); box_merge(b, c); }
            store_box(boxes, rt.offset[l] + j, b);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (l >= rt.levels) return;
    for (uint32_t j = threadIdx.x; j < rt.count[l]; j += kBlock) {   // the first level that fits: from global memory
        Box b = load_box(boxes, rt.offset[l - 1] + 2 * j);
        if (2 * j + 1 < rt.count[l - 1]) { const Box c = load_box(boxes, rt.offset[l - 1] + 2 * j + 1); box_merge(b, c); }
        store_box(boxes, rt.offset[l] + j, b);
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_a[k][j] = b.lo[k]; s_a[3 + k][j] = b.hi[k]; }
    }
    __syncthreads();
    for (++l; l < rt.levels; ++l) {
        const uint32_t below = rt.count[l - 1];
        for (uint32_t j = threadIdx.x; j < rt.count[l]; j += kBlock) {
            Box b;
#pragma unroll
            for (int k = 0; k < 3; ++k) { b.lo[k] = s_a[k][2 * j]; b.hi[k] = s_a[3 + k][2 * j]; }
            if (2 * j + 1 < below) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { b.lo[k] = fminf(b.lo[k], s_a[k][2 * j + 1]); b.hi[k] = fmaxf(b.hi[k], s_a[3 + k][2 * j + 1]); }
            }
            store_box(boxes, rt.offset[l] + j, b);
#pragma unroll
            for (int k = 0; k < 3; ++k) { s_b[k][j] = b.lo[k]; s_b[3 + k][j] = b.hi[k]; }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < rt.count[l]; j += kBlock) {
#pragma unroll
            for (int k = 0; k < 6; ++k) s_a[k][j] = s_b[k][j];
        }
        __syncthreads();
    }
}

// Triangle records + leaf boxes, one thread per pair of Morton-sorted positions.
// Leaf k = records [k*g, k*g+g); its box (padded, see below) is entry k of range-tree level 0.
// Bytes per triangle: 4 (sorted id) + 12 (indices) + 36 (vertices) read, 48 written, + 64/g (leaf box + the tree above it).
// One triangle: record p written, the triangle's fattened bounds returned (empty past the end).
template <bool MESH_RECORDS>
__device__ __forceinline__ Box leaf_triangle(const float *__restrict__ verts, const uint32_t *__restrict__ tris, uint32_t gid, bool have,
                                             uint32_t p, TriRecord *__restrict__ records)
{
    Box bx = box_empty();
    if (have) {
        const uint32_t i0 = tris[3 * (size_t)gid + 0], i1 = tris[3 * (size_t)gid + 1], i2 = tris[3 * (size_t)gid + 2];
        const V3 v0 = {verts[3 * (size_t)i0], verts[3 * (size_t)i0 + 1], verts[3 * (size_t)i0 + 2]};
        const V3 v1 = {verts[3 * (size_t)i1], verts[3 * (size_t)i1 + 1], verts[3 * (size_t)i1 + 2]};
        const V3 v2 = {verts[3 * (size_t)i2], verts[3 * (size_t)i2 + 1], verts[3 * (size_t)i2 + 2]};
        const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
        const V3 Ng = cross_fma(e2, e1);
        const float NgC = dot_fma(Ng, v0);
        float4 *r = reinterpret_cast<float4 *>(records + p);
        if (MESH_RECORDS) {   // the corners as given: the trace kernel carries them into the sensor frame of its frame
            r[0] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(gid));
            r[1] = make_float4(v1.x, v1.y, v1.z, 0.0f);
            r[2] = make_float4(v2.x, v2.y, v2.z, 0.0f);
        } else {
            r[0] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(gid));
            r[1] = make_float4(e1.x, e1.y, e1.z, NgC);
            r[2] = make_float4(e2.x, e2.y, e2.z, 0.0f);
        }
        // The triangle test accepts rays that miss the exact triangle by rounding error, so boxes
        // are fattened by 2^-16 of the largest |coordinate|: BVH result == exhaustive result.
        const float m = fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v1.x))),
                              fmaxf(fmaxf(fmaxf(fabsf(v1.y), fabsf(v1.z)), fmaxf(fabsf(v2.x), fabsf(v2.y))),
                                    fabsf(v2.z)));
        const float pad = m * 0x1p-16f;
        bx.lo[0] = fminf(v0.x, fminf(v1.x, v2.x)) - pad; bx.hi[0] = fmaxf(v0.x, fmaxf(v1.x, v2.x)) + pad;
        bx.lo[1] = fminf(v0.y, fminf(v1.y, v2.y)) - pad; bx.hi[1] = fmaxf(v0.y, fmaxf(v1.y, v2.y)) + pad;
        bx.lo[2] = fminf(v0.z, fminf(v1.z, v2.z)) - pad; bx.hi[2] = fmaxf(v0.z, fmaxf(v1.z, v2.z)) + pad;
    }
    return bx;
}

// ------------------------------------------------------------------------------------------
// k_leaves + k_range_bottom in one launch (round 4: a launch in a chain of dependent launches costs ~5 us before it
// does anything, and k_range_bottom read back the 32 MB of leaf boxes k_leaves had just written: 23.5 + 13.4 -> 29.1 us).
// A workgroup takes 512 leaves (512 g triangles, 512 at a time, two per thread with both gathers in flight), writes
// their records and leaf boxes, keeps the boxes in LDS and reduces the nine levels above them there.  The levels above
// need every workgroup's top entry: k_range_top, its own launch (7.7 us).  Handing them to the workgroup that finishes
// last was built and is slower: a release fence per workgroup writes the whole L2 back (1 954 of them: 328 us), and with
// the entries stored past the L2 instead, the 1 954 tickets on one address cost 21 us (~90 atomics per us on one word).
// ------------------------------------------------------------------------------------------
template <bool MESH_RECORDS>
__global__ __launch_bounds__(kBlock) void k_leaves_tree(const float *__restrict__ verts, const uint32_t *__restrict__ tris,
                                                        const uint32_t *__restrict__ sorted_vals, uint32_t ntris, uint32_t g,
                                                        TriRecord *__restrict__ records, RangeTree rt, float4 *__restrict__ boxes)
{
    __shared__ float s[6][512];
    const uint32_t t = threadIdx.x;
    const uint32_t leaf0 = blockIdx.x * 512u;
    const uint32_t lg = g == 1u ? 0u : g == 2u ? 1u : g == 4u ? 2u : 3u;
    for (uint32_t round = 0; round < g; ++round) {
        const uint32_t pa = (leaf0 << lg) + round * 512u + t, pb = pa + kBlock;
        const bool ha = pa < ntris, hb = pb < ntris;
        const uint32_t ga = ha ? sorted_vals[pa] : 0u, gb = hb ? sorted_vals[pb] : 0u;
        Box a = leaf_triangle<MESH_RECORDS>(verts, tris, ga, ha, pa, records);
        Box b = leaf_triangle<MESH_RECORDS>(verts, tris, gb, hb, pb, records);
        for (uint32_t off = 1; off < g; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a.lo[k] = fminf(a.lo[k], __shfl_xor(a.lo[k], off)); a.hi[k] = fmaxf(a.hi[k], __shfl_xor(a.hi[k], off));
                b.lo[k] = fminf(b.lo[k], __shfl_xor(b.lo[k], off)); b.hi[k] = fmaxf(b.hi[k], __shfl_xor(b.hi[k], off));
            }
        }
        if ((t & (g - 1u)) == 0u) {
            const uint32_t ka = pa >> lg, kb = pb >> lg;   // leaves
            if (ha) store_box(boxes, ka, a);
            if (hb) store_box(boxes, kb, b);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s[k][ka - leaf0] = a.lo[k]; s[3 + k][ka - leaf0] = a.hi[k];
                s[k][kb - leaf0] = b.lo[k]; s[3 + k][kb - leaf0] = b.hi[k];
            }
        }
    }
    __syncthreads();
    Box b;
#pragma unroll
    for (int k = 0; k < 3; ++k) { b.lo[k] = fminf(s[k][2 * t], s[k][2 * t + 1]); b.hi[k] = fmaxf(s[3 + k][2 * t], s[3 + k][2 * t + 1]); }
    __syncthreads();
    for (uint32_t l = 1; l <= (uint32_t)kBottomLevels && l < rt.levels; ++l) {
        const uint32_t n = 512u >> l;  // entries of this level owned by the block
        if (t < n) {
            const uint32_t j = (leaf0 >> l) + t;
            if (j < rt.count[l]) store_box(boxes, rt.offset[l] + j, b);
#pragma unroll
            for (int k = 0; k < 3; ++k) { s[k][t] = b.lo[k]; s[3 + k][t] = b.hi[k]; }
        }
        __syncthreads();
        if (t < (n >> 1)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = fminf(s[k][2 * t], s[k][2 * t + 1]);
                b.hi[k] = fmaxf(s[3 + k][2 * t], s[3 + k][2 * t + 1]);
            }
        }
        __syncthreads();
    }
}

// delta(i, j): length of the common prefix of leaf keys i and j (equal keys: extended by the indices), -1 outside [0, L).
__device__ __forceinline__ int delta_of(int i, uint32_t ki, int j, uint32_t kj)
{
    const uint32_t x = ki ^ kj;
    return x ? __clz(x) : 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

// A workgroup keeps the keys of the leaves within kKeyWindow of its own 256 in LDS.  A lane's two searches (Karras' range
// end and split) run on that copy; the lane whose node reaches beyond it -- one node in five hundred -- gives up (far) and
// the WAVE repeats its searches on global memory, 64 probes at a time (coop_search): three or four memory round trips for
// the root of a million leaves where the lane alone needed forty, one after the other (38 of this kernel's 68 us were the
// chains of its few long-range nodes: every later probe of a bisection lands near the far end, outside any window).
constexpr int kKeyWindow = 512;
struct KeyWindow {
    const uint32_t *lds;    // keys of leaves [first, first + n)
    int first, n;
};
__device__ __forceinline__ int delta_win(const KeyWindow &kw, int L, int i, uint32_t ki, int j, bool &far)
{
    if (j < 0 || j >= L) return -1;
    const uint32_t at = (uint32_t)(j - kw.first);
    if (at >= (uint32_t)kw.n) { far = true; return -1; }
    return delta_of(i, ki, j, kw.lds[at]);
}

// the largest t in [lo, hi) with delta(i, i + t d) > thresh, given that it holds at lo, fails at hi and is monotone in
// between; all 64 lanes of the wave work on the one node (arguments uniform)
// A far node is worked on by a GROUP of kFarLanes lanes (LS_FAR_LANES: 64 = the whole wave, one node after the other, as in
// round 3; 16 = four nodes at a time, a quarter of the probes per round).  More probes per round were measured and are
// slower (two per lane 65.2, four 69.5 us against 63.5): what the far nodes cost is their probes -- every one a 64-byte
// line of the key array for four bytes -- not the number of rounds.
#ifndef LS_FAR_LANES
#define LS_FAR_LANES 16
#endif
constexpr int kFarLanes = LS_FAR_LANES;
__device__ __forceinline__ int group_count(bool p, uint32_t lane)
{
    const unsigned long long b = __ballot(p);
    if (kFarLanes == 64) return (int)__popcll(b);
    const uint32_t q = lane / (uint32_t)kFarLanes;
    return (int)__popcll((b >> (q * (uint32_t)kFarLanes)) & ((1ull << (kFarLanes & 63)) - 1ull));
}
// all groups of the wave call this together (a group without a node: lo = hi = 0)
__device__ __forceinline__ int coop_search(const uint32_t *__restrict__ keys, uint32_t g, int L, int i, uint32_t ki, int d, int thresh,
                                           int lo, int hi, uint32_t lane)
{
    const int sub = (int)(lane % (uint32_t)kFarLanes);
    while (__any(hi - lo > 1)) {
        const int step = max(1, (hi - lo + kFarLanes - 1) / kFarLanes);
        const int t = lo + (sub + 1) * step;
        bool p = false;
        if (t < hi) {
            const int j = i + t * d;
            p = j >= 0 && j < L && delta_of(i, ki, j, keys[(size_t)j * g]) > thresh;
        }
        const int c = group_count(p, lane);   // monotone: the lanes that hold are the first c of the group
        hi = min(hi, lo + (c + 1) * step);
        lo = lo + c * step;
    }
    return lo;
}

// bounds of leaves [l, r] from the aligned-range tree: level v contributes entry a_v = ceil(l / 2^v) if that is odd and
// entry e_v - 1, e_v = floor((r + 1) / 2^v), if e_v is odd, as long as a_v < e_v
__device__ __forceinline__ Box coop_range_box(const RangeTree &rt, const float4 *__restrict__ boxes, uint32_t l, uint32_t r, uint32_t lane, bool have)
{
    Box b = box_empty();
    const uint32_t sub = lane % (uint32_t)kFarLanes;
#pragma unroll
    for (uint32_t k = 0; k < 64u / (uint32_t)kFarLanes; ++k) {   // 64 (level, side) slots over the group's lanes
        const uint32_t slot = sub + k * (uint32_t)kFarLanes, lev = slot >> 1;
        if (have && lev < rt.levels) {
            const uint32_t a = (l + (1u << lev) - 1u) >> lev, e = (r + 1u) >> lev;
            const bool take = a < e && ((slot & 1u) ? (e & 1u) : (a & 1u));
            if (take) { const Box c = load_box(boxes, rt.offset[lev] + ((slot & 1u) ? e - 1u : a)); box_merge(b, c); }
        }
    }
#pragma unroll
    for (int off = kFarLanes / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = fminf(b.lo[k], __shfl_xor(b.lo[k], off));
            b.hi[k] = fmaxf(b.hi[k], __shfl_xor(b.hi[k], off));
        }
    }
    return b;
}

// the same for one lane's node, both children at once: [l, gamma] and [gamma + 1, r].  The four loads of a level go out
// before the previous level's four are merged (a load instruction per entry and level, masked to the lanes that take
// the entry: the number of loads in flight stays known to the compiler, which waits for all but the newest)
__device__ __forceinline__ void range_box_pair(const RangeTree &rt, const float4 *__restrict__ boxes, uint32_t l, uint32_t gamma, uint32_t r,
                                               Box &bl, Box &br)
{
    bl = box_empty();
    br = box_empty();
    float4 p[8];
    bool had[4] = {false, false, false, false};
    for (uint32_t lev = 0;; ++lev) {
        const uint32_t round_up = (1u << lev) - 1u;
        const uint32_t a0 = (l + round_up) >> lev, e0 = (gamma + 1u) >> lev, a1 = (gamma + 1u + round_up) >> lev, e1 = (r + 1u) >> lev;
        const bool act0 = lev < rt.levels && a0 < e0, act1 = lev < rt.levels && a1 < e1;
        const bool has[4] = {act0 && (a0 & 1u), act0 && (e0 & 1u), act1 && (a1 & 1u), act1 && (e1 & 1u)};
        const uint32_t base = act0 || act1 ? rt.offset[lev] : 0u;
        const uint32_t at[4] = {base + a0, base + e0 - 1u, base + a1, base + e1 - 1u};
        float4 q[8];
        // only the lanes that take an entry load it (a load's cost follows its active lanes): k_refit_nodes 55.4 -> 48.0 us,
        // k_hierarchy 58.8 -> 57.3 us once it is held at six waves per SIMD (61.2 without: 88 registers, five waves)
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (has[k]) { q[2 * k] = boxes[2 * (size_t)at[k]]; q[2 * k + 1] = boxes[2 * (size_t)at[k] + 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (had[k]) {
                Box &b = k < 2 ? bl : br;
                b.lo[0] = fminf(b.lo[0], p[2 * k].x); b.lo[1] = fminf(b.lo[1], p[2 * k].y); b.lo[2] = fminf(b.lo[2], p[2 * k].z);
                b.hi[0] = fmaxf(b.hi[0], p[2 * k + 1].x); b.hi[1] = fmaxf(b.hi[1], p[2 * k + 1].y); b.hi[2] = fmaxf(b.hi[2], p[2 * k + 1].z);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = q[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) had[k] = has[k];
        if (!act0 && !act1) break;   // (this level loaded nothing that counts; the previous one's loads have been merged)
    }
}

// ------------------------------------------------------------------------------------------
// Radix-tree hierarchy (Karras 2012) over the leaf keys, one thread per internal node i:
// the thread finds its node's leaf range [l,r] and split from the keys alone and takes the boxes of
// its two children [l,split] and [split+1,r] from the aligned-range tree -- no bottom-up pass, no
// atomics, no cross-workgroup hand-off.  The left child precedes the right one in Morton order.
// ------------------------------------------------------------------------------------------
// (six waves per SIMD asked for: with the range queries' loads masked the allocator would otherwise take 88 registers and five)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_hierarchy(const uint32_t *__restrict__ keys, uint32_t L_, uint32_t g,
                                                      RangeTree rt, const float4 *__restrict__ boxes,
                                                      FatNode *__restrict__ nodes)
{
    const int L = (int)L_;
    __shared__ uint32_t s_keys[kBlock + 2 * kKeyWindow];
    // (a wave's 64 nodes on their way out: quarter k of lane's node at [k * 68 + lane] -- lanes side by side when written, and
    // the 16 consecutive quarters a store pass reads (4 lanes' nodes) land in 16 different 16-byte columns: 68 = 4 mod 16.  As
    // [4 * lane + k] every write was a four-way bank conflict; the counters had LDS conflicts at 45 % of this kernel's LDS cycles)
    __shared__ float4 s_out[kBlock / 64][4 * 68];
    const KeyWindow kw = {s_keys, (int)(blockIdx.x * kBlock) - kKeyWindow, (int)(kBlock + 2 * kKeyWindow)};
    for (int a = (int)threadIdx.x; a < kw.n; a += (int)kBlock) {
        const int j = kw.first + a;
        s_keys[a] = (j >= 0 && j < L) ? keys[(size_t)j * g] : 0u;
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    const uint32_t lane = threadIdx.x & 63u;
    const bool node = i < L - 1;
    const uint32_t ki = s_keys[threadIdx.x + kKeyWindow];
    bool far = false;
    int d = 1, dmin = -1, len = 0, s = 0;
    if (node) {
        d = delta_win(kw, L, i, ki, i + 1, far) > delta_win(kw, L, i, ki, i - 1, far) ? 1 : -1;   // (both inside the window)
        dmin = delta_win(kw, L, i, ki, i - d, far);
        int lmax = 2;
        while (delta_win(kw, L, i, ki, i + lmax * d, far) > dmin) lmax <<= 1;
        for (int t = lmax >> 1; t >= 1; t >>= 1)
            if (delta_win(kw, L, i, ki, i + (len + t) * d, far) > dmin) len += t;
        if (!far) {
            const int dnode = delta_win(kw, L, i, ki, i + len * d, far);
            int t = len;
            do {
                t = (t + 1) >> 1;
                if (delta_win(kw, L, i, ki, i + (s + t) * d, far) > dnode) s += t;
            } while (t > 1);
        }
    }
    Box bl = box_empty(), br = box_empty();
    // the wave's far nodes, 64 / kFarLanes at a time, a group of lanes on each
    for (unsigned long long todo = __ballot(node && far); todo;) {
        constexpr int kGroups = 64 / kFarLanes;
        const int q = (int)(lane / (uint32_t)kFarLanes);
        int f = -1;   // the lane whose node this group takes: the q-th of the far lanes left
        unsigned long long batch = 0ull, m = todo;
#pragma unroll
        for (int k = 0; k < kGroups; ++k) {
            if (m) {
                const int b = __builtin_ctzll(m);
                if (k == q) f = b;
                batch |= 1ull << b;
                m &= m - 1ull;
            }
        }
        todo = m;
        const bool have = f >= 0;
        const int src = have ? f : 0;
        const int fi = __shfl(i, src), fd = __shfl(d, src), fdmin = __shfl(dmin, src);
        const uint32_t fki = (uint32_t)__shfl((int)ki, src);
        const int reach = fd > 0 ? L - 1 - fi : fi;   // the last t with i + t d inside [0, L)
        const int flen = coop_search(keys, g, L, fi, fki, fd, fdmin, have ? 1 : 0, have ? reach + 1 : 0, lane);
        const int fj = fi + flen * fd;
        const int fdnode = have ? delta_of(fi, fki, fj, keys[(size_t)fj * g]) : 0;
        const int fs = coop_search(keys, g, L, fi, fki, fd, fdnode, 0, have ? flen : 0, lane);
        const int fgamma = fi + fs * fd + min(fd, 0), fl = min(fi, fj), fr = max(fi, fj);
        const Box cl = coop_range_box(rt, boxes, (uint32_t)fl, (uint32_t)fgamma, lane, have);
        const Box cr = coop_range_box(rt, boxes, (uint32_t)fgamma + 1u, (uint32_t)fr, lane, have);
        // the owner takes the result from the first lane of the group that worked on its node (its rank among the batch)
        const bool mine_now = (batch >> lane) & 1ull;
        const int from = (int)__popcll(batch & ((1ull << lane) - 1ull)) * kFarLanes;
        const int olen = __shfl(flen, from), os = __shfl(fs, from);
        Box obl, obr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            obl.lo[k] = __shfl(cl.lo[k], from); obl.hi[k] = __shfl(cl.hi[k], from);
            obr.lo[k] = __shfl(cr.lo[k], from); obr.hi[k] = __shfl(cr.hi[k], from);
        }
        if (mine_now) { len = olen; s = os; bl = obl; br = obr; }
    }
    const int j = i + len * d;
    const int gamma = i + s * d + min(d, 0);
    const int l = min(i, j), r = max(i, j);
    const uint32_t left = (l == gamma) ? (kLeafBit | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t right = (r == gamma + 1) ? (kLeafBit | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    if (node && !far) range_box_pair(rt, boxes, (uint32_t)l, (uint32_t)gamma, (uint32_t)r, bl, br);
    // the wave's 64 nodes are 4 KB in a row: through LDS, so that every store instruction writes 1 KB contiguous (a lane
    // storing its own node's four quarters writes 16 bytes of 64 different lines each time)
    float4 *mine = s_out[threadIdx.x >> 6];
    if (node) {
        mine[0 * 68 + lane] = make_float4(bl.lo[0], bl.lo[1], bl.lo[2], __uint_as_float(left));
        mine[1 * 68 + lane] = make_float4(bl.hi[0], bl.hi[1], bl.hi[2], __uint_as_float(right));
        // (the fourth words of the right child's box: the node's leaf range, for k_refit_nodes)
        mine[2 * 68 + lane] = make_float4(br.lo[0], br.lo[1], br.lo[2], __uint_as_float((uint32_t)l));
        mine[3 * 68 + lane] = make_float4(br.hi[0], br.hi[1], br.hi[2], __uint_as_float((uint32_t)r));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int wave_first = i - (int)lane;   // the wave's first node
    float4 *out = reinterpret_cast<float4 *>(nodes + wave_first);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = k * 64 + (int)lane;   // quarter q of the wave's 256
        if (wave_first + (q >> 2) < L - 1) out[q] = mine[(q & 3) * 68 + (q >> 2)];
    }
}

// ------------------------------------------------------------------------------------------
// Refit: the hierarchy's topology stands (same sorted keys), only the boxes follow the vertices.  k_hierarchy left every
// node's leaf range next to its child references, so a node needs no search: its two range queries and 48 bytes of stores.
// Nodes with long ranges go to groups of lanes as k_hierarchy's far nodes do.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_refit_nodes(uint32_t L_, RangeTree rt, const float4 *__restrict__ boxes, FatNode *__restrict__ nodes)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63u;
    const bool node = i + 1u < L_;
    uint32_t left = 0, right = 0, l = 0, r = 0;
    if (node) {
        const float4 *q = reinterpret_cast<const float4 *>(nodes + i);
        left = __float_as_uint(q[0].w); right = __float_as_uint(q[1].w);
        l = __float_as_uint(q[2].w); r = __float_as_uint(q[3].w);
    }
    const uint32_t gamma = left & ~kLeafBit;
    const bool wide = node && r - l >= (uint32_t)(kBlock + 2 * kKeyWindow);
    Box bl = box_empty(), br = box_empty();
    for (unsigned long long todo = __ballot(wide); todo;) {
        constexpr int kGroups = 64 / kFarLanes;
        const int q = (int)(lane / (uint32_t)kFarLanes);
        int f = -1;
        unsigned long long batch = 0ull, m = todo;
#pragma unroll
        for (int k = 0; k < kGroups; ++k) {
            if (m) {
                const int b = __builtin_ctzll(m);
                if (k == q) f = b;
                batch |= 1ull << b;
                m &= m - 1ull;
            }
        }
        todo = m;
        const bool have = f >= 0;
        const int src = have ? f : 0;
        const uint32_t fl = (uint32_t)__shfl((int)l, src), fg = (uint32_t)__shfl((int)gamma, src), fr = (uint32_t)__shfl((int)r, src);
        const Box cl = coop_range_box(rt, boxes, fl, fg, lane, have);
        const Box cr = coop_range_box(rt, boxes, fg + 1u, fr, lane, have);
        const int from = (int)__popcll(batch & ((1ull << lane) - 1ull)) * kFarLanes;
        Box obl, obr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            obl.lo[k] = __shfl(cl.lo[k], from); obl.hi[k] = __shfl(cl.hi[k], from);
            obr.lo[k] = __shfl(cr.lo[k], from); obr.hi[k] = __shfl(cr.hi[k], from);
        }
        if ((batch >> lane) & 1ull) { bl = obl; br = obr; }
    }
    if (node && !wide) range_box_pair(rt, boxes, l, gamma, r, bl, br);
    if (node) {
        float4 *out = reinterpret_cast<float4 *>(nodes + i);
        out[0] = make_float4(bl.lo[0], bl.lo[1], bl.lo[2], __uint_as_float(left));
        out[1] = make_float4(bl.hi[0], bl.hi[1], bl.hi[2], __uint_as_float(right));
        out[2] = make_float4(br.lo[0], br.lo[1], br.lo[2], __uint_as_float(l));
        out[3] = make_float4(br.hi[0], br.hi[1], br.hi[2], __uint_as_float(r));
    }
}

// ------------------------------------------------------------------------------------------
// Fused ray generation + closest-hit traversal, persistent waves with ray refill.
//
// The grid is a fixed number of resident blocks per CU (trace_grid_blocks: 2 of the 5 that the 32 KB of LDS
// stack per block would admit).  The shard's rays sit in kQueues queues, one per XCD: queue x =
// azimuth sector x, enumerated channel by channel, so that the 64 rays a wave takes together are 64
// consecutive azimuth columns of one channel (coherent: their node fetches share cache lines) and
// an XCD's L2 keeps seeing its own sector of the BVH.  A wave reads its XCD id from HW_REG_XCC_ID
// (speed only: an exhausted queue is followed by the others).  Whenever at least kRefillMin lanes
// of a wave are idle -- wave-ballot -- the idle lanes take the next rays of the queue with ONE
// atomicAdd: no lane waits for the slowest ray of its wave, and the kernel has no tail.
//
// Per lane: BVH2 traversal with both child boxes in the parent (one 64-byte fetch resolves two
// boxes), left child first = front to back for every ray (sensor-centred Morton order), pending
// right children on a per-lane stack in LDS (spill to global past 32 entries).
// ------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float4 ld16(const float4 *p)
{
    if (MODE == 1) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    if (MODE == 2) {
        float4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return *p;
}

template <bool COUNT, int MODE>
__global__ __launch_bounds__(kBlock) void k_trace(SensorTables tb, RayQueues rq, const FatNode *__restrict__ nodes,
                                                  const TriRecord *__restrict__ records, uint32_t L, uint32_t g,
                                                  uint32_t ntris, float *__restrict__ t_out,
                                                  uint32_t *__restrict__ gid_out, uint32_t *__restrict__ spill,
                                                  unsigned long long *__restrict__ visit_counts)
{
    __shared__ uint32_t s_stack[kStackLds][kBlock];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t *my_spill = spill + ((size_t)blockIdx.x * kBlock + tid) * kStackSpill;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t qsel = xcc & (kQueues - 1);
    uint32_t tried = 0;
    bool drained = false;
    const uint32_t root = (L > 1u) ? 0u : kLeafBit;

    bool has = false;
    V3 d = {0.f, 0.f, 1.f};
    float ix = 1.f, iy = 1.f, iz = 1.f, best = INFINITY;
    uint32_t bid = kInvalid, q = 0, cur = kInvalid, sp = 0;
    uint32_t cn = 0, ctri = 0, trips = 0;
    const float4 *rec4 = reinterpret_cast<const float4 *>(records);

    while (true) {
        unsigned long long act = __ballot(has);
        if (!drained && (uint32_t)__popcll(act) <= 64u - rq.refill_min) {
            const unsigned long long idle = ~act;
            const uint32_t nidle = (uint32_t)__popcll(idle);
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            while (tried < (uint32_t)kQueues) {
                const uint32_t first = (uint32_t)(((unsigned long long)qsel * tb.naz) / kQueues);
                const uint32_t width = (uint32_t)(((unsigned long long)(qsel + 1u) * tb.naz) / kQueues) - first;
                const uint32_t len = width * tb.V;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&rq.heads[qsel * 16u], nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base < len) {
                    const uint32_t sidx = base + rank;
                    if (!has && sidx < len) {
                        const uint32_t j = sidx / width, c = sidx - j * width;
                        // (j and chan_mul are below V: the product fits 32 bits for every real sensor; the 64-bit remainder is ~100 instructions)
                        const uint32_t jj = tb.V < 65536u ? (j * rq.chan_mul) % tb.V : (uint32_t)(((unsigned long long)j * rq.chan_mul) % tb.V);
                        const uint32_t v = rq.chan_order ? rq.chan_order[tb.V - 1u - jj] : jj;   // from the highest ring down
                        const uint32_t hl = first + c, h = tb.az0 + hl;
                        // LidarDevice.cpp:310-316: d = (sin(theta)cos(phi), sin(theta)sin(phi), cos(theta))
                        const float st = tb.sin_theta[v];
                        d = {st * tb.cos_phi[h], st * tb.sin_phi[h], tb.cos_theta[v]};
                        ix = safe_inv(d.x); iy = safe_inv(d.y); iz = safe_inv(d.z);
                        q = v * tb.naz + hl;
                        best = INFINITY; bid = kInvalid; sp = 0; cur = root; has = true;
                    }
                    break;
                }
                ++tried;
                qsel = (qsel + 1u) & (kQueues - 1);
            }
            if (tried >= (uint32_t)kQueues) drained = true;
            act = __ballot(has);
        }
        if (act == 0ull) break;
        if (COUNT) ++trips;
        // the leaf tests run when enough lanes stand at a leaf (or nobody has a node to go to): a lane at a leaf waits
        bool leaf_phase = true;
        if (rq.leaf_wait) {
            const unsigned long long at_leaf = __ballot(has && (cur & kLeafBit));
            const unsigned long long at_node = __ballot(has && !(cur & kLeafBit));
            leaf_phase = (uint32_t)__popcll(at_leaf) >= rq.leaf_wait || at_node == 0ull;
        }
        if (has) {
            if (!(cur & kLeafBit)) {
                const float4 *nd = nodes[cur].q;
                const float4 A = ld16<MODE>(nd), B = ld16<MODE>(nd + 1), C = ld16<MODE>(nd + 2), D = ld16<MODE>(nd + 3);
                if (COUNT) ++cn;
                float x1 = A.x * ix, x2 = B.x * ix, y1 = A.y * iy, y2 = B.y * iy, z1 = A.z * iz, z2 = B.z * iz;
                const float tnL = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
                const float tfL = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), best));
                x1 = C.x * ix; x2 = D.x * ix; y1 = C.y * iy; y2 = D.y * iy; z1 = C.z * iz; z2 = D.z * iz;
                const float tnR = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
                const float tfR = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), best));
                const bool hl_ = tnL <= tfL, hr_ = tnR <= tfR;
                const uint32_t left = __float_as_uint(A.w), right = __float_as_uint(B.w);
                if (hl_) {
                    cur = left;
                    if (hr_) {  // push the far (right) child
                        if (sp < (uint32_t)kStackLds) s_stack[sp][tid] = right;
                        else if (sp < (uint32_t)(kStackLds + kStackSpill)) my_spill[sp - kStackLds] = right;
                        ++sp;
                    }
                } else if (hr_) {
                    cur = right;
                } else {
                    cur = kInvalid;
                    if (sp) { --sp; cur = sp < (uint32_t)kStackLds ? s_stack[sp][tid] : my_spill[sp - kStackLds]; }
                }
            }
            while (leaf_phase && cur != kInvalid && (cur & kLeafBit)) {
                const uint32_t first = (cur & ~kLeafBit) * g;
                const uint32_t last = min(first + g, ntris);
                for (uint32_t s = first; s < last; ++s) {
                    const float4 r0 = ld16<MODE>(rec4 + 3 * (size_t)s), r1 = ld16<MODE>(rec4 + 3 * (size_t)s + 1), r2 = ld16<MODE>(rec4 + 3 * (size_t)s + 2);
                    if (COUNT) ++ctri;
                    float t;
                    if (tri_test(d, {r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}, r1.w, t)) {
                        const uint32_t id = __float_as_uint(r0.w);
                        if (t < best || (t == best && id < bid)) { best = t; bid = id; }
                    }
                }
                cur = kInvalid;
                if (sp) { --sp; cur = sp < (uint32_t)kStackLds ? s_stack[sp][tid] : my_spill[sp - kStackLds]; }
            }
            if (cur == kInvalid) {
                t_out[q] = (bid == kInvalid) ? -1.0f : best;
                gid_out[q] = bid;
                has = false;
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { cn += __shfl_xor(cn, off); ctri += __shfl_xor(ctri, off); }
        if (lane == 0) {
            atomicAdd(&visit_counts[0], (unsigned long long)cn);
            atomicAdd(&visit_counts[1], (unsigned long long)ctri);
            atomicAdd(&visit_counts[2], (unsigned long long)trips);  // traversal-loop trips of this wave
            atomicMax(&visit_counts[3], (unsigned long long)trips);
        }
    }
}

// Four-wide nodes of a binary hierarchy (round 6): wide node i stands for binary node i with its CHILDREN skipped -- its slots are
// node i's grandchildren, a child that is a leaf keeps a slot of its own -- two to four slots, the rest empty (reference kInvalid).
// One lane per binary node, no order among them: every node gets a wide twin at its own index (a walk from the root uses every
// other level's; which ones is not known here, and a second, compacting pass would cost more than the unused half's stores).
// The two children of a node are neighbours in the node array (Karras: split and split + 1), and most nodes are close to their
// children: the gathers mostly land in lines a neighbouring lane reads as its own node.
__global__ __launch_bounds__(kBlock) void k_widen(const FatNode *__restrict__ nodes, uint32_t n_nodes, WideNode *__restrict__ wide)
{
    // a wave's 64 wide nodes are 8 KB in a row: they leave through LDS, so that every store instruction writes 1 KB
    // contiguous (a lane storing its own node's eight quarters would write 16 bytes of 64 different lines each time).
    // Quarter k of lane's node sits at [k * 66 + lane]: lanes side by side when written; read back as node-major
    // quarters q = 8 * node + k, eight consecutive q of one node are 66 apart = 2 mod 16 sixteen-byte columns: two nodes'
    // worth (16 quarters) touch every column pair once.
    __shared__ float4 s_out[kBlock / 64][8 * 66];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x, lane = threadIdx.x & 63u;
    float4 lo[4], hi[4];
    const float4 none_lo = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(kInvalid)), none_hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c) { lo[c] = none_lo; hi[c] = none_hi; }
    if (i < n_nodes) {
        const float4 *nd = nodes[i].q;
        const float4 A = nd[0], B = nd[1], C = nd[2], D = nd[3];
        const uint32_t left = __float_as_uint(A.w), right = __float_as_uint(B.w);
        // slots 0, 1: the left child's side; 2, 3: the right child's.  Both children's nodes are fetched before either is used.
        const bool lin = !(left & kLeafBit), rin = !(right & kLeafBit);
        float4 la = none_lo, lb = none_hi, lc = none_lo, ld = none_hi, ra = none_lo, rb = none_hi, rc = none_lo, rd = none_hi;
        if (lin) { const float4 *c = nodes[left].q; la = c[0]; lb = c[1]; lc = c[2]; ld = c[3]; }
        if (rin) { const float4 *c = nodes[right].q; ra = c[0]; rb = c[1]; rc = c[2]; rd = c[3]; }
        if (lin) {
            lo[0] = make_float4(la.x, la.y, la.z, la.w); hi[0] = make_float4(lb.x, lb.y, lb.z, 0.0f);
            lo[1] = make_float4(lc.x, lc.y, lc.z, lb.w); hi[1] = make_float4(ld.x, ld.y, ld.z, 0.0f);
        } else {
            lo[0] = make_float4(A.x, A.y, A.z, __uint_as_float(left));
            hi[0] = make_float4(B.x, B.y, B.z, 0.0f);
        }
        if (rin) {
            lo[2] = make_float4(ra.x, ra.y, ra.z, ra.w); hi[2] = make_float4(rb.x, rb.y, rb.z, 0.0f);
            lo[3] = make_float4(rc.x, rc.y, rc.z, rb.w); hi[3] = make_float4(rd.x, rd.y, rd.z, 0.0f);
        } else {
            lo[2] = make_float4(C.x, C.y, C.z, __uint_as_float(right));
            hi[2] = make_float4(D.x, D.y, D.z, 0.0f);
        }
    }
    float4 *mine = s_out[threadIdx.x >> 6];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mine[c * 66 + lane] = lo[c]; mine[(4 + c) * 66 + lane] = hi[c]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t wave_first = i - lane;
    float4 *out = reinterpret_cast<float4 *>(wide + wave_first);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t q = (uint32_t)k * 64u + lane;   // quarter q of the wave's 512: node q / 8, its quarter q % 8
        if (wave_first + (q >> 3) < n_nodes) out[q] = mine[(q & 7u) * 66u + (q >> 3)];
    }
}

// Top of a hierarchy in breadth-first order, for the LDS of k_trace_inst: slot 0 = the root; a node's child that is an
// internal node gets the next free slot while there are any (its reference becomes kTreeletBit | slot), leaves and
// the nodes below the last slot keep their references.  One workgroup; a level at a time.
__global__ __launch_bounds__(kBlock) void k_treelet(const FatNode *__restrict__ nodes, FatNode *__restrict__ out)
{
    __shared__ uint32_t s_q[kTreeletNodes];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) { s_q[0] = 0u; s_n = 1u; }
    __syncthreads();
    uint32_t done = 0;
    while (true) {
        const uint32_t end = min(s_n, kTreeletNodes);
        __syncthreads();
        if (done >= end) break;
        for (uint32_t slot = done + threadIdx.x; slot < end; slot += kBlock) {
            const float4 *nd = nodes[s_q[slot]].q;
            float4 A = nd[0], B = nd[1];
            const float4 C = nd[2], D = nd[3];
            uint32_t ref[2] = {__float_as_uint(A.w), __float_as_uint(B.w)};
            for (int c = 0; c < 2; ++c) {
                if (ref[c] & kLeafBit) continue;
                const uint32_t idx = atomicAdd(&s_n, 1u);
                if (idx < kTreeletNodes) { s_q[idx] = ref[c]; ref[c] = kTreeletBit | idx; }
            }
            A.w = __uint_as_float(ref[0]);
            B.w = __uint_as_float(ref[1]);
            float4 *o = out[slot].q;
            o[0] = A; o[1] = B; o[2] = C; o[3] = D;
        }
        __syncthreads();
        done = end;
    }
    // (a hierarchy of fewer nodes than slots: the rest reads as zero -- no memset in front of this kernel)
    for (uint32_t slot = min(s_n, kTreeletNodes) + threadIdx.x; slot < kTreeletNodes; slot += kBlock) {
        float4 *o = out[slot].q;
        o[0] = o[1] = o[2] = o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// Instanced mode: the same persistent waves and ray refill as k_trace, over one hierarchy PER GEOMETRY built once in
// mesh space (InstBatch, ls_kernels.h).  A lane walks the geometries one after the other: ray origin o and direction
// minv * d in that geometry's mesh space (the parameter t stays the sensor-space one, so the closest hit so far prunes
// across geometries), slab tests against boxes widened by eps -- (lo - eps - o) * inv and (hi + eps - o) * inv as one
// fused multiply-add each --, the nearer child first (a mesh-space hierarchy has no front-to-back order by
// construction), and at a leaf the exact test of every other path: the three corners through this frame's transform
// (xform_vertex: the bits of k_transform / k_project), e1, e2, NgC, Embree's test against the table direction.
// ------------------------------------------------------------------------------------------
//
// WIDE (round 6, the default): the nodes are the FOUR-wide ones k_widen makes of the same hierarchy -- a node's slots are its
// grandchildren (a child that is a leaf keeps its slot), 128 bytes: one fetch resolves four boxes and a trip goes down two
// levels of the binary tree.  The kernel is bound by what a trip costs a wave in latency (EXPERIMENTS.md E4): half the trips,
// each with four independent slab tests in flight instead of two.  The nearest hit child is walked on, the other hits are
// pushed as they stand (three stores whatever they hold; the stack pointer counts the hits).  The same
// boxes, leaves and exact test as the binary walk: a grandchild is visited when ITS box is hit (the binary walk also asks
// its parent's, which contains it: the wide walk visits a superset), so the closest hit and its tie-break are the same bits.
template <bool COUNT, bool SINGLE /* one geometry (the usual scene): its descriptor is wave-uniform, scalar registers */, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_trace_inst(SensorTables tb, RayQueues rq, InstBatch batch, const FatNode *__restrict__ nodes,
                                                       const WideNode *__restrict__ wide,
                                                       const TriRecord *__restrict__ records, uint32_t g,
                                                       const FatNode *__restrict__ treelet, float *__restrict__ t_out,
                                                       uint32_t *__restrict__ gid_out, uint32_t *__restrict__ spill,
                                                       unsigned long long *__restrict__ visit_counts)
{
    __shared__ uint32_t s_stack[kStackLds][kBlock];
    // one geometry: the top of its hierarchy (launch_treelet: breadth-first, kTreeletNodes nodes) sits in LDS; the first
    // levels of every ray -- the fetches every lane of every wave makes -- are LDS reads instead of cache round trips
    // (binary nodes only: it was worth 0.6 % there, E4, and the wide walk spends the LDS on nothing)
    __shared__ float4 s_tree[(SINGLE && !WIDE) ? 4 * kTreeletNodes : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const bool use_tree = SINGLE && !WIDE && treelet != nullptr && batch.g[0].n_leaves > 1u;
    if (SINGLE && !WIDE && use_tree) {
        const float4 *src = reinterpret_cast<const float4 *>(treelet);
        for (uint32_t i = tid; i < 4u * kTreeletNodes; i += kBlock) s_tree[i] = src[i];
        __syncthreads();
    }
    uint32_t *my_spill = spill + ((size_t)blockIdx.x * kBlock + tid) * kStackSpill;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t qsel = xcc & (kQueues - 1);
    uint32_t tried = 0;
    bool drained = false;

    bool has = false;
    V3 d = {0.f, 0.f, 1.f};
    float ix = 1.f, iy = 1.f, iz = 1.f, best = INFINITY;
    float cxl = 0.f, cxh = 0.f, cyl = 0.f, cyh = 0.f, czl = 0.f, czh = 0.f;   // -(o +- eps) * inv per axis
    uint32_t bid = kInvalid, q = 0, cur = kInvalid, sp = 0, gi = 0;
    uint32_t cn = 0, ctri = 0, trips = 0;
    const float4 *rec4 = reinterpret_cast<const float4 *>(records);

    // the ray of this lane in geometry `gi`'s mesh space; cur = its root (kInvalid: nothing there)
    auto enter = [&](uint32_t k) {
        const InstGeom &ig = batch.g[k];
        cur = kInvalid;
        if (!ig.n_leaves) return;
        const float dx = (ig.minv[0] * d.x + ig.minv[1] * d.y) + ig.minv[2] * d.z;
        const float dy = (ig.minv[3] * d.x + ig.minv[4] * d.y) + ig.minv[5] * d.z;
        const float dz = (ig.minv[6] * d.x + ig.minv[7] * d.y) + ig.minv[8] * d.z;
        ix = safe_inv(dx); iy = safe_inv(dy); iz = safe_inv(dz);
        cxl = -(ig.o[0] + ig.eps) * ix; cxh = -(ig.o[0] - ig.eps) * ix;
        cyl = -(ig.o[1] + ig.eps) * iy; cyh = -(ig.o[1] - ig.eps) * iy;
        czl = -(ig.o[2] + ig.eps) * iz; czh = -(ig.o[2] - ig.eps) * iz;
        cur = ig.n_leaves > 1u ? (use_tree ? kTreeletBit : 0u) : kLeafBit;
    };
    // next thing to do for a lane whose current subtree is finished: the stack, else the next geometry, else done
    auto advance = [&]() {
        cur = kInvalid;
        if (sp) { --sp; cur = sp < (uint32_t)kStackLds ? s_stack[sp][tid] : my_spill[sp - kStackLds]; return; }
        if (!SINGLE)
            while (cur == kInvalid && ++gi < batch.n) enter(gi);
    };

    while (true) {
        unsigned long long act = __ballot(has);
        if (!drained && (uint32_t)__popcll(act) <= 64u - rq.refill_min) {
            const unsigned long long idle = ~act;
            const uint32_t nidle = (uint32_t)__popcll(idle);
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            while (tried < (uint32_t)kQueues) {
                const uint32_t first = (uint32_t)(((unsigned long long)qsel * tb.naz) / kQueues);
                const uint32_t width = (uint32_t)(((unsigned long long)(qsel + 1u) * tb.naz) / kQueues) - first;
                const uint32_t len = width * tb.V;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&rq.heads[qsel * 16u], nidle);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base < len) {
                    const uint32_t sidx = base + rank;
                    if (!has && sidx < len) {
                        const uint32_t j = sidx / width, c = sidx - j * width;
                        // (j and chan_mul are below V: the product fits 32 bits for every real sensor; the 64-bit remainder is ~100 instructions)
                        const uint32_t jj = tb.V < 65536u ? (j * rq.chan_mul) % tb.V : (uint32_t)(((unsigned long long)j * rq.chan_mul) % tb.V);
                        const uint32_t v = rq.chan_order ? rq.chan_order[tb.V - 1u - jj] : jj;   // from the highest ring down
                        const uint32_t hl = first + c, h = tb.az0 + hl;
                        // LidarDevice.cpp:310-316: d = (sin(theta)cos(phi), sin(theta)sin(phi), cos(theta))
                        const float st = tb.sin_theta[v];
                        d = {st * tb.cos_phi[h], st * tb.sin_phi[h], tb.cos_theta[v]};
                        q = v * tb.naz + hl;
                        best = INFINITY; bid = kInvalid; sp = 0; has = true;
                        gi = 0;
                        enter(0);
                        if (!SINGLE)
                            while (cur == kInvalid && ++gi < batch.n) enter(gi);
                    }
                    break;
                }
                ++tried;
                qsel = (qsel + 1u) & (kQueues - 1);
            }
            if (tried >= (uint32_t)kQueues) drained = true;
            act = __ballot(has);
        }
        if (act == 0ull) break;
        if (COUNT) ++trips;
        bool leaf_phase = true;
        // one traversal step of this lane in geometry `ig`: a node (both child boxes), then the leaves it leads to
        auto step = [&](const InstGeom &ig) {
            // `ig` is geometry `gi` as this trip started; advance() may move the lane on to another geometry, whose
            // leaves must not be tested with `ig`'s records and transform: they wait for the next trip
            const uint32_t at_entry = gi;
            if (WIDE && cur != kInvalid && !(cur & kLeafBit)) {
                const float4 *nd = wide[ig.node_first + cur].q;
                const float4 l0 = nd[0], l1 = nd[1], l2 = nd[2], l3 = nd[3], h0 = nd[4], h1 = nd[5], h2 = nd[6], h3 = nd[7];
                if (COUNT) ++cn;
                auto slab = [&](const float4 &lo, const float4 &hi, float &tn) {
                    const float x1 = fmaf(lo.x, ix, cxl), x2 = fmaf(hi.x, ix, cxh), y1 = fmaf(lo.y, iy, cyl), y2 = fmaf(hi.y, iy, cyh),
                                z1 = fmaf(lo.z, iz, czl), z2 = fmaf(hi.z, iz, czh);
                    tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
                    const float tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), best));
                    // (the far bound gets two ulps: the products above round once each; an empty slot's reference is kInvalid)
                    return tn <= tf * 1.0000003f && __float_as_uint(lo.w) != kInvalid;
                };
                float k0, k1, k2, k3;
                const bool b0 = slab(l0, h0, k0), b1 = slab(l1, h1, k1), b2 = slab(l2, h2, k2), b3 = slab(l3, h3, k3);
                uint32_t r0 = __float_as_uint(l0.w), r1 = __float_as_uint(l1.w), r2 = __float_as_uint(l2.w), r3 = __float_as_uint(l3.w);
                k0 = b0 ? k0 : INFINITY; k1 = b1 ? k1 : INFINITY; k2 = b2 ? k2 : INFINITY; k3 = b3 ? k3 : INFINITY;
                r0 = b0 ? r0 : kInvalid; r1 = b1 ? r1 : kInvalid; r2 = b2 ? r2 : kInvalid; r3 = b3 ? r3 : kInvalid;
                // The nearest hit child comes to the front (three comparators; a missed slot never wins: its key is +inf, and among
                // equal keys a hit goes before a miss) and is walked on; the other hits are pushed as they stand.  Sorting all four
                // (a five-comparator network) was measured: the same cloud, 0.6 % slower at SYN-1M, 4 % at configs[4]'s size -- a trip
                // is a chain of dependent instructions at two waves per SIMD, and the order of what waits on the stack matters less
                // than the twenty instructions it costs (EXPERIMENTS.md E8.3).
                auto cswap = [](float &ka, uint32_t &ra, float &kb, uint32_t &rb) {
                    const bool sw = kb < ka || (ra == kInvalid && rb != kInvalid);
                    const float kt = sw ? kb : ka; kb = sw ? ka : kb; ka = kt;
                    const uint32_t rt = sw ? rb : ra; rb = sw ? ra : rb; ra = rt;
                };
                cswap(k0, r0, k1, r1); cswap(k0, r0, k2, r2); cswap(k0, r0, k3, r3);
                if (r0 == kInvalid) {
                    advance();
                } else {
                    const uint32_t v3 = r3 != kInvalid ? 1u : 0u, v2 = r2 != kInvalid ? 1u : 0u, v1 = r1 != kInvalid ? 1u : 0u;
                    if (sp + 3u <= (uint32_t)kStackLds) {   // the usual case: three stores, whatever they hold; the pointer counts the hits
                        s_stack[sp][tid] = r3;
                        s_stack[sp + v3][tid] = r2;
                        s_stack[sp + v3 + v2][tid] = r1;
                        sp += v3 + v2 + v1;
                    } else {
                        auto push = [&](uint32_t ref) {
                            if (sp < (uint32_t)kStackLds) s_stack[sp][tid] = ref;
                            else if (sp < (uint32_t)(kStackLds + kStackSpill)) my_spill[sp - kStackLds] = ref;
                            ++sp;
                        };
                        if (v3) push(r3);
                        if (v2) push(r2);
                        if (v1) push(r1);
                    }
                    cur = r0;
                }
            }
            if (!WIDE && cur != kInvalid && !(cur & kLeafBit)) {
                float4 A, B, C, D;
                if (SINGLE && (cur & kTreeletBit)) {
                    const float4 *nd = s_tree + 4u * (cur & (kTreeletBit - 1u));
                    A = nd[0]; B = nd[1]; C = nd[2]; D = nd[3];
                } else {
                    const float4 *nd = nodes[ig.node_first + cur].q;
                    A = nd[0]; B = nd[1]; C = nd[2]; D = nd[3];
                }
                if (COUNT) ++cn;
                float x1 = fmaf(A.x, ix, cxl), x2 = fmaf(B.x, ix, cxh), y1 = fmaf(A.y, iy, cyl), y2 = fmaf(B.y, iy, cyh),
                      z1 = fmaf(A.z, iz, czl), z2 = fmaf(B.z, iz, czh);
                const float tnL = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
                const float tfL = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), best));
                x1 = fmaf(C.x, ix, cxl); x2 = fmaf(D.x, ix, cxh); y1 = fmaf(C.y, iy, cyl); y2 = fmaf(D.y, iy, cyh);
                z1 = fmaf(C.z, iz, czl); z2 = fmaf(D.z, iz, czh);
                const float tnR = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.0f));
                const float tfR = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), best));
                // (the far bound gets two ulps: the products above round once each)
                const bool hl_ = tnL <= tfL * 1.0000003f, hr_ = tnR <= tfR * 1.0000003f;
                const uint32_t left = __float_as_uint(A.w), right = __float_as_uint(B.w);
                if (hl_ && hr_) {
                    const bool left_first = tnL <= tnR;
                    const uint32_t far_child = left_first ? right : left;
                    cur = left_first ? left : right;
                    if (sp < (uint32_t)kStackLds) s_stack[sp][tid] = far_child;
                    else if (sp < (uint32_t)(kStackLds + kStackSpill)) my_spill[sp - kStackLds] = far_child;
                    ++sp;
                } else if (hl_) {
                    cur = left;
                } else if (hr_) {
                    cur = right;
                } else {
                    advance();
                }
            }
            // (a leaf reached above belongs to `ig` unless advance() moved on to another geometry: then the loop does not
            // run and the next trip continues there)
            while (leaf_phase && cur != kInvalid && (cur & kLeafBit) && (SINGLE || gi == at_entry)) {
                const uint32_t first = (cur & ~kLeafBit) * g;
                const uint32_t last = min(first + g, ig.n_tris);
                for (uint32_t s = first; s < last; ++s) {
                    const size_t at = 3 * ((size_t)ig.rec_first + s);
                    const float4 r0 = rec4[at], r1 = rec4[at + 1], r2 = rec4[at + 2];
                    const uint32_t local = __float_as_uint(r0.w);
                    if (COUNT) ++ctri;
                    V3 v0, v1, v2;
                    if (ig.xform == 2) {
                        v0 = xform_vertex_sensor_only(ig.m, reinterpret_cast<const uint8_t *>(&r0));
                        v1 = xform_vertex_sensor_only(ig.m, reinterpret_cast<const uint8_t *>(&r1));
                        v2 = xform_vertex_sensor_only(ig.m, reinterpret_cast<const uint8_t *>(&r2));
                    } else {
                        v0 = xform_vertex(ig.m, reinterpret_cast<const uint8_t *>(&r0));
                        v1 = xform_vertex(ig.m, reinterpret_cast<const uint8_t *>(&r1));
                        v2 = xform_vertex(ig.m, reinterpret_cast<const uint8_t *>(&r2));
                    }
                    const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
                    const float NgC = dot_fma(cross_fma(e2, e1), v0);
                    float t;
                    if (tri_test(d, v0, e1, e2, NgC, t)) {
                        const uint32_t id = ig.gid_first + local;
                        if (t < best || (t == best && id < bid)) { best = t; bid = id; }
                    }
                }
                advance();
            }
        };
        // the leaf tests run when enough lanes stand at a leaf (or nobody has a node to go to): a lane at a leaf waits
        if (rq.leaf_wait) {
            const unsigned long long at_leaf = __ballot(has && cur != kInvalid && (cur & kLeafBit));
            const unsigned long long at_node = __ballot(has && cur != kInvalid && !(cur & kLeafBit));
            leaf_phase = (uint32_t)__popcll(at_leaf) >= rq.leaf_wait || at_node == 0ull;
        }
        if (has) {
            if (SINGLE) step(batch.g[0]);
            else step(batch.g[gi]);
            if (cur == kInvalid) {
                t_out[q] = (bid == kInvalid) ? -1.0f : best;
                gid_out[q] = bid;
                has = false;
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { cn += __shfl_xor(cn, off); ctri += __shfl_xor(ctri, off); }
        if (lane == 0) {
            atomicAdd(&visit_counts[0], (unsigned long long)cn);
            atomicAdd(&visit_counts[1], (unsigned long long)ctri);
            atomicAdd(&visit_counts[2], (unsigned long long)trips);
            atomicMax(&visit_counts[3], (unsigned long long)trips);
        }
    }
}

// hits per block of 256 consecutive rays (feeds the ordered pack)
// (and the ray queues' heads go back to zero for the next frame's k_trace -- the traversal that used them is done: no
// memset launch in front of a frame)
__global__ __launch_bounds__(kBlock) void k_rowcount(const uint32_t *__restrict__ gid, uint32_t n,
                                                     uint32_t *__restrict__ block_counts, uint32_t *__restrict__ queue_heads)
{
    __shared__ uint32_t s_cnt[kBlock / 64];
    static_assert(kQueues * 16 <= kBlock, "one thread per word of the queue heads");
    if (blockIdx.x == 0 && queue_heads && threadIdx.x < (uint32_t)kQueues * 16u) queue_heads[threadIdx.x] = 0u;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    const bool hit = q < n && gid[q] != kInvalid;
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ------------------------------------------------------------------------------------------
// Ordered pack: one thread per ray q (shard-local index v*naz + column); block b owns rays
// [256b, 256b+256).  Its output offset is the sum of the hit counts of the blocks before it (each
// block reduces its own prefix: no scan kernel, no cross-block hand-off).  Emits the 32-byte
// PointCloud2 record and the 16-byte ls_hit, both in ray-index order.
// Bytes per hit: 8 (t,gid) + 48 written.
// ------------------------------------------------------------------------------------------
// FROM_KEYS (projection engine): the per-ray results are the 64-bit closest-hit keys; the kernel also
// writes the dense t / gid arrays, re-arms the keys, zeroes the OTHER frame-parity's block counters
// and the big-triangle queue length, so that a frame needs no memset and no resolve pass.
template <bool FROM_KEYS>
__global__ __launch_bounds__(kBlock) void k_pack(SensorTables tb, float *__restrict__ t_io, uint32_t *__restrict__ gid_io,
                                                 unsigned long long *__restrict__ keys,
                                                 const uint32_t *__restrict__ block_counts,
                                                 uint32_t *__restrict__ next_block_counts, uint32_t *__restrict__ big_count,
                                                 GeomTable gt, float4 *__restrict__ points, uint4 *__restrict__ hits,
                                                 uint32_t *__restrict__ n_points, uint32_t compact, uint32_t block0, uint32_t n_blocks,
                                                 ProgressArgs pg)
{
    // (block0, n_blocks: the launch covers ray blocks [block0, block0 + gridDim.x) of n_blocks -- a synchronous frame that
    // reports its progress to the host packs its two halves in two launches, ls_trace_scene_expand)
    const uint32_t block = blockIdx.x + block0;
    __shared__ uint32_t s_part[kBlock / 64];
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_hint[kBlock / 64];
    const uint32_t nq = tb.V * tb.naz;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;

    // every load of the kernel is issued up front -- the ray's key (or t / gid), its table entries, the first geometry
    // slot, then the counts of the blocks before this one -- so that the kernel is one memory round trip deep, not four
    // (it is a few microseconds long: round trips are what it consists of)
    const uint32_t q = block * kBlock + threadIdx.x;
    uint32_t gid = kInvalid;
    float t = -1.0f;
    unsigned long long key = ~0ull;
    if (q < nq) {
        if (FROM_KEYS) {
            key = keys[q];
            // no dense per-ray arrays on this path: ls_debug_dense_hits rebuilds them from the records
        } else {
            gid = gid_io[q];
            t = t_io[q];
        }
    }
    const uint32_t qq = q < nq ? q : 0u;
    const uint32_t v = qq / tb.naz, h = tb.az0 + (qq - v * tb.naz);
    const float st = tb.sin_theta[v], ctv = tb.cos_theta[v];
    const float2 cs = tb.cs_phi[h];
    uint32_t first0 = 0, geom0 = 0, shift0 = 0;   // slot 0: the only one of a one-mesh scene
    if (gt.n) { first0 = gt.tri_first[0]; geom0 = gt.geom_ids[0]; shift0 = gt.prim_shift[0]; }

    // (eight of a thread's counts requested at a time -- all of them for a raster of up to 2 048 blocks: as a plain loop the last
    // blocks of the grid, which start last and add up the most, walked four dependent round trips of two loads; 15.35 -> 15.05 us
    // per frame with three in flight, 24.9 -> 24.6 with one; starting those blocks first as well: no further gain)
    uint32_t acc = 0;
    constexpr uint32_t kCountsAhead = 8;
    for (uint32_t r0 = threadIdx.x; r0 < block; r0 += kCountsAhead * kBlock) {
        uint32_t c[kCountsAhead];
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) c[k] = r0 + k * kBlock < block ? block_counts[r0 + k * kBlock] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) acc += c[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);

    if (FROM_KEYS) {
        if (key != ~0ull) {
            keys[q] = ~0ull;   // re-arm (a key nobody touched is armed already)
            gid = (uint32_t)key;
            t = __uint_as_float((uint32_t)(key >> 32));
        }
        if (threadIdx.x == 0) next_block_counts[block] = 0u;
        if (q == 0) big_count[0] = 0u;                                             // queue length
        // group-cull survivor counts (one per list segment): all of them, whatever the shard's size -- a shard of 256 rays
        // or fewer is ONE workgroup, which used to re-arm the first 256 of the 512 counters only
        if (block == 0u) {
            // (read before they go: the fullest segment tells the host how large the next frames' culled launch has to be)
            uint32_t fullest = 0;
            for (uint32_t i = threadIdx.x; i < kCullCounters; i += kBlock) {
                fullest = max(fullest, big_count[kCullCountAt + i * 16u]);
                big_count[kCullCountAt + i * 16u] = 0u;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, off));
            if (lane == 0) s_hint[w] = fullest;
        }
    }
    const bool hit = gid != kInvalid;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) { s_part[w] = acc; s_wave[w] = (uint32_t)__popcll(m); }
    __syncthreads();
    uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    for (uint32_t k = 0; k < w; ++k) base += s_wave[k];
    if (block == n_blocks - 1 && threadIdx.x == 0) *n_points = base + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (FROM_KEYS && block == 0u && threadIdx.x == 0 && pg.cull_hint)
        __hip_atomic_store(pg.cull_hint, 1u + max(max(s_hint[0], s_hint[1]), max(s_hint[2], s_hint[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (hit) {
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t dst = base + rank;

        // EmbreeTracer.cpp:341-345: xyz = tfar*dir, intensity 64.0; ring = channel (LidarDeviceKernels.cu:51)
        if (compact == 3u) {
            // LS_OPT_EMIT_POINTS = 0: hit records only (a sharded group rebuilds the points from the gathered records)
        } else if (compact == 2u) {
            // ls_trace_scene_begin: 8 bytes cross PCIe, (ray, t); ls_trace_scene_expand rebuilds the point from the host's
            // copy of the factor tables with the operations below
            reinterpret_cast<uint2 *>(points)[dst] = make_uint2(v * tb.H + h, __float_as_uint(t));
        } else if (compact) {
            // host-visible compact form (LS_OPT_HOST_OUTPUT = 2): 16 bytes cross PCIe, ls_expand_points rebuilds the record
            points[dst] = make_float4(t * (st * cs.x), t * (st * cs.y), t * ctv, __int_as_float((int)v));
        } else {
            points[2 * (size_t)dst] = make_float4(t * (st * cs.x), t * (st * cs.y), t * ctv, 0.0f);
            points[2 * (size_t)dst + 1] = make_float4(64.0f, __int_as_float((int)v), 0.0f, 0.0f);
        }
        // (geomID, primID) from the global triangle id: last geometry slot whose first id <= gid
        uint32_t lo = 0, hi = gt.n;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (gt.tri_first[mid] <= gid) lo = mid; else hi = mid;
        }
        const uint32_t first = lo ? gt.tri_first[lo] : first0, geom = lo ? gt.geom_ids[lo] : geom0, shift = lo ? gt.prim_shift[lo] : shift0;
        if (hits) hits[dst] = make_uint4(v * tb.H + h, geom, (gid - first) >> shift, __float_as_uint(t));
    }
    if (!FROM_KEYS || !pg.host || block != 0u) return;
    // ---- ls_trace_scene_begin: the frame's very first pack workgroup tells the host how many points there will be, how
    //      many of them the launch before `split` packs, and where the points halve (the next frame's split).  Thread t
    //      owns a contiguous chunk of the block counts: its sum, an exclusive scan of the 256 sums, then the one thread
    //      whose chunk holds the half-way point walks it.  (No "last workgroup of the finish pass" detection: 2 048
    //      atomics on one counter are 22 us.)
    __shared__ uint32_t s_scan[kBlock / 64], s_first[kBlock / 64], s_even;
    const uint32_t chunk = (n_blocks + kBlock - 1u) / kBlock;
    const uint32_t c0 = min(threadIdx.x * chunk, n_blocks), c1 = min(c0 + chunk, n_blocks);
    uint32_t mine = 0, mine_first = 0;
    for (uint32_t i0 = c0; i0 < c1; i0 += kCountsAhead) {   // (the chunk's loads side by side: this workgroup's word is what the host waits for)
        uint32_t c[kCountsAhead];
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) c[k] = i0 + k < c1 ? block_counts[i0 + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) {
            mine += c[k];
            if (i0 + k < pg.split) mine_first += c[k];
        }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine_first += __shfl_xor(mine_first, off);
    if (threadIdx.x == 0) s_even = n_blocks;
    __syncthreads();
    if (lane == 63u) s_scan[w] = incl;
    if (lane == 0) s_first[w] = mine_first;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t k = 0; k < w; ++k) before += s_scan[k];
    const uint32_t total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    if (mine && 2u * before < total && 2u * (before + mine) >= total) {   // exactly one thread (total > 0)
        uint32_t run = before, at = c0;
        while (at < c1 && 2u * run < total) run += block_counts[at++];
        s_even = at;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pg.host->total, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pg.host->n_first, s_first[0] + s_first[1] + s_first[2] + s_first[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pg.host->even_split, s_even, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pg.host->total_epoch, pg.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_pack<true> with RPT rays per lane (thread t takes rays t, t + 256, ... of its workgroup's RPT x 256): fewer, fatter waves for
// the frames that overlap on three streams, where the device is short of wave slots (k_project_finish_wide, ls_project.hip, has
// the measurements).  Everything k_pack<true> does except the progress reports of a two-step trace, which is one frame in flight.
template <uint32_t RPT>
__global__ __launch_bounds__(kBlock) void k_pack_wide(SensorTables tb, unsigned long long *__restrict__ keys,
                                                      const uint32_t *__restrict__ block_counts, uint32_t *__restrict__ next_block_counts,
                                                      uint32_t *__restrict__ big_count, GeomTable gt, float4 *__restrict__ points,
                                                      uint4 *__restrict__ hits, uint32_t *__restrict__ n_points, uint32_t compact,
                                                      uint32_t n_blocks, uint32_t *cull_hint)
{
    __shared__ uint32_t s_part[kBlock / 64];
    __shared__ uint32_t s_cnt[RPT][kBlock / 64];
    __shared__ uint32_t s_hint[kBlock / 64];
    const uint32_t nq = tb.V * tb.naz;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t wg = blockIdx.x, block0 = wg * RPT;           // the first of this workgroup's 256-ray blocks
    const uint32_t q0 = block0 * kBlock + threadIdx.x;
    unsigned long long key[RPT];
    uint32_t v[RPT], h[RPT];
    float st[RPT], ctv[RPT];
    float2 cs[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t q = q0 + j * kBlock;
        key[j] = q < nq ? keys[q] : ~0ull;
    }
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t qq = min(q0 + j * kBlock, nq - 1u);
        v[j] = qq / tb.naz;
        h[j] = tb.az0 + (qq - v[j] * tb.naz);
        st[j] = tb.sin_theta[v[j]];
        ctv[j] = tb.cos_theta[v[j]];
        cs[j] = tb.cs_phi[h[j]];
    }
    uint32_t first0 = 0, geom0 = 0, shift0 = 0;
    if (gt.n) { first0 = gt.tri_first[0]; geom0 = gt.geom_ids[0]; shift0 = gt.prim_shift[0]; }
    uint32_t acc = 0;
    constexpr uint32_t kCountsAhead = 8;
    for (uint32_t r0 = threadIdx.x; r0 < block0; r0 += kCountsAhead * kBlock) {
        uint32_t c[kCountsAhead];
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) c[k] = r0 + k * kBlock < block0 ? block_counts[r0 + k * kBlock] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < kCountsAhead; ++k) acc += c[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    unsigned long long m[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        if (key[j] != ~0ull) keys[q0 + j * kBlock] = ~0ull;   // re-arm
        m[j] = __ballot(key[j] != ~0ull);
        if (lane == 0) s_cnt[j][w] = (uint32_t)__popcll(m[j]);
    }
    if (threadIdx.x < RPT && block0 + threadIdx.x < n_blocks) next_block_counts[block0 + threadIdx.x] = 0u;
    if (q0 == 0) big_count[0] = 0u;
    if (wg == 0u) {
        uint32_t fullest = 0;
        for (uint32_t i = threadIdx.x; i < kCullCounters; i += kBlock) {
            fullest = max(fullest, big_count[kCullCountAt + i * 16u]);
            big_count[kCullCountAt + i * 16u] = 0u;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, off));
        if (lane == 0) s_hint[w] = fullest;
    }
    if (lane == 0) s_part[w] = acc;
    __syncthreads();
    uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (wg == gridDim.x - 1u && threadIdx.x == 0) {
        uint32_t total = base;
        for (uint32_t j = 0; j < RPT; ++j) total += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        *n_points = total;
    }
    if (wg == 0u && threadIdx.x == 0 && cull_hint)
        __hip_atomic_store(cull_hint, 1u + max(max(s_hint[0], s_hint[1]), max(s_hint[2], s_hint[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        uint32_t before = base;
        for (uint32_t k = 0; k < w; ++k) before += s_cnt[j][k];
        base += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        if (key[j] == ~0ull) continue;
        const uint32_t dst = before + (uint32_t)__popcll(m[j] & ((1ull << lane) - 1ull));
        const uint32_t gid = (uint32_t)key[j];
        const float t = __uint_as_float((uint32_t)(key[j] >> 32));
        if (compact == 3u) {
        } else if (compact) {
            points[dst] = make_float4(t * (st[j] * cs[j].x), t * (st[j] * cs[j].y), t * ctv[j], __int_as_float((int)v[j]));
        } else {
            points[2 * (size_t)dst] = make_float4(t * (st[j] * cs[j].x), t * (st[j] * cs[j].y), t * ctv[j], 0.0f);
            points[2 * (size_t)dst + 1] = make_float4(64.0f, __int_as_float((int)v[j]), 0.0f, 0.0f);
        }
        uint32_t lo = 0, hi = gt.n;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (gt.tri_first[mid] <= gid) lo = mid; else hi = mid;
        }
        const uint32_t first = lo ? gt.tri_first[lo] : first0, geom = lo ? gt.geom_ids[lo] : geom0, shift = lo ? gt.prim_shift[lo] : shift0;
        if (hits) hits[dst] = make_uint4(v[j] * tb.H + h[j], geom, (gid - first) >> shift, __float_as_uint(t));
    }
}

// one word for the host, in stream order (ls_trace_scene_expand polls it)
__global__ void k_signal(uint32_t *word, uint32_t value) { __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// Debug view of the projection engine's result: dense per-ray (t, global triangle id) arrays from
// the packed hit records (ls_debug_dense_hits); k_dense_clear first, then one thread per record.
__global__ __launch_bounds__(kBlock) void k_dense_clear(uint32_t nq, float *__restrict__ t, uint32_t *__restrict__ gid)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q < nq) { t[q] = -1.0f; gid[q] = kInvalid; }
}

__global__ __launch_bounds__(kBlock) void k_dense_from_hits(SensorTables tb, const uint4 *__restrict__ hits,
                                                            const uint32_t *__restrict__ n_points, GeomTable gt,
                                                            float *__restrict__ t, uint32_t *__restrict__ gid)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *n_points) return;
    const uint4 rec = hits[i];                        // (ray, geomID, primID, bits(t))
    const uint32_t v = rec.x / tb.H, h = rec.x - v * tb.H;
    uint32_t lo = 0, hi = gt.n;                       // geometry slot of geomID (slots ascend by id)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (gt.geom_ids[mid] <= rec.y) lo = mid; else hi = mid;
    }
    const uint32_t q = v * tb.naz + (h - tb.az0);
    t[q] = __uint_as_float(rec.w);
    gid[q] = gt.tri_first[lo] + (rec.z << gt.prim_shift[lo]);   // a quad's first triangle stands for the quad
}

// Sensor frame -> world frame for the x,y,z of 32-byte points: p_world = R * (A * p) + t
// (CloudTransformer.cpp:283-318 + LidarDevice.cpp:393-401, same operation order as the oracle).
// One thread per point; the count and the output offset are read on the device.
__global__ __launch_bounds__(kBlock) void k_cloud_to_world(Affine m /* a = A, rinv = R, t */, const float4 *__restrict__ in,
                                                           const uint32_t *__restrict__ n_points, float4 *__restrict__ out,
                                                           const uint32_t *__restrict__ out_base, uint32_t *__restrict__ out_total,
                                                           uint32_t capacity, uint32_t max_points)
{
    const uint32_t n = min(*n_points, max_points);
    const uint32_t base = out_base ? *out_base : 0u;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0 && out_total) *out_total = min(base + n, max(capacity, base));
    if (i >= n || base + i >= capacity) return;
    const float4 p = in[2 * (size_t)i], rest = in[2 * (size_t)i + 1];
    float q[3], w[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = ((m.a[4 * r + 0] * p.x + m.a[4 * r + 1] * p.y) + m.a[4 * r + 2] * p.z) + m.a[4 * r + 3];
#pragma unroll
    for (int r = 0; r < 3; ++r) w[r] = ((m.rinv[3 * r + 0] * q[0] + m.rinv[3 * r + 1] * q[1]) + m.rinv[3 * r + 2] * q[2]) + m.t[r];
    out[2 * (size_t)(base + i)] = make_float4(w[0], w[1], w[2], p.w);
    out[2 * (size_t)(base + i) + 1] = rest;
}

// Multi-GPU: turn gathered per-rank hit slots into one contiguous cloud.  Slot r (slot_words uint32 each)
// = [n_r | 15 words pad | n_r..cap ls_hit records]; output = all records in rank order (ascending
// azimuth sector) + their 32-byte points rebuilt from (ray, t): xyz = t * dir(ray) (EmbreeTracer.cpp:341-345).
__global__ __launch_bounds__(kBlock) void k_expand_slots(SensorTables tb, const uint32_t *__restrict__ gathered,
                                                         uint32_t world, uint32_t cap, uint32_t slot_words,
                                                         float4 *__restrict__ points, uint4 *__restrict__ hits,
                                                         uint32_t *__restrict__ n_points, GatherStat *__restrict__ stat, uint32_t epoch)
{
    // cap: records that TRAVELLED per slot (the sized gather moves the front of a slot: its header holds the rank's true
    // count, its records beyond cap stayed behind -- such a frame is reported as truncated, never delivered as complete)
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= world * cap) return;
    const uint32_t r = i / cap, j = i - r * cap;
    // (the record is requested before the headers are looked at -- record j < cap of slot r lies inside what travelled whether
    // the slot holds that many hits or not --, so that the kernel is two round trips deep, not three)
    const uint4 rec = reinterpret_cast<const uint4 *>(gathered + (size_t)r * slot_words + 16)[j];
    uint32_t off = 0, total = 0, most = 0;
    for (uint32_t k = 0; k < world; ++k) {
        const uint32_t c_true = gathered[(size_t)k * slot_words], c = min(c_true, cap);
        if (k < r) off += c;
        total += c;
        most = max(most, c_true);
    }
    if (i == 0) {
        *n_points = total;
        if (stat) {   // pinned host memory: what the host sizes the next gathers from, and whether this one was big enough
            stat->max_count = most;
            stat->truncated = most > cap ? 1u : 0u;
            __hip_atomic_store(&stat->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (j >= min(gathered[(size_t)r * slot_words], cap)) return;
    const uint32_t v = rec.x / tb.H, h = rec.x - v * tb.H;
    const float t = __uint_as_float(rec.w);
    const float st = tb.sin_theta[v];
    const float2 cs = tb.cs_phi[h];
    const uint32_t dst = off + j;
    hits[dst] = rec;
    points[2 * (size_t)dst] = make_float4(t * (st * cs.x), t * (st * cs.y), t * tb.cos_theta[v], 0.0f);
    points[2 * (size_t)dst + 1] = make_float4(64.0f, __int_as_float((int)v), 0.0f, 0.0f);
}

// LidarDevice::allRaysGPU (LidarDeviceKernels.cu:25-52): directions as SoA, shard-compact order.
__global__ __launch_bounds__(kBlock) void k_raygen(SensorTables tb, float *__restrict__ dx, float *__restrict__ dy,
                                                   float *__restrict__ dz)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= tb.V * tb.naz) return;
    const uint32_t v = q / tb.naz, h = tb.az0 + (q - v * tb.naz);
    const float st = tb.sin_theta[v];
    dx[q] = st * tb.cos_phi[h];
    dy[q] = st * tb.sin_phi[h];
    dz[q] = tb.cos_theta[v];
}

// allRaysGPUKernel's outputs in its own layout (LidarDeviceKernels.cu:25-52): lidarshooter::Ray = origin xyz, tmin,
// direction xyz, tmax (Ray.hpp:16-35); lidarshooter::Hit = t, normal xyz, intensity, ring (Hit.hpp:16-29), initialised
// as there: t = 1e16, intensity = 64.0, ring = channel.  Two 16-byte stores per ray, three 8-byte stores per hit record.
__global__ __launch_bounds__(kBlock) void k_raygen_aos(SensorTables tb, float4 *__restrict__ rays, float2 *__restrict__ hits)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= tb.V * tb.naz) return;
    const uint32_t v = q / tb.naz, h = tb.az0 + (q - v * tb.naz);
    const float st = tb.sin_theta[v];
    if (rays) {
        rays[2 * (size_t)q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                      // origin, tmin
        rays[2 * (size_t)q + 1] = make_float4(st * tb.cos_phi[h], st * tb.sin_phi[h], tb.cos_theta[v], 1e16f);   // direction, tmax
    }
    if (hits) {
        hits[3 * (size_t)q] = make_float2(1e16f, 0.0f);                        // t, normal.x
        hits[3 * (size_t)q + 1] = make_float2(0.0f, 0.0f);                     // normal.y, normal.z
        hits[3 * (size_t)q + 2] = make_float2(64.0f, __int_as_float((int)v));  // intensity, ring
    }
}

// ------------------------------------------------------------------------------------------
// Exhaustive checker: every ray against every triangle straight from the scene arrays (does not
// touch the BVH or the triangle records).  Triangles are staged through LDS 256 at a time.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_bruteforce(SensorTables tb, const float *__restrict__ verts,
                                                       const uint32_t *__restrict__ tris, uint32_t ntris,
                                                       float *__restrict__ t_out, uint32_t *__restrict__ gid_out)
{
    __shared__ float s[kBlock][10];
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t nq = tb.V * tb.naz;
    const bool active = q < nq;
    const uint32_t qq = active ? q : 0u;
    const uint32_t v = qq / tb.naz, h = tb.az0 + (qq - v * tb.naz);
    const float st = tb.sin_theta[v];
    const V3 d = {st * tb.cos_phi[h], st * tb.sin_phi[h], tb.cos_theta[v]};
    float best = INFINITY;
    uint32_t bid = kInvalid;
    for (uint32_t base = 0; base < ntris; base += kBlock) {
        const uint32_t k = base + threadIdx.x;
        __syncthreads();
        if (k < ntris) {
            const uint32_t i0 = tris[3 * (size_t)k], i1 = tris[3 * (size_t)k + 1], i2 = tris[3 * (size_t)k + 2];
            const V3 v0 = {verts[3 * (size_t)i0], verts[3 * (size_t)i0 + 1], verts[3 * (size_t)i0 + 2]};
            const V3 v1 = {verts[3 * (size_t)i1], verts[3 * (size_t)i1 + 1], verts[3 * (size_t)i1 + 2]};
            const V3 v2 = {verts[3 * (size_t)i2], verts[3 * (size_t)i2 + 1], verts[3 * (size_t)i2 + 2]};
            const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
            const float NgC = dot_fma(cross_fma(e2, e1), v0);
            float *o = s[threadIdx.x];
            o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = e1.x; o[4] = e1.y; o[5] = e1.z;
            o[6] = e2.x; o[7] = e2.y; o[8] = e2.z; o[9] = NgC;
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)kBlock, ntris - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            const float *o = s[j];
            float t;
            if (tri_test(d, {o[0], o[1], o[2]}, {o[3], o[4], o[5]}, {o[6], o[7], o[8]}, o[9], t)) {
                if (t < best) { best = t; bid = base + j; }  // ascending ids: ties keep the lowest
            }
        }
    }
    if (active) {
        t_out[q] = (bid == kInvalid) ? -1.0f : best;
        gid_out[q] = bid;
    }
}

inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

// ---- launch wrappers -----------------------------------------------------------------------
void launch_transform(hipStream_t s, const void *raw, uint32_t stride, uint32_t n, const float *affine12,
                      const float *rinv9, const float *t3, float *out_xyz, uint32_t *d_maxabs_bits)
{
    if (!n) return;
    Affine m;
    for (int i = 0; i < 12; ++i) m.a[i] = affine12[i];
    for (int i = 0; i < 9; ++i) m.rinv[i] = rinv9[i];
    for (int i = 0; i < 3; ++i) m.t[i] = t3[i];
    const uint32_t blocks = (n + (uint32_t)kTransformBlock - 1u) / (uint32_t)kTransformBlock;
    hipLaunchKernelGGL(k_transform, dim3(min(blocks, 128u)), dim3(kTransformBlock), 0, s, static_cast<const uint8_t *>(raw),
                       stride, n, m, out_xyz, d_maxabs_bits);
}

void launch_rebase(hipStream_t s, const uint32_t *idx, uint32_t n_idx, uint32_t vbase, uint32_t *out)
{
    if (!n_idx) return;
    hipLaunchKernelGGL(k_rebase, dim3(blocks_for(n_idx)), dim3(kBlock), 0, s, idx, n_idx, vbase, out);
}

// vals: null = not written (the sort is told the values are the positions); first_counts: null, or sort_first_counts()
// of the scratch the keys are going to be sorted with (then launch_sort(..., first_counted = true))
void launch_morton(hipStream_t s, const float *verts, const uint32_t *tris, uint32_t ntris,
                   const uint32_t *d_maxabs_bits, uint32_t *keys, uint32_t *vals, uint32_t *first_counts)
{
    if (!ntris) return;
    const uint32_t blocks = (ntris + kSortTile - 1u) / kSortTile;
    if (first_counts)
        hipLaunchKernelGGL(k_morton<true>, dim3(blocks), dim3(kMortonBlock), 0, s, verts, tris, ntris, d_maxabs_bits, keys, vals, first_counts);
    else
        hipLaunchKernelGGL(k_morton<false>, dim3(blocks), dim3(kMortonBlock), 0, s, verts, tris, ntris, d_maxabs_bits, keys, vals, first_counts);
}

// records, leaf boxes and the aligned-range tree: k_leaves_tree, then k_range_top for the levels above a workgroup's 512 leaves
void launch_leaves_tree(hipStream_t s, const float *verts, const uint32_t *tris, const uint32_t *sorted_vals, uint32_t ntris,
                        uint32_t leaf_size, TriRecord *records, const RangeTree &rt, float4 *boxes, bool mesh_records)
{
    if (!ntris) return;
    const uint32_t blocks = (rt.count[0] + 511u) / 512u;
    if (mesh_records)
        hipLaunchKernelGGL(k_leaves_tree<true>, dim3(blocks), dim3(kBlock), 0, s, verts, tris, sorted_vals, ntris, leaf_size, records, rt, boxes);
    else
        hipLaunchKernelGGL(k_leaves_tree<false>, dim3(blocks), dim3(kBlock), 0, s, verts, tris, sorted_vals, ntris, leaf_size, records, rt, boxes);
    if (rt.levels > (uint32_t)kBottomLevels + 1u) hipLaunchKernelGGL(k_range_top, dim3(1), dim3(kBlock), 0, s, rt, boxes);
}

void launch_hierarchy(hipStream_t s, const uint32_t *sorted_keys, uint32_t nleaves, uint32_t leaf_size,
                      const RangeTree &rt, const float4 *boxes, FatNode *nodes)
{
    if (nleaves < 2) return;
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks_for(nleaves - 1)), dim3(kBlock), 0, s, sorted_keys, nleaves,
                       leaf_size, rt, boxes, nodes);
}

// refit: the nodes hold the topology of the same sorted keys (launch_hierarchy ran on them): boxes only
void launch_refit_nodes(hipStream_t s, uint32_t nleaves, const RangeTree &rt, const float4 *boxes, FatNode *nodes)
{
    if (nleaves < 2) return;
    hipLaunchKernelGGL(k_refit_nodes, dim3(blocks_for(nleaves - 1)), dim3(kBlock), 0, s, nleaves, rt, boxes, nodes);
}

uint32_t trace_grid_blocks(int device)
{
    // Persistent grid: 2 blocks (8 waves) per CU.  The LDS stacks would admit 5, but the traversal is bound by what a trip
    // costs a wave in issue and dependency latency, not by residency, and fewer waves leave more
    // L1 per wave: with whole-wave refills and sequential channels, k_trace_inst takes 0.142 ms with 2, 0.148 with 3,
    // 0.180 with 4, 0.20 with 1 (headline frame).  LS_TRACE_BLOCKS_PER_CU overrides.
    hipDeviceProp_t prop;
    uint32_t cus = 256u;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        cus = (uint32_t)prop.multiProcessorCount;
    uint32_t per_cu = 2u;
    per_cu = (uint32_t)max(1, lsi::tune_int("LS_TRACE_BLOCKS_PER_CU", (int)per_cu));
    return cus * per_cu;
}

size_t trace_spill_bytes(uint32_t grid_blocks) { return (size_t)grid_blocks * kBlock * kStackSpill * sizeof(uint32_t); }

void launch_trace(hipStream_t s, uint32_t grid_blocks, const SensorTables &tb, const RayQueues &rq,
                  const FatNode *nodes, const TriRecord *records, uint32_t nleaves, uint32_t leaf_size,
                  uint32_t ntris, float *t_out, uint32_t *gid_out, uint32_t *spill, unsigned long long *visit_counts)
{
    const uint32_t nq = tb.V * tb.naz;
    if (!nq || !nleaves) return;
    const uint32_t grid = min(grid_blocks, (nq + kBlock - 1) / kBlock);
    static const int mode = lsi::tune_int("LS_TRACE_LOAD_MODE", 0);
    if (visit_counts)
        hipLaunchKernelGGL((k_trace<true, 0>), dim3(grid), dim3(kBlock), 0, s, tb, rq, nodes, records, nleaves, leaf_size,
                           ntris, t_out, gid_out, spill, visit_counts);
    else if (mode == 1)
        hipLaunchKernelGGL((k_trace<false, 1>), dim3(grid), dim3(kBlock), 0, s, tb, rq, nodes, records, nleaves,
                           leaf_size, ntris, t_out, gid_out, spill, visit_counts);
    else
        hipLaunchKernelGGL((k_trace<false, 0>), dim3(grid), dim3(kBlock), 0, s, tb, rq, nodes, records, nleaves,
                           leaf_size, ntris, t_out, gid_out, spill, visit_counts);
}

void launch_treelet(hipStream_t s, const FatNode *nodes, uint32_t n_leaves, FatNode *treelet)
{
    if (n_leaves < 2u) return;
    hipLaunchKernelGGL(k_treelet, dim3(1), dim3(kBlock), 0, s, nodes, treelet);
}

void launch_trace_instanced(hipStream_t s, uint32_t grid_blocks, const SensorTables &tb, const RayQueues &rq, const InstBatch &batch,
                            const FatNode *nodes, const WideNode *wide, const TriRecord *records, uint32_t leaf_size, const FatNode *treelet, float *t_out,
                            uint32_t *gid_out, uint32_t *spill, unsigned long long *visit_counts)
{
    const uint32_t nq = tb.V * tb.naz;
    if (!nq || !batch.n) return;
    const uint32_t grid = min(grid_blocks, (nq + kBlock - 1) / kBlock);
#define LS_TRACE_INST(C, S, W) hipLaunchKernelGGL((k_trace_inst<C, S, W>), dim3(grid), dim3(kBlock), 0, s, tb, rq, batch, nodes, wide, records, leaf_size, \
                                                   treelet, t_out, gid_out, spill, visit_counts)
    if (wide) {
        if (visit_counts) { if (batch.n == 1u) LS_TRACE_INST(true, true, true); else LS_TRACE_INST(true, false, true); }
        else { if (batch.n == 1u) LS_TRACE_INST(false, true, true); else LS_TRACE_INST(false, false, true); }
    } else {
        if (visit_counts) { if (batch.n == 1u) LS_TRACE_INST(true, true, false); else LS_TRACE_INST(true, false, false); }
        else { if (batch.n == 1u) LS_TRACE_INST(false, true, false); else LS_TRACE_INST(false, false, false); }
    }
#undef LS_TRACE_INST
}

void launch_widen(hipStream_t s, const FatNode *nodes, uint32_t n_leaves, WideNode *wide)
{
    if (n_leaves < 2u) return;
    hipLaunchKernelGGL(k_widen, dim3(blocks_for(n_leaves - 1u)), dim3(kBlock), 0, s, nodes, n_leaves - 1u, wide);
}

void launch_quads_to_triangles(hipStream_t s, const uint32_t *quad_idx, uint32_t n_quads, uint32_t *tri_idx)
{
    if (!n_quads) return;
    hipLaunchKernelGGL(k_quads_to_triangles, dim3(blocks_for(n_quads)), dim3(kBlock), 0, s, quad_idx, n_quads, tri_idx);
}

void launch_index_max(hipStream_t s, const uint32_t *idx, uint32_t n, uint32_t *d_max)
{
    (void)hipMemsetAsync(d_max, 0, 4, s);
    if (!n) return;
    const uint32_t blocks = std::min<uint32_t>(blocks_for((n + 3u) / 4u), 256u);
    hipLaunchKernelGGL(k_index_max, dim3(blocks), dim3(kBlock), 0, s, idx, n, d_max);
}

void launch_rowcount(hipStream_t s, const uint32_t *gid, uint32_t nrays, uint32_t *row_counts, uint32_t *queue_heads)
{
    if (!nrays) return;
    hipLaunchKernelGGL(k_rowcount, dim3(blocks_for(nrays)), dim3(kBlock), 0, s, gid, nrays, row_counts, queue_heads);
}

void launch_pack(hipStream_t s, const SensorTables &tb, float *t, uint32_t *gid, const uint32_t *block_counts,
                 const GeomTable &gt, uint8_t *points32, void *hits, uint32_t *n_points, uint32_t compact)
{
    const uint32_t nq = tb.V * tb.naz;
    if (!nq) return;
    hipLaunchKernelGGL(k_pack<false>, dim3(blocks_for(nq)), dim3(kBlock), 0, s, tb, t, gid,
                       static_cast<unsigned long long *>(nullptr), block_counts, static_cast<uint32_t *>(nullptr),
                       static_cast<uint32_t *>(nullptr), gt, reinterpret_cast<float4 *>(points32),
                       reinterpret_cast<uint4 *>(hits), n_points, compact, 0u, blocks_for(nq), ProgressArgs{nullptr, 0u, 0u, nullptr});
}

void launch_pack_keys(hipStream_t s, const SensorTables &tb, unsigned long long *keys, float *t, uint32_t *gid,
                      const uint32_t *block_counts, uint32_t *next_block_counts, uint32_t *big_count,
                      const GeomTable &gt, uint8_t *points32, void *hits, uint32_t *n_points, uint32_t compact,
                      const ProgressArgs *progress, uint32_t rays_per_lane)
{
    const uint32_t nq = tb.V * tb.naz;
    if (!nq) return;
    const uint32_t nb = blocks_for(nq);
    auto launch = [&](uint32_t block0, uint32_t count) {
        launch_k(k_pack<true>, dim3(count), dim3(kBlock), 0, s, tb, t, gid, keys, block_counts, next_block_counts, big_count, gt,
                 reinterpret_cast<float4 *>(points32), reinterpret_cast<uint4 *>(hits), n_points, compact, block0, nb,
                 progress ? *progress : ProgressArgs{nullptr, 0u, 0u, nullptr});
    };
    if (rays_per_lane >= 2u && (!progress || !progress->host) && compact != 2u) {
        uint32_t *hint = progress ? progress->cull_hint : nullptr;
        float4 *p4 = reinterpret_cast<float4 *>(points32);
        uint4 *h4 = reinterpret_cast<uint4 *>(hits);
        if (rays_per_lane >= 8u)
            launch_k(k_pack_wide<8>, dim3((nb + 7u) / 8u), dim3(kBlock), 0, s, tb, keys, block_counts, next_block_counts, big_count, gt, p4, h4, n_points, compact, nb, hint);
        else if (rays_per_lane >= 4u)
            launch_k(k_pack_wide<4>, dim3((nb + 3u) / 4u), dim3(kBlock), 0, s, tb, keys, block_counts, next_block_counts, big_count, gt, p4, h4, n_points, compact, nb, hint);
        else
            launch_k(k_pack_wide<2>, dim3((nb + 1u) / 2u), dim3(kBlock), 0, s, tb, keys, block_counts, next_block_counts, big_count, gt, p4, h4, n_points, compact, nb, hint);
        return;
    }
    if (!progress || !progress->host) { launch(0u, nb); return; }
    // two launches, each followed by a word for the host: a kernel's writes to pinned host memory are complete when the
    // kernel is, so the word that follows it in the stream says "these records have arrived" without a fence per wave
    // (a system-scope fence in every workgroup made the pack pass wait for PCIe round trips: 0.2 ms instead of 0.08)
    // (a stream write-value packet where the runtime takes it -- cheaper on the host than a launch --, else a one-lane kernel)
    auto signal = [&](uint32_t *word) {
        static bool use_packet = true;
        if (use_packet && hipStreamWriteValue32(s, word, progress->epoch, 0) == hipSuccess) return;
        use_packet = false;
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, word, progress->epoch);
    };
    const uint32_t first = std::min(progress->split, nb);
    if (first) launch(0u, first);
    signal(&progress->host->half_epoch);
    if (nb > first) launch(first, nb - first);
    signal(&progress->host->all_epoch);
}

// One wave that does nothing for `ticks` of the 100 MHz wall clock: ls_tracer.cpp uses it to find out which
// streams share a hardware queue (two of these take twice as long on one queue as on two).
__global__ void k_spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0) *out = t0;
}

void launch_spin(hipStream_t s, unsigned long long ticks) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, static_cast<unsigned long long *>(nullptr)); }

void launch_cloud_to_world(hipStream_t s, const Affine &m, const void *in, const uint32_t *n_points, void *out,
                           const uint32_t *out_base, uint32_t *out_total, uint32_t capacity, uint32_t max_points)
{
    if (!max_points) return;
    hipLaunchKernelGGL(k_cloud_to_world, dim3(blocks_for(max_points)), dim3(kBlock), 0, s, m, static_cast<const float4 *>(in),
                       n_points, static_cast<float4 *>(out), out_base, out_total, capacity, max_points);
}

void launch_dense_from_hits(hipStream_t s, const SensorTables &tb, const void *hits, const uint32_t *n_points,
                            const GeomTable &gt, float *t, uint32_t *gid)
{
    const uint32_t nq = tb.V * tb.naz;
    if (!nq) return;
    hipLaunchKernelGGL(k_dense_clear, dim3(blocks_for(nq)), dim3(kBlock), 0, s, nq, t, gid);
    hipLaunchKernelGGL(k_dense_from_hits, dim3(blocks_for(nq)), dim3(kBlock), 0, s, tb, static_cast<const uint4 *>(hits), n_points,
                       gt, t, gid);
}

void launch_expand_slots(hipStream_t s, const SensorTables &tb, const uint32_t *gathered, uint32_t world, uint32_t cap,
                         uint32_t slot_words, uint8_t *points32, void *hits, uint32_t *n_points, GatherStat *stat, uint32_t epoch)
{
    if (!world || !cap) return;
    launch_k(k_expand_slots, dim3(blocks_for(world * cap)), dim3(kBlock), 0, s, tb, gathered, world, cap,
             slot_words, reinterpret_cast<float4 *>(points32), reinterpret_cast<uint4 *>(hits), n_points, stat, epoch);
}

void launch_raygen(hipStream_t s, const SensorTables &tb, float *dx, float *dy, float *dz)
{
    const uint32_t n = tb.V * tb.naz;
    if (!n) return;
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for(n)), dim3(kBlock), 0, s, tb, dx, dy, dz);
}

void launch_raygen_aos(hipStream_t s, const SensorTables &tb, void *rays32, void *hits24)
{
    const uint32_t n = tb.V * tb.naz;
    if (!n) return;
    hipLaunchKernelGGL(k_raygen_aos, dim3(blocks_for(n)), dim3(kBlock), 0, s, tb, static_cast<float4 *>(rays32), static_cast<float2 *>(hits24));
}

void launch_bruteforce(hipStream_t s, const SensorTables &tb, const float *verts, const uint32_t *tris,
                       uint32_t ntris, float *t_out, uint32_t *gid_out)
{
    const uint32_t n = tb.V * tb.naz;
    if (!n) return;
    hipLaunchKernelGGL(k_bruteforce, dim3(blocks_for(n)), dim3(kBlock), 0, s, tb, verts, tris, ntris, t_out,
                       gid_out);
}

}  // namespace ls
