// ls_project.hip -- "sensor-space projection" closest-hit engine (gfx950, wave64).
//
// Every ray of a LiDAR frame starts at the sensor origin and its direction is a cell (channel v,
// azimuth column h) of the sensor's angular raster (LidarDevice.cpp:306-316).  So instead of
// walking each ray through a hierarchy -- a chain of dependent, divergent memory fetches -- the
// triangles are streamed ONCE, coalesced; each triangle bounds its own footprint on the raster
// (conservatively: elevation band -> channel range, azimuth arc -> column ranges), runs the exact
// ray/triangle test of the reference (Embree 3.13.4 Moeller-Trumbore, same operation sequence as
// oracle/ls_oracle.c) against just those rays, and folds hits into the per-ray closest hit with one
// 64-bit atomicMin on (t bits, global triangle id): minimum t, ties to the lowest (geomID, primID).
// The result is the exhaustive closest hit, bit for bit, for any scene; the work is
// O(triangles + covered cells) with no dependent memory chain, i.e. it runs at HBM streaming rate.
//
// Replaces (with the BVH engine in ls_kernels.hip as the general-ray alternative):
// rtcIntersect16 over the committed scene (EmbreeTracer.cpp:297-367, :472-480) / optixLaunch
// (OptixTracer.cpp:317-328, OptixTracerModules.cu:26-86).
#include "ls_kernels.h"
#include "ls_device.h"

namespace ls {

namespace {

constexpr float kRadToDeg = 57.29577951308232f;
constexpr float kAngleMarginDeg = 0.02f;  // ~3.5e-4 rad: covers atan2f / table rounding and tri_test slop
constexpr uint32_t kInlineCols = 16;      // longer rows of cells go to the row queue

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

// squared distance from the 2-D origin to segment a-b
__device__ __forceinline__ float seg_dist2(float ax, float ay, float bx, float by)
{
    const float dx = bx - ax, dy = by - ay;
    const float len2 = dx * dx + dy * dy;
    float s = len2 > 0.0f ? -(ax * dx + ay * dy) / len2 : 0.0f;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    const float px = ax + s * dx, py = ay + s * dy;
    return px * px + py * py;
}

struct TriSetup {
    V3 v0, e1, e2;
    float NgC;
};

__device__ __forceinline__ void load_tri(const float *__restrict__ verts, const uint32_t *__restrict__ tris, uint32_t gid,
                                         V3 &v0, V3 &v1, V3 &v2)
{
    const uint32_t i0 = tris[3 * (size_t)gid + 0], i1 = tris[3 * (size_t)gid + 1], i2 = tris[3 * (size_t)gid + 2];
    v0 = {verts[3 * (size_t)i0], verts[3 * (size_t)i0 + 1], verts[3 * (size_t)i0 + 2]};
    v1 = {verts[3 * (size_t)i1], verts[3 * (size_t)i1 + 1], verts[3 * (size_t)i1 + 2]};
    v2 = {verts[3 * (size_t)i2], verts[3 * (size_t)i2 + 1], verts[3 * (size_t)i2 + 2]};
}

__device__ __forceinline__ TriSetup setup_tri(V3 v0, V3 v1, V3 v2)
{
    TriSetup s;
    s.v0 = v0;
    s.e1 = sub(v0, v1);
    s.e2 = sub(v2, v0);
    s.NgC = dot_fma(cross_fma(s.e2, s.e1), v0);
    return s;
}

// exact test of ray (v, h) against the triangle; fold a hit into the ray's closest-hit key
__device__ __forceinline__ void test_cell(const ProjectParams &pp, const TriSetup &ts, uint32_t gid, uint32_t v, uint32_t h,
                                          unsigned long long *__restrict__ best)
{
    const float st = pp.tb.sin_theta[v];
    const V3 d = {st * pp.tb.cos_phi[h], st * pp.tb.sin_phi[h], pp.tb.cos_theta[v]};
    float t;
    if (tri_test(d, ts.v0, ts.e1, ts.e2, ts.NgC, t)) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | gid;  // t > 0: bits order like t
        atomicMin(&best[(size_t)v * pp.tb.naz + (h - pp.tb.az0)], key);
    }
}

// ------------------------------------------------------------------------------------------
// One thread per triangle: conservative footprint on the (channel, column) raster, short rows
// tested in place, long rows pushed to the row queue for k_project_rows.
// Algorithmic bytes per triangle: 12 (indices) + 36 (vertex gather).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_project_tris(ProjectParams pp, const float *__restrict__ verts,
                                                         const uint32_t *__restrict__ tris, uint32_t ntris,
                                                         unsigned long long *__restrict__ best,
                                                         uint4 *__restrict__ rows, uint32_t row_capacity,
                                                         uint32_t *__restrict__ row_count,
                                                         unsigned long long *__restrict__ stats)
{
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    uint32_t ntest = 0;
    if (gid < ntris) {
        V3 v0, v1, v2;
        load_tri(verts, tris, gid, v0, v1, v2);

        // ---- elevation band: e = atan2(z, rho) over the triangle is inside
        //      [atan2(zmin, zmin >= 0 ? rho_max : rho_min), atan2(zmax, zmax > 0 ? rho_min : rho_max)]
        const float zmin = fminf(v0.z, fminf(v1.z, v2.z)), zmax = fmaxf(v0.z, fmaxf(v1.z, v2.z));
        const float r0 = v0.x * v0.x + v0.y * v0.y, r1 = v1.x * v1.x + v1.y * v1.y, r2 = v2.x * v2.x + v2.y * v2.y;
        const float rho_max = sqrtf(fmaxf(r0, fmaxf(r1, r2)));
        const float c0 = cross2(v0.x, v0.y, v1.x, v1.y), c1 = cross2(v1.x, v1.y, v2.x, v2.y),
                    c2 = cross2(v2.x, v2.y, v0.x, v0.y);
        const float ctol = 1e-6f * rho_max * rho_max;
        const bool inside = (c0 >= -ctol && c1 >= -ctol && c2 >= -ctol) || (c0 <= ctol && c1 <= ctol && c2 <= ctol);
        float rho_min = 0.0f;
        if (!inside)
            rho_min = sqrtf(fminf(seg_dist2(v0.x, v0.y, v1.x, v1.y),
                                  fminf(seg_dist2(v1.x, v1.y, v2.x, v2.y), seg_dist2(v2.x, v2.y, v0.x, v0.y))));
        rho_min *= 0.9999f;  // rounding slack, keeps the band conservative
        const float e_lo = atan2f(zmin, zmin >= 0.0f ? rho_max * 1.0001f : rho_min) * kRadToDeg - kAngleMarginDeg;
        const float e_hi = atan2f(zmax, zmax > 0.0f ? rho_min : rho_max * 1.0001f) * kRadToDeg + kAngleMarginDeg;

        // channels with elevation in [e_lo, e_hi]: a contiguous range of the sorted channel table
        uint32_t i0 = 0, i1 = pp.tb.V;
        {
            uint32_t lo = 0, hi = pp.tb.V;  // first index with chan_sorted >= e_lo
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (pp.chan_sorted[m] < e_lo) lo = m + 1; else hi = m; }
            i0 = lo;
            hi = pp.tb.V;                   // first index with chan_sorted > e_hi
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (pp.chan_sorted[m] <= e_hi) lo = m + 1; else hi = m; }
            i1 = lo;
        }
        if (i0 < i1) {
            // ---- azimuth arc: the three vertex azimuths minus the largest gap between them; if that
            //      gap is not larger than 180 deg the origin is inside the projection: full circle
            const uint32_t az_first = pp.tb.az0, az_last = pp.tb.az0 + pp.tb.naz - 1u;
            float phi_lo = 0.0f, span = 360.0f;
            bool full = inside || !(fabsf(pp.step_deg) > 0.0f);
            if (!full) {
                float a = atan2f(v0.y, v0.x) * kRadToDeg, b = atan2f(v1.y, v1.x) * kRadToDeg, c = atan2f(v2.y, v2.x) * kRadToDeg;
                float t;
                if (a > b) { t = a; a = b; b = t; }
                if (b > c) { t = b; b = c; c = t; }
                if (a > b) { t = a; a = b; b = t; }
                const float g0 = b - a, g1 = c - b, g2 = a + 360.0f - c;
                float maxgap = g2;
                phi_lo = a;                     // arc = [a, c]
                if (g0 > maxgap) { maxgap = g0; phi_lo = b; }   // arc = [b, a+360]
                if (g1 > maxgap) { maxgap = g1; phi_lo = c; }   // arc = [c, b+360]
                span = 360.0f - maxgap;
                if (!(maxgap > 180.0f + 2.0f * kAngleMarginDeg)) full = true;
            }
            const TriSetup ts = setup_tri(v0, v1, v2);
            // column index (real) of an azimuth: u = (phi - begin) / step; the raster repeats every P columns
            const float inv_step = full ? 0.0f : 1.0f / pp.step_deg;
            const float P = full ? 0.0f : 360.0f * fabsf(inv_step);
            float u_lo = 0.0f, u_hi = 0.0f;
            int n_min = 0, n_max = 0;
            if (!full) {
                const float ua = (phi_lo - kAngleMarginDeg - pp.begin_deg) * inv_step;
                const float ub = (phi_lo + span + kAngleMarginDeg - pp.begin_deg) * inv_step;
                u_lo = fminf(ua, ub) - 1.0f;   // one column of slack on both sides
                u_hi = fmaxf(ua, ub) + 1.0f;
                n_min = (int)ceilf(((float)az_first - u_hi) / P);
                n_max = (int)floorf(((float)az_last - u_lo) / P);
                if (u_hi - u_lo >= P) { full = true; }
            }
            if (full) { n_min = n_max = 0; }
            for (int n = n_min; n <= n_max; ++n) {
                uint32_t h0 = az_first, h1 = az_last;
                if (!full) {
                    const float lo = ceilf(u_lo + (float)n * P), hi = floorf(u_hi + (float)n * P);
                    if (hi < (float)az_first || lo > (float)az_last) continue;
                    h0 = (uint32_t)fmaxf(lo, (float)az_first);
                    h1 = (uint32_t)fminf(hi, (float)az_last);
                    if (h0 > h1) continue;
                }
                const uint32_t ncol = h1 - h0 + 1u;
                for (uint32_t i = i0; i < i1; ++i) {
                    const uint32_t v = pp.chan_perm[i];
                    bool inline_row = ncol <= kInlineCols;
                    if (!inline_row) {
                        const uint32_t slot = atomicAdd(row_count, 1u);
                        if (slot < row_capacity) rows[slot] = make_uint4(gid, v, h0, h1);
                        else inline_row = true;  // queue full: correct, just slower
                    }
                    if (inline_row) {
                        for (uint32_t h = h0; h <= h1; ++h) test_cell(pp, ts, gid, v, h, best);
                        ntest += ncol;
                    }
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ntest += __shfl_xor(ntest, off);
        if ((threadIdx.x & 63u) == 0 && ntest) atomicAdd(&stats[0], (unsigned long long)ntest);
    }
}

// ------------------------------------------------------------------------------------------
// Long rows (large triangles): one wave per queued row, lanes stride over its columns.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_project_rows(ProjectParams pp, const float *__restrict__ verts,
                                                         const uint32_t *__restrict__ tris,
                                                         unsigned long long *__restrict__ best,
                                                         const uint4 *__restrict__ rows, uint32_t row_capacity,
                                                         const uint32_t *__restrict__ row_count,
                                                         unsigned long long *__restrict__ stats)
{
    const uint32_t n_rows = min(*row_count, row_capacity);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    uint32_t ntest = 0;
    for (uint32_t r = wave; r < n_rows; r += n_waves) {
        const uint4 row = rows[r];
        V3 v0, v1, v2;
        load_tri(verts, tris, row.x, v0, v1, v2);
        const TriSetup ts = setup_tri(v0, v1, v2);
        for (uint32_t h = row.z + lane; h <= row.w; h += 64u) test_cell(pp, ts, row.x, row.y, h, best);
        if (lane == 0) ntest += row.w - row.z + 1u;
    }
    if (stats && lane == 0 && ntest) atomicAdd(&stats[0], (unsigned long long)ntest);
}

// per-ray closest-hit key -> dense t / gid arrays + hits per row of 64 rays (feeds the ordered pack)
__global__ __launch_bounds__(kBlock) void k_project_resolve(const unsigned long long *__restrict__ best, uint32_t n,
                                                            float *__restrict__ t_out, uint32_t *__restrict__ gid_out,
                                                            uint32_t *__restrict__ row_counts)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    bool hit = false;
    if (q < n) {
        const unsigned long long key = best[q];
        hit = key != ~0ull;
        t_out[q] = hit ? __uint_as_float((uint32_t)(key >> 32)) : -1.0f;
        gid_out[q] = hit ? (uint32_t)key : kInvalid;
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63u) == 0 && (q >> 6) < ((n + 63u) >> 6)) row_counts[q >> 6] = (uint32_t)__popcll(m);
}

}  // namespace

void launch_project_init(hipStream_t s, const ProjectParams &pp, unsigned long long *best, uint32_t *row_count)
{
    const uint32_t nq = pp.tb.V * pp.tb.naz;
    if (!nq) return;
    (void)hipMemsetAsync(best, 0xFF, (size_t)nq * sizeof(unsigned long long), s);
    (void)hipMemsetAsync(row_count, 0, sizeof(uint32_t), s);
}

void launch_project_tris(hipStream_t s, const ProjectParams &pp, const float *verts, const uint32_t *tris,
                         uint32_t ntris, unsigned long long *best, uint4 *rows, uint32_t row_capacity,
                         uint32_t *row_count, unsigned long long *stats)
{
    if (!ntris || !(pp.tb.V * pp.tb.naz)) return;
    hipLaunchKernelGGL(k_project_tris, dim3((ntris + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pp, verts, tris, ntris,
                       best, rows, row_capacity, row_count, stats);
}

void launch_project_rows(hipStream_t s, const ProjectParams &pp, const float *verts, const uint32_t *tris,
                         uint32_t ntris, unsigned long long *best, const uint4 *rows, uint32_t row_capacity,
                         const uint32_t *row_count, uint32_t grid_blocks, unsigned long long *stats)
{
    if (!ntris || !(pp.tb.V * pp.tb.naz)) return;
    hipLaunchKernelGGL(k_project_rows, dim3(grid_blocks), dim3(kBlock), 0, s, pp, verts, tris, best, rows, row_capacity,
                       row_count, stats);
}

void launch_project_resolve(hipStream_t s, const ProjectParams &pp, const unsigned long long *best, float *t_out,
                            uint32_t *gid_out, uint32_t *row_counts)
{
    const uint32_t nq = pp.tb.V * pp.tb.naz;
    if (!nq) return;
    hipLaunchKernelGGL(k_project_resolve, dim3((nq + kBlock - 1) / kBlock), dim3(kBlock), 0, s, best, nq, t_out, gid_out,
                       row_counts);
}

}  // namespace ls
