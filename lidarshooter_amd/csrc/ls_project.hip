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
// The vertex transform into the sensor frame (MeshTransformer.cpp:142-205 + LidarDevice.cpp:383-391)
// is fused in: the kernel reads the mesh as uploaded and transforms the three corners of each
// triangle on the fly (same arithmetic as k_transform, so the same bits).
//
// Replaces (with the BVH engine in ls_kernels.hip as the general-ray alternative):
// rtcIntersect16 over the committed scene (EmbreeTracer.cpp:297-367, :472-480) / optixLaunch
// (OptixTracer.cpp:317-328, OptixTracerModules.cu:26-86).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include "ls_kernels.h"
#include "ls_device.h"
#include "ls_launch.h"
#include "ls_tuning.h"

namespace ls {

namespace {

constexpr float kRadToDeg = 57.29577951308232f;

// The footprint bounds only have to be conservative, not exact: they use the hardware's 1-ulp
// reciprocal / square root and a polynomial arctangent, and every use carries explicit slack
// (factors 0.9999 / 1.0001, the angular margin, 1/16 raster column per side).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

// atan2 in degrees, |error| < 1e-3 deg (Abramowitz & Stegun 4.4.49 on [0,1] + octant folding)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? mn * fast_rcp(mx) : 0.0f;
    const float s = a * a;
    float r = a * (0.9999772256f + s * (-0.3326234763f + s * (0.1935434479f + s * (-0.1164328762f +
                  s * (0.0526533153f + s * -0.0117212052f)))));
    r *= kRadToDeg;
    if (ay > ax) r = 90.0f - r;
    if (x < 0.0f) r = 180.0f - r;
    return y < 0.0f ? -r : r;
}

// Footprint of a triangle on the raster: channels chan_perm[i0 .. i0+nch) x columns
// [h0a, h0a+na) u [h0b, h0b+nb).  A superset of the rays that can hit the triangle.
struct Foot {
    uint32_t i0, nch, h0a, na, h0b, nb;
};

// per-channel tables, staged in LDS by k_project (global memory when V is too large for that)
struct ChanTables {
    const float *tan_up, *tan_dn, *sin_theta, *cos_theta;
    const uint32_t *perm;
    const float2 *cols;   // (cos phi, sin phi) of the shard's columns [az0, az0 + naz) in LDS (ProjectParams::cols_lds), else nullptr
};

// The footprint is computed in two steps: band() (elevation -> channel range; run for every triangle --
// most triangles of a large scene fall between two channels and end there) and columns() (azimuth ->
// column intervals; only for the survivors).
// channels whose elevation (widened by the angular margin) can meet points with z in [zmin, zmax] at horizontal
// distance in [rho_min, rho_max] from the vertical axis (rho_max already carries its 1e-4 slack, rho_min its own)
__device__ __forceinline__ void band_from_ranges(const ProjectParams &pp, const ChanTables &ct, float zmin, float zmax, float rho_min,
                                                 float rho_max, uint32_t &i0_out, uint32_t &nch_out)
{
    float tan_lo = -INFINITY, tan_hi = INFINITY;
    if (rho_max > 0.0f) {
        // quotients rounded outwards by 1e-5 relative (the reciprocal is good to 1 ulp)
        const float inv_max = fast_rcp(rho_max), inv_min = rho_min > 0.0f ? fast_rcp(rho_min) : 0.0f;
        if (zmin >= 0.0f) tan_lo = zmin * inv_max * 0.99999f;
        else if (rho_min > 0.0f) tan_lo = zmin * inv_min * 1.00001f;
        if (zmax <= 0.0f) tan_hi = zmax * inv_max * 0.99999f;
        else if (rho_min > 0.0f) tan_hi = zmax * inv_min * 1.00001f;
    }
    // channels whose elevation (widened by the angular margin) meets the band: chan_tan_up[i] =
    // tan(chi_i + margin), chan_tan_dn[i] = tan(chi_i - margin), both ascending
    uint32_t lo = 0, hi = pp.tb.V;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (ct.tan_up[m] < tan_lo) lo = m + 1; else hi = m; }
    const uint32_t i0 = lo;
    uint32_t i1 = i0;  // bands are narrow: walk forward instead of a second search
    while (i1 < pp.tb.V && ct.tan_dn[i1] <= tan_hi) ++i1;
    i0_out = i0;
    nch_out = i1 > i0 ? i1 - i0 : 0u;
}

__device__ __forceinline__ void band(const ProjectParams &pp, const ChanTables &ct, V3 v0, V3 v1, V3 v2, uint32_t &i0_out,
                                     uint32_t &nch_out)
{
    // ---- elevation band in tangent space: tan(e) = z / rho over the triangle lies inside
    //      [zmin / (zmin >= 0 ? rho_max : rho_min), zmax / (zmax > 0 ? rho_min : rho_max)]
    // rho_max = largest vertex rho.  rho_min: for the unit vector u towards the centroid (2-D), every
    // point p of the triangle has |p| >= p.u >= min_i v_i.u -- exact to second order in the angle the
    // triangle subtends, and <= 0 (-> 0) whenever the vertical axis pierces the triangle.
    const float zmin = fminf(v0.z, fminf(v1.z, v2.z)), zmax = fmaxf(v0.z, fmaxf(v1.z, v2.z));
    const float r0 = v0.x * v0.x + v0.y * v0.y, r1 = v1.x * v1.x + v1.y * v1.y, r2 = v2.x * v2.x + v2.y * v2.y;
    const float rho_max = fast_sqrt(fmaxf(r0, fmaxf(r1, r2))) * 1.0001f;
    const float cx = (v0.x + v1.x) + v2.x, cy = (v0.y + v1.y) + v2.y;
    const float cc = cx * cx + cy * cy;
    float rho_min = 0.0f;
    if (cc > 1e-30f) {
        const float p0 = v0.x * cx + v0.y * cy, p1 = v1.x * cx + v1.y * cy, p2 = v2.x * cx + v2.y * cy;
        // 1e-4 relative for the reciprocal square root and |u|, 1e-6 rho_max absolute for the rounding of the dots
        rho_min = fmaxf(fminf(p0, fminf(p1, p2)) * __builtin_amdgcn_rsqf(cc) * 0.9999f - 1e-6f * rho_max, 0.0f);
    }
    band_from_ranges(pp, ct, zmin, zmax, rho_min, rho_max, i0_out, nch_out);
}

__device__ __forceinline__ void columns(const ProjectParams &pp, V3 v0, V3 v1, V3 v2, uint32_t &h0a, uint32_t &na, uint32_t &h0b,
                                        uint32_t &nb)
{
    h0a = na = h0b = nb = 0;

    // ---- azimuth arc.  Small triangles (both other vertices within 14 deg of vertex 0 as seen from
    //      the vertical axis -- the bulk of a large scene) need one arctangent: the offsets come from
    //      tan(delta) = cross / dot, bounded by the alternating series.  Otherwise: the three vertex
    //      azimuths minus the largest gap between them; if that gap is not larger than 180 deg the
    //      axis pierces the triangle: full circle.
    const uint32_t az_first = pp.tb.az0, az_last = pp.tb.az0 + pp.tb.naz - 1u;
    bool full = !(fabsf(pp.step_deg) > 0.0f);
    float phi_lo = 0.0f, span = 360.0f;
    if (!full) {
        const float d1 = v0.x * v1.x + v0.y * v1.y, c1 = cross2(v0.x, v0.y, v1.x, v1.y);
        const float d2 = v0.x * v2.x + v0.y * v2.y, c2 = cross2(v0.x, v0.y, v2.x, v2.y);
        if (d1 > 0.0f && d2 > 0.0f && fabsf(c1) <= 0.25f * d1 && fabsf(c2) <= 0.25f * d2) {
            // |atan x| <= |x (1 - x^2/3 + x^4/5)| for |x| < 1 (truncated after a positive term)
            const float x1 = c1 * fast_rcp(d1), x2 = c2 * fast_rcp(d2);
            const float s1 = x1 * x1, s2 = x2 * x2;
            const float q1 = x1 * (1.0f - s1 * (0.3333333f - s1 * 0.2f)), q2 = x2 * (1.0f - s2 * (0.3333333f - s2 * 0.2f));
            const float lo = fminf(0.0f, fminf(q1, q2)), hi = fmaxf(0.0f, fmaxf(q1, q2));
            phi_lo = fast_atan2_deg(v0.y, v0.x) + lo * (kRadToDeg * 1.0001f);
            span = (hi - lo) * (kRadToDeg * 1.0001f);
        } else {
            const float r0 = v0.x * v0.x + v0.y * v0.y, r1 = v1.x * v1.x + v1.y * v1.y, r2 = v2.x * v2.x + v2.y * v2.y;
            const float ctol = 1.0002e-6f * fmaxf(r0, fmaxf(r1, r2));
            const float c3 = cross2(v1.x, v1.y, v2.x, v2.y);   // v0 x v1 = c1, v1 x v2 = c3, v2 x v0 = -c2
            const bool inside = (c1 >= -ctol && c3 >= -ctol && -c2 >= -ctol) || (c1 <= ctol && c3 <= ctol && -c2 <= ctol);
            if (inside) {
                full = true;
            } else {
                float a = fast_atan2_deg(v0.y, v0.x), b = fast_atan2_deg(v1.y, v1.x), c = fast_atan2_deg(v2.y, v2.x);
                float t;
                if (a > b) { t = a; a = b; b = t; }
                if (b > c) { t = b; b = c; c = t; }
                if (a > b) { t = a; a = b; b = t; }
                const float g0 = b - a, g1 = c - b, g2 = a + 360.0f - c;
                float maxgap = g2;
                phi_lo = a;                                    // arc [a, c]
                if (g0 > maxgap) { maxgap = g0; phi_lo = b; }  // arc [b, a + 360]
                if (g1 > maxgap) { maxgap = g1; phi_lo = c; }  // arc [c, b + 360]
                span = 360.0f - maxgap;
                if (!(maxgap > 180.0f + 2.0f * pp.margin_deg)) full = true;
            }
        }
    }
    if (!full) {
        // column index (real) of an azimuth: u = (phi - begin) / step; the raster repeats every P columns
        const float inv_step = pp.inv_step_deg;
        const float P = 360.0f * fabsf(inv_step), invP = pp.inv_period;
        const float ua = (phi_lo - pp.margin_deg - pp.begin_deg) * inv_step;
        const float ub = (phi_lo + span + pp.margin_deg - pp.begin_deg) * inv_step;
        // slack per side: the angular margin (in ua/ub) + 1/16 column for the float rounding of u itself
        const float u_lo = fminf(ua, ub) - 0.0625f, u_hi = fmaxf(ua, ub) + 0.0625f;
        // n with [u_lo, u_hi] + n*P meeting [az_first, az_last]; half a column of slack absorbs invP's rounding
        const int n_min = (int)ceilf(((float)az_first - u_hi - 0.5f) * invP);
        const int n_max = (int)floorf(((float)az_last - u_lo + 0.5f) * invP);
        if (u_hi - u_lo >= P || n_max - n_min > 1) {
            full = true;  // more than two pieces: take the whole revolution (a superset)
        } else {
            for (int n = n_min; n <= n_max; ++n) {
                const float l = ceilf(u_lo + (float)n * P), h = floorf(u_hi + (float)n * P);
                if (h < (float)az_first || l > (float)az_last) continue;
                const uint32_t h0 = (uint32_t)fmaxf(l, (float)az_first), h1 = (uint32_t)fminf(h, (float)az_last);
                if (h0 > h1) continue;
                if (n == n_min) { h0a = h0; na = h1 - h0 + 1u; }
                else { h0b = h0; nb = h1 - h0 + 1u; }
            }
        }
    }
    if (full) { h0a = az_first; na = az_last - az_first + 1u; h0b = 0; nb = 0; }
}

// cell m of a footprint -> (channel, column)
__device__ __forceinline__ void foot_cell(const ChanTables &ct, uint32_t i0, uint32_t h0a, uint32_t na, uint32_t h0b,
                                          uint32_t nb, uint32_t m, uint32_t &v, uint32_t &h)
{
    // m / ncol by reciprocal with a one-step correction (m < 2^24 cells per footprint in practice;
    // larger values fall back to the exact division)
    const uint32_t ncol = na + nb;
    uint32_t r;
    if (m < (1u << 24)) {
        r = (uint32_t)((float)m * fast_rcp((float)ncol));
        if (r * ncol > m) --r;
        if ((r + 1u) * ncol <= m) ++r;
    } else {
        r = m / ncol;
    }
    const uint32_t c = m - r * ncol;
    v = ct.perm[i0 + r];
    h = c < na ? h0a + c : h0b + (c - na);
}

// exact test of ray (v, h) against the triangle; fold a hit into the ray's closest-hit key
__device__ __forceinline__ void test_cell(const ProjectParams &pp, const ChanTables &ct, V3 v0, V3 e1, V3 e2, float NgC,
                                          uint32_t gid, uint32_t v, uint32_t h, unsigned long long *__restrict__ best)
{
    // LidarDevice.cpp:310-316: d = (sin(theta)cos(phi), sin(theta)sin(phi), cos(theta))
    const float st = ct.sin_theta[v];
#ifdef LS_EXP_COLS_LDS
    const float2 cs = ct.cols ? ct.cols[h - pp.tb.az0] : pp.tb.cs_phi[h];   // (experiment, tools/exp_build.sh: a small shard's column directions from LDS)
#else
    const float2 cs = pp.tb.cs_phi[h];
#endif
    const V3 d = {st * cs.x, st * cs.y, ct.cos_theta[v]};
    float t;
    if (tri_test(d, v0, e1, e2, NgC, t)) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | gid;  // t > 0: bits order like t
        // no-return atomic: nothing in the loop waits for it (a returning one costs a round trip per iteration)
        atomicMin(&best[(size_t)v * pp.tb.naz + (h - pp.tb.az0)], key);
    }
}

// 80-byte queue entry for a triangle whose footprint is too large for one wave
struct BigItem {
    float v0[3], e1[3], e2[3], NgC;
    uint32_t gid, i0, nch, h0a, na, h0b, nb, pad[3];
};
static_assert(sizeof(BigItem) == 80, "BigItem must be 80 bytes");

// ------------------------------------------------------------------------------------------
// One lane per triangle; each wave then spreads the cells of its 64 footprints evenly over its
// lanes (prefix sum + search), so a wave's cost is its total cell count / 64, not its largest
// footprint.  Algorithmic bytes per triangle: 12 (indices) + 36 (vertex gather); per hit 8.
// ------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes of a wave with DPP (row shifts inside each row of 16, then the
// two row broadcasts): twelve dependent vector instructions, no LDS round trips (a __shfl_up ladder is
// six ds_bpermute round trips; measured -0.8 us on the headline k_project).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}

// LDS of a k_project workgroup (also the scratch of the finish + pack workgroups of k_frame)
struct ProjectLds {
    float tri[kBlock / 64][10][64];      // v0, e1, e2, NgC of the wave's triangles
    uint32_t meta[kBlock / 64][6][64];   // gid, i0, h0a, na, h0b, nb
    uint32_t pref[kBlock / 64][64];      // first cell of every staged footprint (exclusive prefix of the cell counts)
    uint8_t flag[kBlock / 64][64];       // expansion: 1 at the first cell of a footprint inside the current 64-cell chunk
};

// orders this wave's LDS writes before its later LDS reads (LDS executes a wave's operations in order;
// the fence only stops the compiler from moving them)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Per-wave timeline instrumentation of k_project (tools/timeline/, build option -DLS_WAVE_TIMELINE of tools/exp_build.sh;
// DESIGN.md's per-phase numbers come from it): the marks in project_body are empty in the shipped library.
#ifdef LS_WAVE_TIMELINE
#include "../../tools/timeline/ls_wave_timeline.inc"
#else
#define TL_BEGIN()
#define TL_MARK(x)
#define TL_LOADS_LANDED(raw, best)
#define TL_END(block_idx, w, lane, total)
#define TL_HOST_HOOK(s)
#endif

// DEAL (with CULLED): a segment's survivors are dealt to its waves at a stride instead of taken in runs (azimuth shards).  A
// template argument, not a flag read at run time: behind a uniform branch the list entry's read-ahead (it goes out together
// with the segment's count) turned into a dependent load and the full raster's culled launch at ten million triangles took
// 36.3 us instead of 33.9.
template <bool COUNT, bool LDS_TABLES, bool MULTI, bool CULLED, bool DEAL = false>
__device__ __forceinline__ void project_body(const ProjectParams &pp, const GeomBatch &batch, uint32_t block_idx, ProjectLds &lds,
                                             float *s_chan /* LDS_TABLES: tan_up, tan_dn, sin_theta, cos_theta, perm */,
                                             unsigned long long *__restrict__ best, BigItem *__restrict__ big,
                                             uint32_t big_capacity, uint32_t *__restrict__ big_count,
                                             unsigned long long *__restrict__ stats, const uint32_t *__restrict__ cull_list)
{
    auto &s_tri = lds.tri;
    auto &s_meta = lds.meta;
    auto &s_pref = lds.pref;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TL_BEGIN();
    TL_MARK(tl_0);
    // one launch covers up to kGeomsPerLaunch geometries: the workgroup finds its own (uniform)
    uint32_t gi = 0;
    if (MULTI)
        while (gi + 1u < batch.n && block_idx >= batch.block_first[gi + 1u]) ++gi;
    const GeomSource &src = batch.g[gi];
    uint32_t block = MULTI ? block_idx - batch.block_first[gi] : block_idx;
    if (!MULTI && !CULLED && pp.xcd_remap) {
        // workgroup ids go round the eight XCDs: every XCD takes one contiguous eighth of the triangles, so that the vertex
        // lines two neighbouring workgroups share sit in ONE L2 (PMC: 24 MB of vertex lines fetched for 6 MB of vertices).
        // Only with one frame in flight (trace_locked): there it takes 1.2 us off the frame (26.3 -> 25.1 us, the kernel
        // 18.4 -> 17.5 us); with three frames in flight eight address streams an eighth of the mesh apart cost 2 us per frame
        const uint32_t nb = batch.block_first[1];
        const uint32_t x = block & 7u, i = block >> 3, q = nb >> 3, r = nb & 7u;
        block = x * q + min(x, r) + i;
    }
    // CULLED: the groups of kCullGroup sorted triangles that survived k_cull sit in kCullSegs segments of this
    // geometry's part of cull_list, each packed to its front (count in the segment's counter); workgroup b of the
    // geometry works on segment b % kCullSegs, position b / kCullSegs; a wave takes 64 / kCullGroup survivors, so its
    // lanes are dense; workgroups behind a segment's survivors have nothing to do.  The grid holds batch.seg_blocks[gi]
    // workgroups per segment -- room for half of the groups to survive, not for all: at 10 M triangles three quarters
    // of a worst-case grid were workgroups that found nothing, 120 000 waves to launch and retire -- and a wave whose
    // segment holds more than that comes round again (the loop at the end)
    constexpr uint32_t kPerWave = 64u / kCullGroup;
    uint32_t n_live = 0, n_deal = 0, seg = 0, seg_block = 0;
    uint32_t rank = 0, first_entry = 0;   // CULLED: this wave among its segment's; its lanes' list entry, read ahead
    const uint32_t *seg_list = nullptr;
    if (CULLED) {
        seg = block % kCullSegs;
        seg_block = block / kCullSegs;
        rank = seg_block * (kBlock / 64) + w;
        // the lanes' list entry goes out TOGETHER with the segment's count, not behind it: whether the entry is a survivor
        // of this frame (position < count) is known when both have arrived -- one memory round trip less in a chain of
        // dependent ones (count -> entry -> corners), which is what this kernel's waves spend their time in.  Positions
        // behind the count hold survivors of earlier frames or nothing: read (inside the segment), never used
        seg_list = cull_list + batch.list_first[gi] + (size_t)seg * batch.seg_cap[gi];
        if (!DEAL) first_entry = seg_list[min(rank * kPerWave + lane / kCullGroup, batch.seg_cap[gi] - 1u)];
        n_live = (uint32_t)__builtin_amdgcn_readfirstlane((int)big_count[kCullCountAt + (gi * kCullSegs + seg) * 16u]);
        // DEALT (DEAL, azimuth shards): the segment's n_live survivors go to its W = ceil(n_live / kPerWave) waves like
        // cards -- wave r takes entries r, r + W, r + 2 W ... -- instead of kPerWave consecutive ones each.  Consecutive
        // survivors are Morton neighbours; a shard's few hundred waves leave the chip three quarters empty, so its kernel
        // is as long as its heaviest wave, and the waves that hold nothing but the sector's nearest ground walked 12 trips of
        // 64 cells where the mean is 2.4 (tests/analysis/shard_balance.py: dealt, the heaviest walks 4).  Costs the memory
        // round trip the read-ahead above saves (the stride needs the count first)
        n_deal = DEAL ? (n_live + kPerWave - 1u) / kPerWave : 0u;
        if (DEAL ? seg_block * (kBlock / 64) >= n_deal : seg_block * (kBlock / 64) * kPerWave >= n_live) return;   // uniform over the workgroup: before any barrier
    }
    // ---- which triangle this lane takes, and its loads, BEFORE the channel tables are staged: index load -> vertex
    //      gather is a chain of two memory round trips, the staging (global -> LDS, then a workgroup barrier) a third
    //      that does not depend on them -- issued first, the chain runs under the staging instead of after it
    //      (rocprofv3: the waves of this kernel spend half their lifetime in s_waitcnt; eight vector loads per wave)
    uint32_t k = 0xFFFFFFFFu;
    bool live_wave = true;
    if (CULLED) {
        if (COUNT && rank == 0 && lane == 0) atomicAdd(&stats[1], (unsigned long long)n_live);   // counts[2]: surviving groups
        // (a full turn's wave takes consecutive survivors; taking them at a stride, as the unculled path's spread runs do,
        // was within noise at 10 M triangles and puts the count in front of the entry load again)
        if (DEAL) {
            live_wave = rank < n_deal;
            const uint32_t e = (lane / kCullGroup) * n_deal + rank;
            if (live_wave && e < n_live) k = seg_list[e] * kCullGroup + (lane % kCullGroup);
        } else {
            live_wave = rank * kPerWave < n_live;              // the survivors' last workgroup is partly filled
            if (rank * kPerWave + lane / kCullGroup < n_live) k = first_entry * kCullGroup + (lane % kCullGroup);
        }
    } else {
        // a small mesh is cut into more waves than triangles / 64 (tris_per_wave < 64, the upper lanes only
        // join the cell tests): its footprints are large, and the cells are what takes the time
        const uint32_t tris_per_wave = batch.tris_per_wave[gi];
        if (tris_per_wave == 64u && pp.spread) {
            // big mesh: eight runs of eight triangles, a stride of the wave count apart (same reason as above: a wave's
            // share of the near field, where the cells are, is then the same for every wave)
            const uint32_t n_waves = (src.ntris + 63u) / 64u, rank = block * (kBlock / 64) + w;
            // (the grid is rounded up to whole workgroups: a wave behind the last one would alias onto the next run of
            // waves 0..2 and project those triangles a second time)
            if (rank < n_waves) k = ((lane >> 3) * n_waves + rank) * 8u + (lane & 7u);
        } else if (lane < tris_per_wave) {
            k = (block * (kBlock / 64) + w) * tris_per_wave + lane;
        }
    }
    float raw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // the three corners as uploaded (mesh space)
    uint32_t tri_id = 0;   // CULLED: the triangle's number in the caller's order (perm[k], stored with the corners)
    auto load_corners = [&]() {
        if (k < src.ntris) {
            if (CULLED) {
                const float4 *c = src.corners + 3 * (size_t)k;
                const float4 c0 = c[0], c1 = c[1], c2 = c[2];
                raw[0] = c0.x; raw[1] = c0.y; raw[2] = c0.z;
                raw[3] = c1.x; raw[4] = c1.y; raw[5] = c1.z;
                raw[6] = c2.x; raw[7] = c2.y; raw[8] = c2.z;
                tri_id = __float_as_uint(c0.w);
                return;
            }
            const uint32_t a = src.idx[3 * (size_t)k + 0], b = src.idx[3 * (size_t)k + 1], c = src.idx[3 * (size_t)k + 2];
            const float *pa = reinterpret_cast<const float *>(src.verts + (size_t)a * src.stride);
            const float *pb = reinterpret_cast<const float *>(src.verts + (size_t)b * src.stride);
            const float *pc = reinterpret_cast<const float *>(src.verts + (size_t)c * src.stride);
            raw[0] = pa[0]; raw[1] = pa[1]; raw[2] = pa[2];
            raw[3] = pb[0]; raw[4] = pb[1]; raw[5] = pb[2];
            raw[6] = pc[0]; raw[7] = pc[1]; raw[8] = pc[2];
        }
    };
    load_corners();
    ChanTables ct = {pp.chan_tan_up, pp.chan_tan_dn, pp.tb.sin_theta, pp.tb.cos_theta, pp.chan_perm, nullptr};
    if (LDS_TABLES) {
        const uint32_t V = pp.tb.V;
        for (uint32_t i = threadIdx.x; i < V; i += kBlock) {
            s_chan[i] = pp.chan_tan_up[i];
            s_chan[V + i] = pp.chan_tan_dn[i];
            s_chan[2 * V + i] = pp.tb.sin_theta[i];
            s_chan[3 * V + i] = pp.tb.cos_theta[i];
            s_chan[4 * V + i] = __uint_as_float(pp.chan_perm[i]);
        }
        ct = {s_chan, s_chan + V, s_chan + 2 * V, s_chan + 3 * V, reinterpret_cast<const uint32_t *>(s_chan + 4 * V), nullptr};
#ifdef LS_EXP_COLS_LDS
        if (pp.cols_lds) {
            // a small shard's column directions too (behind the channel tables, eight-byte aligned): 4 KB for an eighth of
            // 4 096 columns.  The full raster's 32 KB would cost the kernel its residency (+4 KB of LDS per
            // workgroup cost 2 us there); a shard's few hundred workgroups leave the chip's LDS empty anyway
            float2 *s_cols = reinterpret_cast<float2 *>(s_chan + ((5u * V + 1u) & ~1u));
            for (uint32_t i = threadIdx.x; i < pp.tb.naz; i += kBlock) s_cols[i] = pp.tb.cs_phi[pp.tb.az0 + i];
            ct.cols = s_cols;
        }
#endif
        __syncthreads();
    }
    if (!live_wave) return;   // (no barrier follows)
    TL_MARK(tl_1);   // tables staged
    TL_LOADS_LANDED(raw, best);
    TL_MARK(tl_2);
    uint32_t total = 0;
    for (;;) {   // (once, except for a CULLED wave whose segment holds more survivors than the grid has room for)
    uint32_t cells = 0, slot = 0;
    if (k < src.ntris) {
        V3 v0, v1, v2;
        if (src.xform == 1) {
            v0 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw));
            v1 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw + 3));
            v2 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw + 6));
        } else if (src.xform == 2) {  // mesh transform is exactly the identity: sensor pose only
            v0 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw));
            v1 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw + 3));
            v2 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw + 6));
        } else {
            v0 = {raw[0], raw[1], raw[2]}; v1 = {raw[3], raw[4], raw[5]}; v2 = {raw[6], raw[7], raw[8]};
        }
        if (pp.debug == 1) { if (v0.x + v1.y + v2.z == 12345.678f) best[0] = 0; return; }
        // multi-GPU shards: a triangle wholly on the outer side of one of the two vertical planes that bound
        // the shard's azimuth sector (padded; only when the sector is narrower than 180 degrees) cannot be hit
        // by its rays.  Meshes are spatially coherent, so whole waves skip the footprint and the cell tests.
        bool outside = false;
        if (pp.sector_on) {
            const float a0 = pp.sec_a[0] * v0.y - pp.sec_a[1] * v0.x, a1 = pp.sec_a[0] * v1.y - pp.sec_a[1] * v1.x,
                        a2 = pp.sec_a[0] * v2.y - pp.sec_a[1] * v2.x;   // cross(d_a, p): < 0 right of the first boundary
            const float b0 = v0.x * pp.sec_b[1] - v0.y * pp.sec_b[0], b1 = v1.x * pp.sec_b[1] - v1.y * pp.sec_b[0],
                        b2 = v2.x * pp.sec_b[1] - v2.y * pp.sec_b[0];   // cross(p, d_b): < 0 left of the second boundary
            outside = (a0 < 0.0f && a1 < 0.0f && a2 < 0.0f) || (b0 < 0.0f && b1 < 0.0f && b2 < 0.0f);
        }
        Foot f = {0, 0, 0, 0, 0, 0};
        if (!outside) band(pp, ct, v0, v1, v2, f.i0, f.nch);
        if (pp.debug == 3) { if (f.nch == 0xFFFFFFFFu) best[0] = 0; return; }
        if (f.nch) columns(pp, v0, v1, v2, f.h0a, f.na, f.h0b, f.nb);
        cells = f.nch * (f.na + f.nb);
        if (pp.debug == 2) { if (cells == 0xFFFFFFFFu) best[0] = 0; return; }
        if (cells) {
            const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
            const float NgC = dot_fma(cross_fma(e2, e1), v0);
            // (a geometry with culling data is kept in Morton order; when it goes through the unculled launch after all -- more
            // than kGeomsPerLaunch such geometries -- perm carries the sorted position back to the caller's triangle)
            const uint32_t gid = src.gid_first + (CULLED ? tri_id : (src.perm ? src.perm[k] : k));
            bool queued = false;
            if (cells > pp.big_cells) {
                const uint32_t slot = atomicAdd(big_count, 1u);
                if (slot < big_capacity) {
                    BigItem it;
                    it.v0[0] = v0.x; it.v0[1] = v0.y; it.v0[2] = v0.z;
                    it.e1[0] = e1.x; it.e1[1] = e1.y; it.e1[2] = e1.z;
                    it.e2[0] = e2.x; it.e2[1] = e2.y; it.e2[2] = e2.z;
                    it.NgC = NgC; it.gid = gid; it.i0 = f.i0; it.nch = f.nch;
                    it.h0a = f.h0a; it.na = f.na; it.h0b = f.h0b; it.nb = f.nb;
                    it.pad[0] = it.pad[1] = it.pad[2] = 0;
                    big[slot] = it;
                    queued = true;
                }
            }
            if (queued) {
                cells = 0;
            } else {
                // the lanes in this branch are exactly the footprints the wave expands itself: they are
                // staged packed to the front (slot = rank among them), which makes "the footprint that
                // owns cell j" a rank, computable with a ballot (below)
                slot = lanes_below(__ballot(true));
                s_tri[w][0][slot] = v0.x; s_tri[w][1][slot] = v0.y; s_tri[w][2][slot] = v0.z;
                s_tri[w][3][slot] = e1.x; s_tri[w][4][slot] = e1.y; s_tri[w][5][slot] = e1.z;
                s_tri[w][6][slot] = e2.x; s_tri[w][7][slot] = e2.y; s_tri[w][8][slot] = e2.z;
                s_tri[w][9][slot] = NgC;
                s_meta[w][0][slot] = gid; s_meta[w][1][slot] = f.i0; s_meta[w][2][slot] = f.h0a;
                s_meta[w][3][slot] = f.na; s_meta[w][4][slot] = f.h0b; s_meta[w][5][slot] = f.nb;
            }
        }
    }
    TL_MARK(tl_3);   // footprints done
    // wave-level inclusive scan of the cell counts
    const uint32_t incl = wave_inclusive_scan(cells);
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t n_slots = (uint32_t)__popcll(__ballot(cells != 0));
    if (cells) s_pref[w][slot] = incl - cells;
    wave_lds_fence();
    // lane s < n_slots holds the first cell of staged footprint s
    const uint32_t first = lane < n_slots ? s_pref[w][lane] : 0xFFFFFFFFu;
    for (uint32_t jb = 0; jb < total; jb += 64u) {
        // the footprint that owns cell jb + lane, by rank: footprints that started before this chunk +
        // starts inside the chunk up to this lane (a flag per first cell, then one ballot) - 1
        lds.flag[w][lane] = 0;
        wave_lds_fence();
        if (first - jb < 64u) lds.flag[w][first - jb] = 1;
        wave_lds_fence();
        const uint32_t before = (uint32_t)__popcll(__ballot(first < jb));
        const unsigned long long starts = __ballot(lds.flag[w][lane] != 0);
        const uint32_t j = jb + lane;
        if (j < total) {
            const uint32_t lo = before + lanes_below(starts) + (uint32_t)((starts >> lane) & 1ull) - 1u;
            const uint32_t m = j - s_pref[w][lo];
            uint32_t v, h;
            foot_cell(ct, s_meta[w][1][lo], s_meta[w][2][lo], s_meta[w][3][lo], s_meta[w][4][lo], s_meta[w][5][lo], m, v, h);
            test_cell(pp, ct, {s_tri[w][0][lo], s_tri[w][1][lo], s_tri[w][2][lo]}, {s_tri[w][3][lo], s_tri[w][4][lo], s_tri[w][5][lo]},
                      {s_tri[w][6][lo], s_tri[w][7][lo], s_tri[w][8][lo]}, s_tri[w][9][lo], s_meta[w][0][lo], v, h, best);
        }
        wave_lds_fence();
    }
    if (COUNT && lane == 0 && total) atomicAdd(&stats[0], (unsigned long long)total);
    if (!CULLED) break;
    rank += batch.seg_blocks[gi] * (kBlock / 64);   // the segment's next wave-load that nobody else takes
    if (DEAL ? rank >= n_deal : rank * kPerWave >= n_live) break;
    const uint32_t e = DEAL ? (lane / kCullGroup) * n_deal + rank : rank * kPerWave + lane / kCullGroup;
    k = e < n_live ? seg_list[e] * kCullGroup + (lane % kCullGroup) : 0xFFFFFFFFu;
    load_corners();
    }
    TL_MARK(tl_4);
    TL_END(block_idx, w, lane, total);
}

template <bool COUNT, bool LDS_TABLES, bool MULTI, bool CULLED, bool DEAL = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_project(ProjectParams pp, GeomBatch batch,
                                                    unsigned long long *__restrict__ best, BigItem *__restrict__ big,
                                                    uint32_t big_capacity, uint32_t *__restrict__ big_count,
                                                    unsigned long long *__restrict__ stats, const uint32_t *__restrict__ cull_list)
{
    __shared__ ProjectLds lds;
    extern __shared__ __attribute__((aligned(16))) float s_chan[];
    project_body<COUNT, LDS_TABLES, MULTI, CULLED, DEAL>(pp, batch, blockIdx.x, lds, s_chan, best, big, big_capacity, big_count, stats, cull_list);
}


// ------------------------------------------------------------------------------------------
// Group culling.  A mesh large enough for 64 triangles per wave is kept in Morton order (centroids in mesh space;
// one-off per topology: k_mesh_aabb, k_mesh_morton, radix sort, k_permute_indices) together with a bound for every
// kCullGroup consecutive sorted triangles (k_group_bounds, once per vertex upload, 32 bytes per group): a sheared
// box  { c + a e1 + b e2 + g e3 : |a|,|b|,|g| <= 1 },  e1 = (hx, 0, sx hx), e2 = (0, hy, sy hy), e3 = (0, 0, hz),
// i.e. the footprint rectangle of the group, tilted along its least-squares plane, hz thick.  Per frame k_cull --
// one lane per group -- carries centre and edges into the sensor frame with THIS frame's matrix and bounds
// tan(elevation) = z / rho over the box by its value at the centre +- (first-order variation along the three
// edges + a bound of the second-order remainder); the group is kept only if some channel (angular margin
// included) lies inside, and, for an azimuth shard, if the box reaches into the shard's sector.  Survivors are
// appended to a per-geometry list; k_project<CULLED> takes 64 / kCullGroup of them per wave, so its lanes are dense.
// On the headline frame 57 % of the groups of 4 fall between two rings (a patch of ground 40 m away subtends a
// fifth of the channel spacing); an 8-way azimuth shard keeps an eighth of the rest.
// The bound is evaluated in float with explicit slack (1e-5 relative + the channel tables' 0.005 degrees).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // unsigned order == float order
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// aabb6: lo.xyz as ordered bits (initialised to 0xFFFFFFFF), hi.xyz (initialised to 0)
__global__ __launch_bounds__(kBlock) void k_mesh_aabb(const uint8_t *__restrict__ verts, uint32_t stride, uint32_t nverts,
                                                      uint32_t *__restrict__ aabb6)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < nverts; j += gridDim.x * kBlock) {
        const float *p = reinterpret_cast<const float *>(verts + (size_t)j * stride);
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    }
    // six atomics per workgroup, at most 256 workgroups (a hot address sustains ~90 atomics per microsecond)
    __shared__ float s_red[6][kBlock / 64];
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_red[a][threadIdx.x >> 6] = lo[a]; s_red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = (int)threadIdx.x;
        const float l = fminf(fminf(s_red[a][0], s_red[a][1]), fminf(s_red[a][2], s_red[a][3]));
        const float h = fmaxf(fmaxf(s_red[3 + a][0], s_red[3 + a][1]), fmaxf(s_red[3 + a][2], s_red[3 + a][3]));
        if (l <= h) { atomicMin(&aabb6[a], ordered_bits(l)); atomicMax(&aabb6[3 + a], ordered_bits(h)); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// 30-bit Morton key of the triangle's centroid inside the mesh's bounding cube
__global__ __launch_bounds__(kBlock) void k_mesh_morton(const uint8_t *__restrict__ verts, uint32_t stride, const uint32_t *__restrict__ idx,
                                                        uint32_t ntris, const uint32_t *__restrict__ aabb6, uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= ntris) return;
    const float *a = reinterpret_cast<const float *>(verts + (size_t)idx[3 * (size_t)k + 0] * stride);
    const float *b = reinterpret_cast<const float *>(verts + (size_t)idx[3 * (size_t)k + 1] * stride);
    const float *c = reinterpret_cast<const float *>(verts + (size_t)idx[3 * (size_t)k + 2] * stride);
    uint32_t key = 0;
    // one scale for the three axes (the largest extent): a flat mesh must not spend a third of its key bits on the
    // noise of its thin direction
    float ext = 0.0f;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) ext = fmaxf(ext, from_ordered_bits(aabb6[3 + ax]) - from_ordered_bits(aabb6[ax]));
    const float scale = ext > 0.0f ? 1024.0f / ext : 0.0f;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = from_ordered_bits(aabb6[ax]);
        const float cen = (a[ax] + b[ax] + c[ax]) * (1.0f / 3.0f);
        const uint32_t q = (uint32_t)fminf(fmaxf((cen - lo) * scale, 0.0f), 1023.0f);
        key |= spread10(q) << ax;
    }
    keys[k] = key;
    vals[k] = k;
}

__global__ __launch_bounds__(kBlock) void k_permute_indices(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ perm,
                                                            uint32_t ntris, uint32_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= ntris) return;
    const uint32_t t = perm[k];
    out[3 * (size_t)k + 0] = idx[3 * (size_t)t + 0];
    out[3 * (size_t)k + 1] = idx[3 * (size_t)t + 1];
    out[3 * (size_t)k + 2] = idx[3 * (size_t)t + 2];
}

// A sheared box from running sums: lo / hi = the axis-aligned extent, s** = second moments about its centre (cx, cy, cz);
// returns the slopes of the least-squares plane z = cz + sx (x - cx) + sy (y - cy) through the vertices (any slopes
// give a valid bound: the thickness is measured against whatever plane is chosen)
__device__ __forceinline__ void plane_slopes(float sxx, float syy, float sxy, float sxz, float syz, float &sx, float &sy)
{
    const float det = sxx * syy - sxy * sxy;
    sx = sy = 0.f;
    if (det > 1e-6f * (sxx + syy) * (sxx + syy) && det > 0.f) {
        sx = (sxz * syy - syz * sxy) / det;
        sy = (syz * sxx - sxz * sxy) / det;
        if (!(fabsf(sx) <= 8.f) || !(fabsf(sy) <= 8.f)) sx = sy = 0.f;   // a wall: plain axis-aligned box
    }
}

__device__ __forceinline__ void store_sheared_box(float4 *__restrict__ out, const float *lo, const float *hi, float cx, float cy, float cz,
                                                  float sx, float sy, float rmin, float rmax)
{
    // half extents rounded outwards: the residuals carry a few ulps of |z| and of sx*dx
    const float zc = cz + 0.5f * (rmin + rmax);
    const float scale_ulps = 4e-7f * (fabsf(cz) + fabsf(lo[2]) + fabsf(hi[2]) + (fabsf(sx) * (hi[0] - lo[0]) + fabsf(sy) * (hi[1] - lo[1])));
    const float hx = 0.5f * (hi[0] - lo[0]) * 1.000001f + 1e-7f * (fabsf(lo[0]) + fabsf(hi[0]));
    const float hy = 0.5f * (hi[1] - lo[1]) * 1.000001f + 1e-7f * (fabsf(lo[1]) + fabsf(hi[1]));
    const float hz = 0.5f * (rmax - rmin) * 1.000001f + scale_ulps;
    out[0] = make_float4(cx, cy, zc, hx);
    out[1] = make_float4(hy, hz, sx, sy);
}

__device__ __forceinline__ bool box_extent_bad(const float *lo, const float *hi)
{
    return !(lo[0] <= hi[0]) || !(lo[1] <= hi[1]) || !(lo[2] <= hi[2]) || !(hi[0] - lo[0] < INFINITY) || !(hi[1] - lo[1] < INFINITY) ||
           !(hi[2] - lo[2] < INFINITY);
}

// one lane per sorted triangle: its three corners as uploaded, 16 bytes each, and in the spare word of the first the
// triangle's number in the caller's order -- what k_project<CULLED> reads for a surviving group (GeomSource::corners)
__global__ __launch_bounds__(kBlock) void k_corners(const uint8_t *__restrict__ verts, uint32_t stride, const uint32_t *__restrict__ idx_sorted,
                                                    const uint32_t *__restrict__ perm, uint32_t ntris, float4 *__restrict__ corners)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= ntris) return;
    const uint32_t a = idx_sorted[3 * (size_t)k + 0], b = idx_sorted[3 * (size_t)k + 1], c = idx_sorted[3 * (size_t)k + 2];
    const float *pa = reinterpret_cast<const float *>(verts + (size_t)a * stride);
    const float *pb = reinterpret_cast<const float *>(verts + (size_t)b * stride);
    const float *pc = reinterpret_cast<const float *>(verts + (size_t)c * stride);
    corners[3 * (size_t)k + 0] = make_float4(pa[0], pa[1], pa[2], __uint_as_float(perm[k]));
    corners[3 * (size_t)k + 1] = make_float4(pb[0], pb[1], pb[2], 0.0f);
    corners[3 * (size_t)k + 2] = make_float4(pc[0], pc[1], pc[2], 0.0f);
}

// one lane per group of kCullGroup sorted triangles: bounds[2g] = (cx, cy, cz, hx), bounds[2g+1] = (hy, hz, sx, sy); the 64
// lanes of a wave are the kCullBlockGroups groups of one block, whose bound (same form, over all its vertices, by wave
// reductions) goes to block_bounds[2b], [2b+1]
__global__ __launch_bounds__(kBlock) void k_group_bounds(const uint8_t *__restrict__ verts, uint32_t stride,
                                                         const uint32_t *__restrict__ idx_sorted, uint32_t ntris, float4 *__restrict__ bounds,
                                                         float4 *__restrict__ block_bounds)
{
    static_assert(kCullBlockGroups == 64, "a wave bounds one block");
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t first = g * kCullGroup;
    const bool live = first < ntris;
    const uint32_t n = live ? min(kCullGroup, ntris - first) * 3u : 0u;
    auto vertex = [&](uint32_t j) { return reinterpret_cast<const float *>(verts + (size_t)idx_sorted[3 * (size_t)first + j] * stride); };
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = 0; j < n; ++j) {
        const float *p = vertex(j);
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
    const bool bad = live && box_extent_bad(lo, hi);
    if (live && bad) {
        // NaN / infinite coordinates: a bound that k_cull never rejects
        bounds[2 * (size_t)g] = make_float4(0.f, 0.f, 0.f, INFINITY);
        bounds[2 * (size_t)g + 1] = make_float4(INFINITY, INFINITY, 0.f, 0.f);
    } else if (live) {
        const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
        float sxx = 0.f, syy = 0.f, sxy = 0.f, sxz = 0.f, syz = 0.f;
        for (uint32_t j = 0; j < n; ++j) {
            const float *p = vertex(j);
            const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
            sxx += dx * dx; syy += dy * dy; sxy += dx * dy; sxz += dx * dz; syz += dy * dz;
        }
        float sx, sy;
        plane_slopes(sxx, syy, sxy, sxz, syz, sx, sy);
        float rmin = INFINITY, rmax = -INFINITY;
        for (uint32_t j = 0; j < n; ++j) {
            const float *p = vertex(j);
            const float r = (p[2] - cz) - (sx * (p[0] - cx) + sy * (p[1] - cy));
            rmin = fminf(rmin, r); rmax = fmaxf(rmax, r);
        }
        store_sheared_box(bounds + 2 * (size_t)g, lo, hi, cx, cy, cz, sx, sy, rmin, rmax);
    }
    // ---- the block of this wave's 64 groups (lanes without a group contribute nothing)
    const bool any_live = __any(live);
    if (!any_live) return;
    const bool any_bad = __any(bad);
    float blo[3], bhi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        blo[a] = lo[a]; bhi[a] = hi[a];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { blo[a] = fminf(blo[a], __shfl_xor(blo[a], off)); bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], off)); }
    }
    const float cx = 0.5f * (blo[0] + bhi[0]), cy = 0.5f * (blo[1] + bhi[1]), cz = 0.5f * (blo[2] + bhi[2]);
    float sums[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (!any_bad)
        for (uint32_t j = 0; j < n; ++j) {
            const float *p = vertex(j);
            const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
            sums[0] += dx * dx; sums[1] += dy * dy; sums[2] += dx * dy; sums[3] += dx * dz; sums[4] += dy * dz;
        }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sums[a] += __shfl_xor(sums[a], off);
    }
    float sx, sy;
    plane_slopes(sums[0], sums[1], sums[2], sums[3], sums[4], sx, sy);
    float rmin = INFINITY, rmax = -INFINITY;
    if (!any_bad)
        for (uint32_t j = 0; j < n; ++j) {
            const float *p = vertex(j);
            const float r = (p[2] - cz) - (sx * (p[0] - cx) + sy * (p[1] - cy));
            rmin = fminf(rmin, r); rmax = fmaxf(rmax, r);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { rmin = fminf(rmin, __shfl_xor(rmin, off)); rmax = fmaxf(rmax, __shfl_xor(rmax, off)); }
    if ((threadIdx.x & 63u) == 0) {
        const size_t b = g / kCullBlockGroups;
        if (any_bad) {
            block_bounds[2 * b] = make_float4(0.f, 0.f, 0.f, INFINITY);
            block_bounds[2 * b + 1] = make_float4(INFINITY, INFINITY, 0.f, 0.f);
        } else {
            store_sheared_box(block_bounds + 2 * b, blo, bhi, cx, cy, cz, sx, sy, rmin, rmax);
        }
    }
}

// linear part L = Rinv * A and offset o = Rinv * (A.t - t) of the vertex transform p' = Rinv ((A v) - t), per geometry;
// nrm >= the largest factor by which L stretches a vector (its Frobenius norm, rounded up: sqrt(3) for a rotation).
// Uniform per geometry and only ever used in bounds: formed on the host (launch_project), once per launch.
struct LinearMap { float l[9], o[3], nrm; };

// what k_cull needs of the geometries of a launch (kernel argument)
struct CullGeom {
    const float4 *boxes;   // group bounds, then block bounds (launch_group_bounds)
    uint32_t ntris;
    LinearMap m;
};
struct CullBatch {
    uint32_t n;
    uint32_t cull_first[kGeomsPerLaunch + 1];   // first workgroup of geometry i; [n] = the grid
    uint32_t list_first[kGeomsPerLaunch];       // as GeomBatch
    uint32_t seg_cap[kGeomsPerLaunch];
    uint32_t rounds;                            // groups per workgroup / 256
    CullGeom g[kGeomsPerLaunch];
};

// k_cull's channel query: is there a channel i with tan_up[i] >= tan_lo and tan_dn[i] <= tan_hi (both tables ascending)?
// SEARCH: binary search (any sensor).  LUT: ProjectParams::chan_lut -- one table look-up started one bucket early (the
// float bucket number is good to 1e-3 buckets), two compare-and-steps, one compare; tan_up carries two +inf sentinels
// behind its V entries.  Straight-line code: the four tests of a k_cull lane interleave instead of queueing behind each
// other's LDS round trips.
struct ChanQuery {
    const float *tan_up, *tan_dn;   // LUT: [V + 2] each, +inf padded
    const uint16_t *lut;
    float t0, scale;
    uint32_t V;
};

template <bool LUT>
__device__ __forceinline__ bool chan_query(const ChanQuery &cq, float tan_lo, float tan_hi)
{
    if (LUT) {
        const float x = fminf(fmaxf((tan_lo - cq.t0) * cq.scale - 0.5f, 0.0f), (float)(kCullLutBuckets - 1u));   // NaN -> 0
        uint32_t i = cq.lut[(uint32_t)x];
        i += cq.tan_up[i] < tan_lo ? 1u : 0u;
        i += cq.tan_up[i] < tan_lo ? 1u : 0u;
        return (i < cq.V) & (cq.tan_dn[i] <= tan_hi);
    }
    uint32_t lo_i = 0, hi_i = cq.V;
    while (lo_i < hi_i) { const uint32_t mid = (lo_i + hi_i) >> 1; if (cq.tan_up[mid] < tan_lo) lo_i = mid + 1; else hi_i = mid; }
    return lo_i < cq.V && cq.tan_dn[lo_i] <= tan_hi;
}

// Can any ray of this handle's raster meet the sheared box  { c + a e1 + b e2 + g e3 : |a|, |b|, |g| <= 1 },  e1 = (hx, 0, sx hx),
// e2 = (0, hy, sy hy), e3 = (0, 0, hz)  (mesh space; b0 = (c, hx), b1 = (hy, hz, sx, sy))?  With p' = L p + o the sensor-
// frame point and f = z' / rho' the tangent of its elevation:
//     f(c + d) = f(c) + grad f . (L d) + R,   grad f = (-f x'/rho^2, -f y'/rho^2, 1/rho) at c',
// and  grad f . (L d) = (L^T grad f) . d = g . d  -- the gradient carried back into MESH space, where the edges are
// two-term vectors: the first-order half width is  hx |g.x + sx g.z| + hy |g.y + sy g.z| + hz |g.z|,  a dozen
// operations instead of three transformed edge vectors and their radial / tangential projections.  Second order:
// |R| <= (ez eps + (|z'| + ez) 2 eps^2) / rho  for eps = (er + et) / rho < 1/4  with the box's radial + tangential and
// vertical half extents.  ez = sum |row 3 of L . e_i| is cheap and kept exact (a flat patch of ground has next to none);
// er + et <= sqrt 2 rad with the box's radius  rad = nrm (hx (1 + |sx|) + hy (1 + |sy|) + hz),  nrm = the largest
// singular value of L (1 for a rigid pose; from the host), and the bound is monotone in it.  The same radius bounds the
// reach of the box seen from above (azimuth-sector test of a shard).  Everything is a bound with slack: approximate rsq / products are fine.
// the azimuth-sector half of the test below, alone (k_cull's workgroup-level early out): seen from above the box lies inside
// the disc of radius rad around its centre; it is outside a boundary plane of the shard's sector if the centre is further
// out than that.  Needs no channel table.
__device__ __forceinline__ bool box_out_of_sector(const ProjectParams &pp, const LinearMap &m, float4 b0, float4 b1)
{
    const float hx = b0.w, hy = b1.x, hz = b1.y, sx = b1.z, sy = b1.w;
    if (!(hx < INFINITY)) return false;   // unbounded (NaN / infinite coordinates): never rejected
    const float cx = fmaf(m.l[0], b0.x, fmaf(m.l[1], b0.y, fmaf(m.l[2], b0.z, m.o[0])));
    const float cy = fmaf(m.l[3], b0.x, fmaf(m.l[4], b0.y, fmaf(m.l[5], b0.z, m.o[1])));
    const float rho2 = fmaf(cx, cx, cy * cy);
    if (!(rho2 > 1e-12f) || !(rho2 < INFINITY)) return false;   // on the axis
    const float rad = m.nrm * fmaf(hx, 1.0f + fabsf(sx), fmaf(hy, 1.0f + fabsf(sy), hz)) * 1.00001f;
    const float reach = fmaf(rad, 1.0001f, 1e-6f * rho2 * __builtin_amdgcn_rsqf(rho2));
    return (pp.sec_a[0] * cy - pp.sec_a[1] * cx < -reach) | (cx * pp.sec_b[1] - cy * pp.sec_b[0] < -reach);
}

template <bool LUT>
__device__ __forceinline__ bool group_meets_raster(const ProjectParams &pp, const ChanQuery &cq, const LinearMap &m, float4 b0, float4 b1)
{
    const float hx = b0.w, hy = b1.x, hz = b1.y, sx = b1.z, sy = b1.w;
    // (no early returns: every special case is a flag, folded in at the end in the order the cases apply)
    const bool unbounded = !(hx < INFINITY);
    const float cx = fmaf(m.l[0], b0.x, fmaf(m.l[1], b0.y, fmaf(m.l[2], b0.z, m.o[0])));
    const float cy = fmaf(m.l[3], b0.x, fmaf(m.l[4], b0.y, fmaf(m.l[5], b0.z, m.o[1])));
    const float cz = fmaf(m.l[6], b0.x, fmaf(m.l[7], b0.y, fmaf(m.l[8], b0.z, m.o[2])));
    const float rho2 = fmaf(cx, cx, cy * cy);
    const bool on_axis = !(rho2 > 1e-12f) || !(rho2 < INFINITY);
    const float inv_rho = __builtin_amdgcn_rsqf(rho2), inv_rho2 = inv_rho * inv_rho;
    const float fc = cz * inv_rho, k = fc * inv_rho2;
    // g = L^T grad f,  grad f = (-k cx, -k cy, inv_rho)
    const float ax = -k * cx, ay = -k * cy;
    const float gx = fmaf(m.l[0], ax, fmaf(m.l[3], ay, m.l[6] * inv_rho));
    const float gy = fmaf(m.l[1], ax, fmaf(m.l[4], ay, m.l[7] * inv_rho));
    const float gz = fmaf(m.l[2], ax, fmaf(m.l[5], ay, m.l[8] * inv_rho));
    const float w = fmaf(hx, fabsf(fmaf(sx, gz, gx)), fmaf(hy, fabsf(fmaf(sy, gz, gy)), hz * fabsf(gz)));
    // vertical half extent of the box in the sensor frame (exact to rounding: a flat patch of ground has next to none,
    // and it multiplies the larger of the two second-order terms), and its radius
    const float ez = fmaf(hx, fabsf(fmaf(sx, m.l[8], m.l[6])), fmaf(hy, fabsf(fmaf(sy, m.l[8], m.l[7])), hz * fabsf(m.l[8]))) * 1.00001f;
    const float rad = m.nrm * fmaf(hx, 1.0f + fabsf(sx), fmaf(hy, 1.0f + fabsf(sy), hz)) * 1.00001f;
    bool out_of_sector = false;
    if (pp.sector_on) {   // uniform
        // seen from above the box lies inside the disc of radius rad around its centre: outside a boundary plane of the
        // shard's sector if the centre is further out than that
        const float reach = fmaf(rad, 1.0001f, 1e-6f * rho2 * inv_rho);
        out_of_sector = (pp.sec_a[0] * cy - pp.sec_a[1] * cx < -reach) | (cx * pp.sec_b[1] - cy * pp.sec_b[0] < -reach);
    }
    const float eps = 1.4142137f * rad * inv_rho;
    const bool near_axis = !(eps < 0.25f);   // next to the vertical axis: no first-order bound
    const float rem = fmaf(ez, eps, (fabsf(cz) + ez) * 2.0f * eps * eps) * inv_rho;
    const float half = fmaf(w + rem, 1.0001f, 3e-6f * (fabsf(fc) + 1e-3f));   // rounding of the two dozen products above
    const bool some_channel = chan_query<LUT>(cq, fc - half, fc + half);
    return unbounded | on_axis | (!out_of_sector & (near_axis | some_channel));
}

// `rounds` x 256 groups per workgroup (one lane per group and round; batch.rounds, 2 .. kCullMaxRounds, is chosen by the
// host so that the whole pass is ONE round of resident workgroups where it can: at 1 024 groups per workgroup SYN-10M
// took 2 441 workgroups for 2 048 places -- two rounds of an 8 us workgroup); a workgroup belongs to one geometry
// (batch.cull_first).  Round `it` of wave w is block  first/64 + 4 it + w  of the mesh -- 64 consecutive groups under one
// coarse bound: every wave tests the workgroup's 4 x rounds block bounds first (one per lane) and skips, loads included,
// the rounds whose block no ring can meet (on SYN-10M 39 % of the blocks; a lone group survives 23 % of the time).
// Survivors are collected in LDS and appended to one of the geometry's kCullSegs list segments -- workgroup j to
// segment j % kCullSegs -- with ONE global atomic per workgroup on that segment's counter.
constexpr uint32_t kCullMaxRounds = 8;
constexpr uint32_t kCullMaxPerBlock = kCullMaxRounds * kBlock;

template <bool LDS_TABLES, bool COUNT, bool LUT /* needs LDS_TABLES */, bool SECTOR /* an azimuth shard: pp.sector_on */>
__global__ __launch_bounds__(kBlock) void k_cull(ProjectParams pp, CullBatch batch, uint32_t *__restrict__ list, uint32_t *__restrict__ counts,
                                                 unsigned long long *__restrict__ stats)
{
    extern __shared__ float s_tan[];   // LDS_TABLES: tan_up[V + 2], tan_dn[V + 2] (two +inf sentinels each), LUT: then chan_lut
    __shared__ uint32_t s_keep[kCullMaxPerBlock];
    __shared__ uint32_t s_n, s_base;
    static_assert((kBlock / 64) * kCullBlockGroups == kBlock, "a round of a wave is one block");
    const uint32_t rounds = batch.rounds;
    uint32_t gi = 0;
    while (gi + 1u < batch.n && blockIdx.x >= batch.cull_first[gi + 1u]) ++gi;
    const CullGeom &src = batch.g[gi];
    const uint32_t wg = blockIdx.x - batch.cull_first[gi];   // this workgroup among the geometry's
    const uint32_t g0 = wg * rounds * kBlock;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t n_groups = (src.ntris + kCullGroup - 1u) / kCullGroup;
    const uint32_t n_blocks = (n_groups + kCullBlockGroups - 1u) / kCullBlockGroups;
    // The coarse bounds go out first, before the channel tables are staged (the two waits overlap) -- ALL of the workgroup's
    // 4 x rounds block bounds in every wave (lane j holds block g0 / 64 + j = round j / 4 of wave j % 4; a kilobyte out of the
    // L2), so that every wave comes to the same answer without a word of LDS: under an azimuth shard a workgroup none of whose
    // blocks reaches into the sector ends HERE -- before the staging, the barrier and the ring tests.  Seven in eight do for an
    // eighth of a turn, and at ten million triangles the pass is 1 953 workgroups that would each have filled a slot for a
    // table staging and a barrier's worth of time.
    static_assert(kCullMaxRounds * (kBlock / 64) <= 32, "the workgroup's block bounds fit the lanes of one wave, their ballot a word");
    const float4 *__restrict__ block_boxes = src.boxes + 2 * (size_t)n_groups;
    // (the full turn has no such early out: there a wave looks at its own blocks only -- lane `it` holds round `it`'s)
    const uint32_t my_block = SECTOR ? g0 / kCullBlockGroups + lane : g0 / kCullBlockGroups + lane * (kBlock / 64) + w;
    const bool has_block = (SECTOR ? lane < rounds * (kBlock / 64) : lane < rounds) && my_block < n_blocks;
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
    if (has_block) { c0 = block_boxes[2 * (size_t)my_block]; c1 = block_boxes[2 * (size_t)my_block + 1]; }
    const LinearMap &m = src.m;
    if (SECTOR && !__any(has_block && !box_out_of_sector(pp, m, c0, c1))) return;   // (the same in all four waves: no barrier is left waiting)
    const uint32_t V = pp.tb.V;
    ChanQuery cq = {pp.chan_tan_up, pp.chan_tan_dn, pp.chan_lut, pp.lut_t0, pp.lut_scale, V};
    if (threadIdx.x == 0) s_n = 0;
    if (LDS_TABLES) {
        float *s_up = s_tan, *s_dn = s_tan + V + 2u;
        for (uint32_t i = threadIdx.x; i < V + 2u; i += kBlock) {
            s_up[i] = i < V ? pp.chan_tan_up[i] : INFINITY;
            s_dn[i] = i < V ? pp.chan_tan_dn[i] : INFINITY;
        }
        cq.tan_up = s_up;
        cq.tan_dn = s_dn;
        if (LUT) {
            uint32_t *s_lut = reinterpret_cast<uint32_t *>(s_dn + V + 2u);
            const uint32_t *g_lut = reinterpret_cast<const uint32_t *>(pp.chan_lut);
            for (uint32_t i = threadIdx.x; i < kCullLutBuckets / 2u; i += kBlock) s_lut[i] = g_lut[i];
            cq.lut = reinterpret_cast<const uint16_t *>(s_lut);
        }
    }
    __syncthreads();
    const bool block_alive = group_meets_raster<LUT>(pp, cq, m, c0, c1);
    const uint32_t alive_all = (uint32_t)__ballot(has_block && block_alive);
    uint32_t alive = SECTOR ? 0u : alive_all & ((1u << rounds) - 1u);   // bit `it`: this wave's block of round `it`
    if (SECTOR)
        for (uint32_t it = 0; it < rounds; ++it) alive |= ((alive_all >> (it * (kBlock / 64) + w)) & 1u) << it;
    // the rounds whose block is alive, two at a time: both rounds' bound loads go out first, then the two tests -- straight-
    // line code -- interleave (an odd round out is tested on its own: its partner is a predicated-off copy of itself)
    if (COUNT && lane == 0 && alive) atomicAdd(&stats[2], (unsigned long long)(__popc(alive) * kCullBlockGroups));   // counts[3]: group bounds read
    auto append = [&](uint32_t it, bool keep) {
        const unsigned long long mask = __ballot(keep);
        if (mask) {   // uniform over the wave
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&s_n, (uint32_t)__popcll(mask));
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (keep) s_keep[at + lanes_below(mask)] = g0 + it * kBlock + threadIdx.x;
        }
    };
    for (uint32_t todo = alive; todo;) {   // uniform over the wave
        const uint32_t ia = (uint32_t)__builtin_ctz(todo);
        todo &= todo - 1u;
        const bool pair = todo != 0u;
        const uint32_t ib = pair ? (uint32_t)__builtin_ctz(todo) : ia;
        todo &= todo - 1u;   // (0 stays 0)
        const uint32_t ga = g0 + ia * kBlock + threadIdx.x, gb = g0 + ib * kBlock + threadIdx.x;
        const uint32_t la = min(ga, n_groups - 1u), lb = min(gb, n_groups - 1u);
        const float4 a0 = src.boxes[2 * (size_t)la], a1 = src.boxes[2 * (size_t)la + 1];
        const float4 b0 = src.boxes[2 * (size_t)lb], b1 = src.boxes[2 * (size_t)lb + 1];
        const bool ka = group_meets_raster<LUT>(pp, cq, m, a0, a1), kb = group_meets_raster<LUT>(pp, cq, m, b0, b1);
        append(ia, ka && ga < n_groups);
        if (pair) append(ib, kb && gb < n_groups);
    }
    // (one global atomic per WORKGROUP: one per wave -- four times as many on the same 32 counters -- measured 3 us
    // slower on SYN-10M, the counters' ~11 ns per atomic showing again)
    __syncthreads();
    const uint32_t n = s_n;
    if (!n) return;
    const uint32_t seg = wg % kCullSegs;
    if (threadIdx.x == 0) s_base = atomicAdd(&counts[(gi * kCullSegs + seg) * 16u], n);
    __syncthreads();
    uint32_t *out = list + batch.list_first[gi] + seg * batch.seg_cap[gi] + s_base;
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) out[i] = s_keep[i];
}

// Finish pass, one thread per ray: (1) triangles whose footprint was too large for one wave were
// queued by k_project; here every ray gathers from that (normally empty) queue itself -- no atomics,
// no dependency between blocks -- and folds the hits into its own key.  A workgroup first culls the
// queue against the rectangle of its 256 rays on the (elevation rank, column) raster, 1024 entries at
// a time, so a ray only walks the footprints that reach its workgroup; (2) hits per 256-ray block,
// which the ordered pack needs.
constexpr uint32_t kCullChunk = 1024;

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_project_finish(ProjectParams pp, unsigned long long *__restrict__ best,
                                                           const BigItem *__restrict__ big, uint32_t big_capacity,
                                                           const uint32_t *__restrict__ big_count,
                                                           uint32_t *__restrict__ block_counts,
                                                           unsigned long long *__restrict__ stats)
{
    __shared__ uint32_t s_cnt[kBlock / 64];
    __shared__ uint32_t s_box[4][kBlock / 64];   // rank min / max, column min / max per wave
    __shared__ uint16_t s_list[kCullChunk];
    __shared__ uint32_t s_n;
    const uint32_t n = pp.tb.V * pp.tb.naz;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t n_big = min(*big_count, big_capacity);
    bool hit = false;
    unsigned long long key = ~0ull;
    if (q < n) key = best[q];
    if (n_big) {   // uniform
        uint32_t rank = 0, h = 0;
        V3 d = {0.f, 0.f, 0.f};
        uint32_t rmin = 0xFFFFFFFFu, rmax = 0, cmin = 0xFFFFFFFFu, cmax = 0;
        if (q < n) {
            const uint32_t v = q / pp.tb.naz;
            h = pp.tb.az0 + (q - v * pp.tb.naz);
            rank = pp.chan_rank[v];  // position of channel v in the elevation-sorted table
            const float st = pp.tb.sin_theta[v];
            const float2 cs = pp.tb.cs_phi[h];
            d = {st * cs.x, st * cs.y, pp.tb.cos_theta[v]};
            rmin = rmax = rank;
            cmin = cmax = h;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            rmin = min(rmin, (uint32_t)__shfl_xor(rmin, off)); rmax = max(rmax, (uint32_t)__shfl_xor(rmax, off));
            cmin = min(cmin, (uint32_t)__shfl_xor(cmin, off)); cmax = max(cmax, (uint32_t)__shfl_xor(cmax, off));
        }
        if (lane == 0) { s_box[0][w] = rmin; s_box[1][w] = rmax; s_box[2][w] = cmin; s_box[3][w] = cmax; }
        __syncthreads();
        rmin = min(min(s_box[0][0], s_box[0][1]), min(s_box[0][2], s_box[0][3]));
        rmax = max(max(s_box[1][0], s_box[1][1]), max(s_box[1][2], s_box[1][3]));
        cmin = min(min(s_box[2][0], s_box[2][1]), min(s_box[2][2], s_box[2][3]));
        cmax = max(max(s_box[3][0], s_box[3][1]), max(s_box[3][2], s_box[3][3]));
        uint32_t ntest = 0;
        for (uint32_t base = 0; base < n_big; base += kCullChunk) {
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
            const uint32_t m = min(kCullChunk, n_big - base);
            for (uint32_t j = threadIdx.x; j < m; j += kBlock) {
                const BigItem &it = big[base + j];
                // [i0, i0+nch) x ([h0a, h0a+na) u [h0b, h0b+nb)) against [rmin, rmax] x [cmin, cmax]
                const bool rows = it.i0 <= rmax && it.i0 + it.nch > rmin;
                const bool cols = (it.na && it.h0a <= cmax && it.h0a + it.na > cmin) || (it.nb && it.h0b <= cmax && it.h0b + it.nb > cmin);
                if (rows && cols) s_list[atomicAdd(&s_n, 1u)] = (uint16_t)j;
            }
            __syncthreads();
            const uint32_t cnt = s_n;
            if (q < n) {
                for (uint32_t k = 0; k < cnt; ++k) {
                    const BigItem &it = big[base + s_list[k]];
                    if (rank - it.i0 >= it.nch) continue;
                    if (h - it.h0a >= it.na && h - it.h0b >= it.nb) continue;
                    ++ntest;
                    float t;
                    if (tri_test(d, {it.v0[0], it.v0[1], it.v0[2]}, {it.e1[0], it.e1[1], it.e1[2]}, {it.e2[0], it.e2[1], it.e2[2]},
                                 it.NgC, t)) {
                        const unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | it.gid;
                        key = k2 < key ? k2 : key;
                    }
                }
            }
            __syncthreads();
        }
        if (q < n) best[q] = key;
        if (COUNT && ntest) atomicAdd(&stats[0], (unsigned long long)ntest);
    }
    hit = q < n && key != ~0ull;
    const unsigned long long mask = __ballot(hit);
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// The finish pass with RPT rays per lane (thread t takes rays t, t + 256, ... of its workgroup's RPT x 256), for frames that
// overlap on three streams.  There the device is short of WAVE SLOTS -- six and nine frames in flight are no faster than three
// (tools/two_handles_probe.py) -- and what a frame costs is the sum over its waves of how long each holds its slot.  A wave of
// this pass holds it for one memory round trip whatever it carries: a quarter of the waves with four key loads in flight each
// 15.2 -> 14.8 us per frame; k_pack_wide (ls_kernels.hip) the same way 14.8 -> 13.4.  A frame ALONE is slower like this
// (24.3 -> 25.4 us: fewer waves to hide the latency behind), so ls_trace.cpp asks for it in three-stream mode only, and only
// while a workgroup per CU is left (a shard's 256 ray blocks stay as they are).
template <uint32_t RPT>
__global__ __launch_bounds__(kBlock) void k_project_finish_wide(ProjectParams pp, unsigned long long *__restrict__ best,
                                                                const BigItem *__restrict__ big, uint32_t big_capacity,
                                                                const uint32_t *__restrict__ big_count, uint32_t *__restrict__ block_counts)
{
    __shared__ uint32_t s_cnt[RPT][kBlock / 64];
    __shared__ uint32_t s_box[4][kBlock / 64];
    __shared__ uint16_t s_list[kCullChunk];
    __shared__ uint32_t s_n;
    const SensorTables &tb = pp.tb;
    const uint32_t n = tb.V * tb.naz, n_blocks = (n + kBlock - 1u) / kBlock;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t q0 = blockIdx.x * RPT * kBlock + threadIdx.x;
    const uint32_t n_big = min(*big_count, big_capacity);
    unsigned long long key[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t q = q0 + j * kBlock;
        key[j] = q < n ? best[q] : ~0ull;
    }
    if (n_big) {   // uniform
        for (uint32_t j = 0; j < RPT; ++j) {
            const uint32_t q = q0 + j * kBlock;
            uint32_t rank = 0, h = 0, rmin = 0xFFFFFFFFu, rmax = 0, cmin = 0xFFFFFFFFu, cmax = 0;
            V3 d = {0.f, 0.f, 0.f};
            if (q < n) {
                const uint32_t v = q / tb.naz;
                h = tb.az0 + (q - v * tb.naz);
                rank = pp.chan_rank[v];
                const float st = tb.sin_theta[v];
                const float2 cs = tb.cs_phi[h];
                d = {st * cs.x, st * cs.y, tb.cos_theta[v]};
                rmin = rmax = rank;
                cmin = cmax = h;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                rmin = min(rmin, (uint32_t)__shfl_xor(rmin, off)); rmax = max(rmax, (uint32_t)__shfl_xor(rmax, off));
                cmin = min(cmin, (uint32_t)__shfl_xor(cmin, off)); cmax = max(cmax, (uint32_t)__shfl_xor(cmax, off));
            }
            __syncthreads();
            if (lane == 0) { s_box[0][w] = rmin; s_box[1][w] = rmax; s_box[2][w] = cmin; s_box[3][w] = cmax; }
            __syncthreads();
            rmin = min(min(s_box[0][0], s_box[0][1]), min(s_box[0][2], s_box[0][3]));
            rmax = max(max(s_box[1][0], s_box[1][1]), max(s_box[1][2], s_box[1][3]));
            cmin = min(min(s_box[2][0], s_box[2][1]), min(s_box[2][2], s_box[2][3]));
            cmax = max(max(s_box[3][0], s_box[3][1]), max(s_box[3][2], s_box[3][3]));
            for (uint32_t base = 0; base < n_big; base += kCullChunk) {
                if (threadIdx.x == 0) s_n = 0;
                __syncthreads();
                const uint32_t m = min(kCullChunk, n_big - base);
                for (uint32_t k = threadIdx.x; k < m; k += kBlock) {
                    const BigItem &it = big[base + k];
                    const bool rows = it.i0 <= rmax && it.i0 + it.nch > rmin;
                    const bool cols = (it.na && it.h0a <= cmax && it.h0a + it.na > cmin) || (it.nb && it.h0b <= cmax && it.h0b + it.nb > cmin);
                    if (rows && cols) s_list[atomicAdd(&s_n, 1u)] = (uint16_t)k;
                }
                __syncthreads();
                const uint32_t cnt = s_n;
                if (q < n) {
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const BigItem &it = big[base + s_list[k]];
                        if (rank - it.i0 >= it.nch) continue;
                        if (h - it.h0a >= it.na && h - it.h0b >= it.nb) continue;
                        float t;
                        if (tri_test(d, {it.v0[0], it.v0[1], it.v0[2]}, {it.e1[0], it.e1[1], it.e1[2]}, {it.e2[0], it.e2[1], it.e2[2]},
                                     it.NgC, t)) {
                            const unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | it.gid;
                            key[j] = k2 < key[j] ? k2 : key[j];
                        }
                    }
                }
                __syncthreads();
            }
            if (q < n) best[q] = key[j];
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const unsigned long long m = __ballot(key[j] != ~0ull);
        if (lane == 0) s_cnt[j][w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < RPT) {
        const uint32_t b = blockIdx.x * RPT + threadIdx.x;
        if (b < n_blocks) block_counts[b] = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
    }
}

// ------------------------------------------------------------------------------------------
// LS_OPT_PIPELINE: finish + pack of one frame as a single set of workgroups that can ride in the same
// launch as the next frame's k_project (k_frame) -- one launch per frame instead of three, and the
// tail of the triangle streaming is filled with the previous frame's per-ray work.
// Per workgroup of 256 rays: the queue gather of k_project_finish, then the ordered pack of
// k_pack<true>; hits before this workgroup = sum of the counts that the workgroups before it publish,
// each tagged with the frame's epoch (no reset between frames).  A workgroup only ever waits for
// lower-numbered ones of its own set, which the dispatcher started earlier (the chained-scan premise).
// ------------------------------------------------------------------------------------------
template <bool COUNT>
__device__ __forceinline__ void finish_pack_body(const ProjectParams &pp, const FinishPackArgs &fa, uint32_t block_idx,
                                                 uint32_t *scratch /* LDS, >= 560 words */,
                                                 unsigned long long *__restrict__ stats)
{
    uint32_t *s_cnt = scratch;                                   // [4]
    uint32_t *s_part = scratch + 4;                              // [4]
    uint32_t(*s_box)[kBlock / 64] = reinterpret_cast<uint32_t(*)[kBlock / 64]>(scratch + 8);   // [4][4]
    uint32_t *s_n = scratch + 24;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(scratch + 32);   // [kCullChunk]
    const SensorTables &tb = pp.tb;
    unsigned long long *__restrict__ best = fa.best;
    const BigItem *__restrict__ big = static_cast<const BigItem *>(fa.big);
    const uint32_t n = tb.V * tb.naz;
    const uint32_t q = block_idx * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t n_big = min(*fa.big_count, fa.big_capacity);
    const uint32_t epoch = fa.epoch_word ? *fa.epoch_word : fa.epoch;
    const uint32_t publish_epoch = fa.epoch_word ? epoch : fa.publish_epoch;
    unsigned long long key = ~0ull;
    uint32_t v = 0, h = 0, rank = 0;
    V3 d = {0.f, 0.f, 0.f};
    if (q < n) {
        key = best[q];
        best[q] = ~0ull;   // re-armed for the frame after the next
        v = q / tb.naz;
        h = tb.az0 + (q - v * tb.naz);
        // LidarDevice.cpp:310-316: d = (sin(theta)cos(phi), sin(theta)sin(phi), cos(theta))
        const float st = tb.sin_theta[v];
        const float2 cs = tb.cs_phi[h];
        d = {st * cs.x, st * cs.y, tb.cos_theta[v]};
    }
    if (n_big) {   // uniform
        uint32_t rmin = 0xFFFFFFFFu, rmax = 0, cmin = 0xFFFFFFFFu, cmax = 0;
        if (q < n) {
            rank = pp.chan_rank[v];  // position of channel v in the elevation-sorted table
            rmin = rmax = rank;
            cmin = cmax = h;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            rmin = min(rmin, (uint32_t)__shfl_xor(rmin, off)); rmax = max(rmax, (uint32_t)__shfl_xor(rmax, off));
            cmin = min(cmin, (uint32_t)__shfl_xor(cmin, off)); cmax = max(cmax, (uint32_t)__shfl_xor(cmax, off));
        }
        if (lane == 0) { s_box[0][w] = rmin; s_box[1][w] = rmax; s_box[2][w] = cmin; s_box[3][w] = cmax; }
        __syncthreads();
        rmin = min(min(s_box[0][0], s_box[0][1]), min(s_box[0][2], s_box[0][3]));
        rmax = max(max(s_box[1][0], s_box[1][1]), max(s_box[1][2], s_box[1][3]));
        cmin = min(min(s_box[2][0], s_box[2][1]), min(s_box[2][2], s_box[2][3]));
        cmax = max(max(s_box[3][0], s_box[3][1]), max(s_box[3][2], s_box[3][3]));
        uint32_t ntest = 0;
        for (uint32_t base = 0; base < n_big; base += kCullChunk) {
            if (threadIdx.x == 0) *s_n = 0;
            __syncthreads();
            const uint32_t m = min(kCullChunk, n_big - base);
            for (uint32_t j = threadIdx.x; j < m; j += kBlock) {
                const BigItem &it = big[base + j];
                const bool rows = it.i0 <= rmax && it.i0 + it.nch > rmin;
                const bool cols = (it.na && it.h0a <= cmax && it.h0a + it.na > cmin) || (it.nb && it.h0b <= cmax && it.h0b + it.nb > cmin);
                if (rows && cols) s_list[atomicAdd(s_n, 1u)] = (uint16_t)j;
            }
            __syncthreads();
            const uint32_t cnt = *s_n;
            if (q < n) {
                for (uint32_t k = 0; k < cnt; ++k) {
                    const BigItem &it = big[base + s_list[k]];
                    if (rank - it.i0 >= it.nch) continue;
                    if (h - it.h0a >= it.na && h - it.h0b >= it.nb) continue;
                    ++ntest;
                    float t;
                    if (tri_test(d, {it.v0[0], it.v0[1], it.v0[2]}, {it.e1[0], it.e1[1], it.e1[2]}, {it.e2[0], it.e2[1], it.e2[2]},
                                 it.NgC, t)) {
                        const unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | it.gid;
                        key = k2 < key ? k2 : key;
                    }
                }
            }
            __syncthreads();
        }
        if (COUNT && ntest) atomicAdd(&stats[0], (unsigned long long)ntest);
    }
    const bool hit = key != ~0ull;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t mine = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (threadIdx.x == 0)
        __hip_atomic_store(&fa.status[block_idx], ((unsigned long long)publish_epoch << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // hits of all workgroups before this one.  Waiting is rare (every workgroup publishes within a few
    // microseconds of its neighbours unless the queue gather is heavy) and must stay cheap for the ones
    // still working: the polls back off quickly (s_sleep 1, 4, 16, 64, 127, 127, ... x 64 cycles)
    uint32_t acc = 0;
    bool stuck = false;
    for (uint32_t i = threadIdx.x; i < block_idx; i += kBlock) {
        unsigned long long st = __hip_atomic_load(&fa.status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t spin = 0; (uint32_t)(st >> 32) != epoch; ++spin) {
            if (spin > fa.spin_limit) { stuck = true; break; }   // ~1 s; never seen; keeps a broken premise from hanging the GPU
            if (spin == 0) __builtin_amdgcn_s_sleep(1);
            else if (spin == 1) __builtin_amdgcn_s_sleep(4);
            else if (spin == 2) __builtin_amdgcn_s_sleep(16);
            else if (spin == 3) __builtin_amdgcn_s_sleep(64);
            else __builtin_amdgcn_s_sleep(127);
            st = __hip_atomic_load(&fa.status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc += (uint32_t)st;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    const bool any_stuck = __any(stuck);
    if (lane == 0) s_part[w] = any_stuck ? 0xFFFFFFFFu : acc;
    __syncthreads();
    const bool bad = s_part[0] == 0xFFFFFFFFu || s_part[1] == 0xFFFFFFFFu || s_part[2] == 0xFFFFFFFFu || s_part[3] == 0xFFFFFFFFu;
    uint32_t base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (block_idx == fa.n_blocks - 1u) {
        if (threadIdx.x == 0) *fa.n_points = bad ? 0u : base + mine;
        // Every workgroup before this one has published, so each of them has read the queue length (and the tag) and is done
        // with the queue: the queue / survivor counters are re-armed HERE -- they may be the very ones this frame used (three-
        // stream mode: a slot's next frame follows in stream order), or those of the frame after the next (rider mode) -- and
        // a tag in device memory steps on (0 is what fresh status words carry: skipped).
        if (threadIdx.x == 0) fa.rearm_big_count[0] = 0u;
        uint32_t fullest = 0;   // the fullest survivor segment, read before the counters go (ProgressArgs::cull_hint)
        for (uint32_t i = threadIdx.x; i < kCullCounters; i += kBlock) {
            fullest = max(fullest, fa.rearm_big_count[kCullCountAt + i * 16u]);
            fa.rearm_big_count[kCullCountAt + i * 16u] = 0u;
        }
        if (fa.cull_hint && fa.rearm_big_count == fa.big_count) {   // (uniform; the counters re-armed are the ones this frame used)
            uint32_t *s_hint = scratch + 28;                          // (four spare words between s_n and s_list)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, off));
            if (lane == 0) s_hint[w] = fullest;
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(fa.cull_hint, 1u + max(max(s_hint[0], s_hint[1]), max(s_hint[2], s_hint[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (fa.epoch_word && threadIdx.x == 0) *fa.epoch_word = epoch + 1u ? epoch + 1u : 1u;
    }
    if (bad) {
        // the frame is lost: tell the host (checked at its next wait: ls_trace_scene, ls_tracer_synchronize)
        if (threadIdx.x == 0) __hip_atomic_fetch_or(fa.device_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (uint32_t k = 0; k < w; ++k) base += s_cnt[k];
    if (!hit) return;
    const uint32_t dst = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const uint32_t gid = (uint32_t)key;
    const float t = __uint_as_float((uint32_t)(key >> 32));
    float4 *__restrict__ points = reinterpret_cast<float4 *>(fa.points32);
    uint4 *__restrict__ hits = reinterpret_cast<uint4 *>(fa.hits);
    // EmbreeTracer.cpp:341-345: xyz = tfar*dir, intensity 64.0; ring = channel (LidarDeviceKernels.cu:51)
    if (fa.compact == 3u) {
        // LS_OPT_EMIT_POINTS = 0: hit records only
    } else if (fa.compact) {
        points[dst] = make_float4(t * d.x, t * d.y, t * d.z, __int_as_float((int)v));
    } else {
        points[2 * (size_t)dst] = make_float4(t * d.x, t * d.y, t * d.z, 0.0f);
        points[2 * (size_t)dst + 1] = make_float4(64.0f, __int_as_float((int)v), 0.0f, 0.0f);
    }
    // (geomID, primID) from the global triangle id: last geometry slot whose first id <= gid
    uint32_t lo = 0, hi = fa.gt.n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (fa.gt.tri_first[mid] <= gid) lo = mid; else hi = mid;
    }
    if (hits) hits[dst] = make_uint4(v * tb.H + h, fa.gt.geom_ids[lo], (gid - fa.gt.tri_first[lo]) >> fa.gt.prim_shift[lo], __float_as_uint(t));
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_finish_pack(ProjectParams pp, FinishPackArgs fa, unsigned long long *__restrict__ stats)
{
    __shared__ uint32_t scratch[576];
    finish_pack_body<COUNT>(pp, fa, blockIdx.x, scratch, stats);
}

// One launch per frame: the workgroups of this frame's k_project and, from workgroup fp_start on, those
// of the previous frame's finish + pack (contiguous and in order, as their chained prefix needs).
// (at most 80 SGPRs: one more and a CU admits 7 of these workgroups instead of 8, MI355X_MICROARCH.md "Residency")
template <bool LDS_TABLES, bool MULTI, bool CULLED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_frame(ProjectParams pp, GeomBatch batch, uint32_t fp_start,
                                                  unsigned long long *__restrict__ best, BigItem *__restrict__ big,
                                                  uint32_t big_capacity, uint32_t *__restrict__ big_count, FinishPackArgs fa,
                                                  const uint32_t *__restrict__ cull_list)
{
    __shared__ ProjectLds lds;
    extern __shared__ __attribute__((aligned(16))) float s_chan[];
    const uint32_t rel = blockIdx.x - fp_start;
    if (rel < fa.n_blocks)
        finish_pack_body<false>(pp, fa, rel, reinterpret_cast<uint32_t *>(&lds), nullptr);
    else
        project_body<false, LDS_TABLES, MULTI, CULLED>(pp, batch, blockIdx.x < fp_start ? blockIdx.x : blockIdx.x - fa.n_blocks, lds, s_chan, best,
                                               big, big_capacity, big_count, nullptr, cull_list);
}

}  // namespace

void launch_project_finish(hipStream_t s, const ProjectParams &pp, unsigned long long *best, const void *big,
                           uint32_t big_capacity, const uint32_t *big_count, uint32_t *block_counts,
                           unsigned long long *stats, uint32_t rays_per_lane)
{
    const uint32_t n = pp.tb.V * pp.tb.naz;
    if (!n) return;
    const dim3 grid((n + kBlock - 1) / kBlock);
    if (stats)
        hipLaunchKernelGGL(k_project_finish<true>, grid, dim3(kBlock), 0, s, pp, best, static_cast<const BigItem *>(big),
                           big_capacity, big_count, block_counts, stats);
    else if (rays_per_lane >= 4u)
        launch_k(k_project_finish_wide<4>, dim3((grid.x + 3u) / 4u), dim3(kBlock), 0, s, pp, best, static_cast<const BigItem *>(big), big_capacity,
                 big_count, block_counts);
    else if (rays_per_lane >= 2u)
        launch_k(k_project_finish_wide<2>, dim3((grid.x + 1u) / 2u), dim3(kBlock), 0, s, pp, best, static_cast<const BigItem *>(big), big_capacity,
                 big_count, block_counts);
    else
        launch_k(k_project_finish<false>, grid, dim3(kBlock), 0, s, pp, best, static_cast<const BigItem *>(big),
                 big_capacity, big_count, block_counts, stats);
}

size_t project_big_item_bytes() { return sizeof(BigItem); }

void launch_project_init(hipStream_t s, const ProjectParams &pp, unsigned long long *best, uint32_t *big_count,
                         uint32_t *block_counts2)
{
    const uint32_t nq = pp.tb.V * pp.tb.naz;
    if (!nq) return;
    (void)hipMemsetAsync(best, 0xFF, (size_t)nq * sizeof(unsigned long long), s);
    (void)hipMemsetAsync(big_count, 0, 4 * kCounterSlotWords * sizeof(uint32_t), s);   // every slot's queue length and survivor counts (only ever called with nothing in flight)
    (void)hipMemsetAsync(block_counts2, 0, 2 * (size_t)((nq + kBlock - 1) / kBlock) * sizeof(uint32_t), s);
}

namespace {
uint32_t min_tpw() { static const uint32_t v = (uint32_t)std::max(1, lsi::tune_int("LS_PROJECT_MIN_TPW", 1)); return v; }
uint32_t tpw_waves() { static const uint32_t v = (uint32_t)lsi::tune_int("LS_PROJECT_TPW_WAVES", 8192); return v; }

}  // namespace

// 64 triangles per wave when that still gives every SIMD a few waves; fewer for small meshes
uint32_t project_tris_per_wave(uint32_t ntris)
{
    uint32_t tpw = 64u;
    while (tpw > min_tpw() && (ntris + tpw - 1u) / tpw < tpw_waves()) tpw >>= 1;
    return tpw;
}

namespace {
uint32_t cull_grid_pct() { static const uint32_t v = (uint32_t)std::min(100, std::max(1, lsi::tune_int("LS_PROJECT_CULL_GRID_PCT", 50))); return v; }

// workgroups of k_cull that are resident at once: 8 per CU (four waves each, ~9 KB of LDS)
uint32_t cull_resident_workgroups()
{
    static const uint32_t n = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return 8u * (uint32_t)prop.multiProcessorCount;
        return 2048u;
    }();
    return n;
}

LinearMap linear_map(const GeomSource &src)
{
    LinearMap m;
    if (src.xform == 0) {
        for (int i = 0; i < 9; ++i) m.l[i] = (i % 4 == 0) ? 1.f : 0.f;
        m.o[0] = m.o[1] = m.o[2] = 0.f;
        m.nrm = 1.00001f;
        return m;
    }
    const Affine &a = src.m;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            m.l[3 * i + j] = a.rinv[3 * i] * a.a[j] + a.rinv[3 * i + 1] * a.a[4 + j] + a.rinv[3 * i + 2] * a.a[8 + j];
        m.o[i] = a.rinv[3 * i] * (a.a[3] - a.t[0]) + a.rinv[3 * i + 1] * (a.a[7] - a.t[1]) + a.rinv[3 * i + 2] * (a.a[11] - a.t[2]);
    }
    // an upper bound of the largest singular value of L: the largest absolute row sum of L^T L bounds its largest eigenvalue
    // (exactly 1 for a rigid pose, where L^T L is the identity); it only scales a radius that enters second-order terms and
    // the sector reach
    double top = 0.0;
    for (int i = 0; i < 3; ++i) {
        double row = 0.0;
        for (int j = 0; j < 3; ++j) {
            double sij = 0.0;
            for (int k = 0; k < 3; ++k) sij += (double)m.l[3 * k + i] * (double)m.l[3 * k + j];
            row += std::fabs(sij);
        }
        top = std::max(top, row);
    }
    m.nrm = (float)(std::sqrt(top) * 1.00001);
    return m;
}

// the geometries that carry group bounds, as one batch: false if there are none or more than a launch takes.
// blocks = k_project's grid (worst case: every group survives), entries = survivor-list words, batch.cull_first[n] = k_cull's grid
bool fill_culled_batch(const GeomSource *srcs, uint32_t n_srcs, GeomBatch &batch, uint32_t &blocks, uint32_t &entries, bool sector, uint32_t grid_pct = 0,
                       uint32_t survivors_hint = 0)
{
    if (!grid_pct) grid_pct = cull_grid_pct();
    batch.n = 0;
    blocks = entries = 0;
    uint32_t cull_blocks = 0;
    constexpr uint32_t kGroupsPerWorkgroup = (kBlock / 64) * (64u / kCullGroup);   // survivors one k_project workgroup takes
    // k_cull's groups per workgroup: the fewest rounds (from 2) with which all its workgroups are resident at once (8 per CU)
    unsigned long long all_groups = 0;
    for (uint32_t i = 0; i < n_srcs; ++i)
        if (srcs[i].boxes && srcs[i].ntris) all_groups += (srcs[i].ntris + kCullGroup - 1) / kCullGroup;
    uint32_t rounds = 2;
    while (rounds < kCullMaxRounds && (all_groups + (unsigned long long)rounds * kBlock - 1) / ((unsigned long long)rounds * kBlock) > cull_resident_workgroups()) ++rounds;
    // Under an azimuth shard most workgroups end at their block bounds (k_cull's early out) and the ones that stay are as
    // long as their rounds, which a wave takes two at a time, each pair a memory round trip and a test behind the other: the
    // few live workgroups of a shard do better with a short chain -- but every workgroup that comes and goes is four waves
    // to launch, and three frames overlap: the shortest chain is not the best either
    // (an eighth of a turn at SYN-10M, three frames in flight as graphs, ranks 0 / 4 / 5 -- the sector with the fewest triangles and
    // the two with the most: 4 rounds 9.9 / 12.1 / 12.9 us per frame, the full turn's own 5: 10.4 / 11.8 / 12.5, 6: 10.1 / 12.6 / 12.9, 8: 10.8 / 12.4 /
    // 12.9; at SYN-1M the full turn's own 2: 8.4 / 8.2 / 8.6, 4: 8.6 / 9.7 / 11.0, 8: 10.5 / 10.6 / 11.5.  A group's frame is its slowest
    // rank's: a shard takes the full turn's choice -- one resident round of workgroups -- as it is)
    if (sector) { static const int r = lsi::tune_int("LS_CULL_SHARD_ROUNDS", 0); if (r) rounds = (uint32_t)std::min<int>(std::max(r, 2), (int)kCullMaxRounds); }
    batch.cull_rounds = rounds;
    const uint32_t per_wg = rounds * kBlock;
    for (uint32_t i = 0; i < n_srcs; ++i) {
        const GeomSource &src = srcs[i];
        if (!src.boxes || !src.ntris) continue;
        if (batch.n == (uint32_t)kGeomsPerLaunch) return false;
        const uint32_t groups = (src.ntris + kCullGroup - 1) / kCullGroup;
        const uint32_t cull_wgs = (groups + per_wg - 1) / per_wg;                     // a k_cull workgroup belongs to one geometry
        const uint32_t seg_cap = (cull_wgs + kCullSegs - 1) / kCullSegs * per_wg;     // a segment takes every kCullSegs-th workgroup's survivors
        batch.block_first[batch.n] = blocks;
        batch.tris_per_wave[batch.n] = 64u;
        batch.list_first[batch.n] = entries;
        batch.cull_first[batch.n] = cull_blocks;
        batch.seg_cap[batch.n] = seg_cap;
        batch.g[batch.n] = src;
        // k_project's workgroups per segment: room for cull_grid_pct() % of the groups to survive (a wave loops when more do)
        uint32_t seg_blocks = std::max(1u, (uint32_t)(((unsigned long long)(seg_cap / kGroupsPerWorkgroup) * grid_pct + 99u) / 100u));
        if (survivors_hint) {
            // A recent frame's fullest segment held survivors_hint - 1 groups (read back without a wait: k_pack leaves the
            // number in pinned host memory): a quarter of headroom on that, rounded up to an eighth of itself so that the
            // grid -- a node parameter of a captured frame graph -- changes only when the scene does.  Never more than the
            // percentage above gives; a segment that outgrows it is walked in rounds (project_body's loop).
            const uint32_t need = ((survivors_hint - 1u) + (survivors_hint - 1u) / 4u + kGroupsPerWorkgroup - 1u) / kGroupsPerWorkgroup + 1u;
            uint32_t q = 1u;
            while (q * 16u <= need) q <<= 1;
            seg_blocks = std::min(seg_blocks, (need + q - 1u) / q * q);
        }
        batch.seg_blocks[batch.n] = seg_blocks;
        blocks += kCullSegs * seg_blocks;
        entries += kCullSegs * seg_cap;
        cull_blocks += cull_wgs;
        ++batch.n;
    }
    batch.block_first[batch.n] = blocks;
    batch.list_first[batch.n] = entries;
    batch.cull_first[batch.n] = cull_blocks;
    return batch.n > 0;
}
}  // namespace

uint32_t project_cull_entries(const GeomSource *srcs, uint32_t n_srcs, bool sector)
{
    GeomBatch batch;
    uint32_t blocks, entries;
    return fill_culled_batch(srcs, n_srcs, batch, blocks, entries, sector) ? entries : 0u;
}

void launch_project(hipStream_t s, const ProjectParams &pp, const GeomSource *srcs, uint32_t n_srcs, unsigned long long *best,
                    void *big, uint32_t big_capacity, uint32_t *big_count, unsigned long long *stats, const FinishPackArgs *rider,
                    uint32_t *cull_list, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t survivors_hint)
{
    if (!(pp.tb.V * pp.tb.naz)) return;
    TL_HOST_HOOK(s);
    static const int fp_at = lsi::tune_int("LS_PROJECT_FP_AT", -1);
    BigItem *bq = static_cast<BigItem *>(big);
    const bool lt = pp.tb.V <= 2048u;   // channel tables fit in LDS (40 KB at most)
    const size_t lds = (((5 * (size_t)pp.tb.V + 1) & ~(size_t)1) + (pp.cols_lds ? 2 * (size_t)pp.tb.naz : 0)) * sizeof(float);

    // one launch over `batch` (the previous frame's finish + pack workgroups ride in the first launch of the frame)
    auto launch = [&](const GeomBatch &batch, uint32_t blocks, const uint32_t *list) {
        const bool multi = batch.n > 1, culled = list != nullptr;
        if (rider && !stats) {
            const dim3 grid(blocks + rider->n_blocks);
            // where the riders sit in the grid: behind the triangle workgroups by default (they fill the tail)
            const uint32_t fp_start = fp_at < 0 ? blocks : std::min((uint32_t)fp_at, blocks);
#define LS_FRAME(L, M, C) hipLaunchKernelGGL((k_frame<L, M, C>), grid, dim3(kBlock), (L) ? lds : 0, s, pp, batch, fp_start, best, bq, big_capacity, big_count, *rider, list)
#define LS_FRAME_C(L, M) do { if (culled) LS_FRAME(L, M, true); else LS_FRAME(L, M, false); } while (0)
            if (lt) { if (multi) LS_FRAME_C(true, true); else LS_FRAME_C(true, false); }
            else { if (multi) LS_FRAME_C(false, true); else LS_FRAME_C(false, false); }
#undef LS_FRAME_C
#undef LS_FRAME
            rider = nullptr;
            return;
        }
        const dim3 grid(blocks);
        // ev_start / ev_stop (LS_OPT_TIMING = 2): the events ride on the dispatch packet itself and carry the kernel's own
        // begin / end timestamps (what rocprofv3 reports); events recorded around a launch add ~3 us of barrier packets
        const bool timed = ev_start || ev_stop;
        hipEvent_t e0 = ev_start;
        ev_start = nullptr;   // the first k_project launch of the frame starts the clock, the last one stops it
#define LS_LAUNCH(C, L, M, K, D) do { \
            if (timed) hipExtLaunchKernelGGL((k_project<C, L, M, K, D>), grid, dim3(kBlock), (L) ? (uint32_t)lds : 0u, s, e0, ev_stop, 0u, pp, batch, best, bq, big_capacity, big_count, stats, list); \
            else launch_k(k_project<C, L, M, K, D>, grid, dim3(kBlock), (L) ? (uint32_t)lds : 0u, s, pp, batch, best, bq, big_capacity, big_count, stats, list); } while (0)
#define LS_LAUNCH_K(C, L, M) do { if (culled && pp.cull_deal) LS_LAUNCH(C, L, M, true, true); else if (culled) LS_LAUNCH(C, L, M, true, false); else LS_LAUNCH(C, L, M, false, false); } while (0)
        if (lt) {
            if (stats) { if (multi) LS_LAUNCH_K(true, true, true); else LS_LAUNCH_K(true, true, false); }
            else { if (multi) LS_LAUNCH_K(false, true, true); else LS_LAUNCH_K(false, true, false); }
        } else {
            if (stats) { if (multi) LS_LAUNCH_K(true, false, true); else LS_LAUNCH_K(true, false, false); }
            else { if (multi) LS_LAUNCH_K(false, false, true); else LS_LAUNCH_K(false, false, false); }
        }
#undef LS_LAUNCH_K
#undef LS_LAUNCH
    };

    // ---- geometries with group bounds: k_cull, then a launch whose waves read the survivors
    bool culled_done = false;
    if (cull_list) {
        GeomBatch batch;
        std::memset(static_cast<void *>(&batch), 0, sizeof(batch));
        uint32_t blocks, entries;
        // k_project<CULLED>'s grid: room for half of the groups to survive; under an azimuth shard for the shard's share of
        // the turn (an eighth of a turn: 12 %; a wave whose segment holds more comes round again, so any size is right).
        // A workgroup that finds nothing still has to be launched, read its segment's count and retire -- and at ten
        // million triangles half of the groups are 31 744 workgroups, of which an eighth-of-a-turn shard feeds 1 100
        uint32_t grid_pct = cull_grid_pct();
        if (pp.sector_on && pp.tb.H) grid_pct = std::max(4u, std::min(grid_pct, (uint32_t)((100ull * pp.tb.naz + pp.tb.H - 1u) / pp.tb.H)));
        { static const int shard_pct = lsi::tune_int("LS_PROJECT_SHARD_GRID_PCT", 0); if (shard_pct > 0 && pp.sector_on) grid_pct = (uint32_t)shard_pct; }
        { static const int use_hint = lsi::tune_int("LS_PROJECT_GRID_HINT", 1); if (!use_hint) survivors_hint = 0; }
        if (fill_culled_batch(srcs, n_srcs, batch, blocks, entries, pp.sector_on != 0, grid_pct, survivors_hint)) {
            const dim3 cgrid(batch.cull_first[batch.n]);
            CullBatch cb;
            std::memset(static_cast<void *>(&cb), 0, sizeof(cb));   // (argument bytes are compared frame to frame by the frame graph)
            cb.n = batch.n;
            cb.rounds = batch.cull_rounds;
            for (uint32_t i = 0; i < batch.n; ++i) {
                cb.cull_first[i] = batch.cull_first[i];
                cb.list_first[i] = batch.list_first[i];
                cb.seg_cap[i] = batch.seg_cap[i];
                cb.g[i].boxes = batch.g[i].boxes;
                cb.g[i].ntris = batch.g[i].ntris;
                cb.g[i].m = linear_map(batch.g[i]);
            }
            cb.cull_first[batch.n] = batch.cull_first[batch.n];
            const bool lut = lt && pp.lut_ok;
            const uint32_t clds = lt ? (uint32_t)(2 * ((size_t)pp.tb.V + 2) * sizeof(float) + (lut ? kCullLutBuckets * sizeof(uint16_t) : 0)) : 0u;
            uint32_t *counts = big_count + kCullCountAt;
            // (ev_start: the cull pass belongs to the timed stage, the clock starts with it)
#define LS_CULL_S(L, C, U, S) do { \
                if (ev_start) hipExtLaunchKernelGGL((k_cull<L, C, U, S>), cgrid, dim3(kBlock), clds, s, ev_start, nullptr, 0u, pp, cb, cull_list, counts, stats); \
                else launch_k(k_cull<L, C, U, S>, cgrid, dim3(kBlock), clds, s, pp, cb, cull_list, counts, stats); } while (0)
#define LS_CULL(L, C, U) do { if (pp.sector_on) LS_CULL_S(L, C, U, true); else LS_CULL_S(L, C, U, false); } while (0)
            if (lut) { if (stats) LS_CULL(true, true, true); else LS_CULL(true, false, true); }
            else if (lt) { if (stats) LS_CULL(true, true, false); else LS_CULL(true, false, false); }
            else { if (stats) LS_CULL(false, true, false); else LS_CULL(false, false, false); }
#undef LS_CULL
#undef LS_CULL_S
            ev_start = nullptr;
            launch(batch, blocks, cull_list);
            culled_done = true;
        }
    }
    // ---- the others (small meshes, or everything when culling is off): up to kGeomsPerLaunch per launch
    uint32_t at = 0;
    while (at < n_srcs) {
        GeomBatch batch;
        std::memset(static_cast<void *>(&batch), 0, sizeof(batch));   // (unused entries and padding: the frame graph compares argument bytes)
        batch.n = 0;
        uint32_t blocks = 0;
        for (; at < n_srcs && batch.n < (uint32_t)kGeomsPerLaunch; ++at) {
            const GeomSource &src = srcs[at];
            if (!src.ntris || (culled_done && src.boxes)) continue;
            const uint32_t tpw = project_tris_per_wave(src.ntris);
            const uint32_t per_block = tpw * (kBlock / 64);
            batch.block_first[batch.n] = blocks;
            batch.tris_per_wave[batch.n] = tpw;
            batch.list_first[batch.n] = 0;
            batch.g[batch.n] = src;
            blocks += (src.ntris + per_block - 1) / per_block;
            ++batch.n;
        }
        if (!batch.n) break;
        batch.block_first[batch.n] = blocks;
        launch(batch, blocks, nullptr);
    }
    if (rider) launch_finish_pack(s, pp, *rider, stats);   // nothing was launched for it to ride with
}

bool launch_mesh_order(hipStream_t s, const uint8_t *verts, uint32_t stride, uint32_t nverts, const uint32_t *idx, uint32_t ntris,
                       uint32_t *aabb6, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, void *sort_temp, size_t sort_temp_bytes,
                       uint32_t *perm, uint32_t *idx_sorted)
{
    if (!ntris || !nverts) return true;
    if (!sort_temp || sort_temp_bytes < ls::sort_temp_bytes(ntris)) return false;   // (before anything is launched)
    (void)hipMemsetAsync(aabb6, 0xFF, 12, s);
    (void)hipMemsetAsync(aabb6 + 3, 0, 12, s);
    const uint32_t vgrid = std::min<uint32_t>((nverts + kBlock - 1) / kBlock, 256u);
    hipLaunchKernelGGL(k_mesh_aabb, dim3(vgrid), dim3(kBlock), 0, s, verts, stride, nverts, aabb6);
    const dim3 tgrid((ntris + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_mesh_morton, tgrid, dim3(kBlock), 0, s, verts, stride, idx, ntris, aabb6, keys_a, vals_a);
    if (!launch_sort(s, sort_temp, sort_temp_bytes, keys_a, keys_b, vals_a, perm, ntris)) return false;
    hipLaunchKernelGGL(k_permute_indices, tgrid, dim3(kBlock), 0, s, idx, perm, ntris, idx_sorted);
    return true;
}

size_t project_box_entries(uint32_t ntris)
{
    const size_t groups = (ntris + kCullGroup - 1) / kCullGroup;
    return groups + (groups + kCullBlockGroups - 1) / kCullBlockGroups;
}

void launch_group_bounds(hipStream_t s, const uint8_t *verts, uint32_t stride, const uint32_t *idx_sorted, uint32_t ntris, float4 *boxes)
{
    if (!ntris) return;
    const uint32_t groups = (ntris + kCullGroup - 1) / kCullGroup;
    hipLaunchKernelGGL(k_group_bounds, dim3((groups + kBlock - 1) / kBlock), dim3(kBlock), 0, s, verts, stride, idx_sorted, ntris, boxes,
                       boxes + 2 * (size_t)groups);
}

void launch_corners(hipStream_t s, const uint8_t *verts, uint32_t stride, const uint32_t *idx_sorted, const uint32_t *perm, uint32_t ntris,
                    float4 *corners)
{
    if (!ntris) return;
    hipLaunchKernelGGL(k_corners, dim3((ntris + kBlock - 1) / kBlock), dim3(kBlock), 0, s, verts, stride, idx_sorted, perm, ntris, corners);
}

void launch_finish_pack(hipStream_t s, const ProjectParams &pp, const FinishPackArgs &fa, unsigned long long *stats)
{
    if (!fa.n_blocks) return;
    if (stats) hipLaunchKernelGGL(k_finish_pack<true>, dim3(fa.n_blocks), dim3(kBlock), 0, s, pp, fa, stats);
    else launch_k(k_finish_pack<false>, dim3(fa.n_blocks), dim3(kBlock), 0, s, pp, fa, stats);
}

}  // namespace ls
