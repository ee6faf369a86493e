// k_project_multi_set.hip -- EXPERIMENT (round 4), kept for the record; not compiled into the library (it was, behind
// LS_PROJECT_SETS, while it was measured: the text below is that code -- kernel, then the launcher's branch).
// "Every wave finishes for itself, but over S times the triangles": S sets of 64 triangles per wave, all their loads up
// front, the band test S times with the transformed corners kept in registers, the survivors of all sets compacted into a
// list in LDS, ONE dense pass that fetches their corners from the owning lanes' registers (ds_bpermute), runs columns(),
// stages, scans and walks the cells in trips batched four at a time.  Parity-green (tools/exp_run.sh) for S = 2, 3, 4 and
// slower for each (rocprofv3, SYN-128 x SYN-1M, one frame in flight; k_project 16.55 us on the same box):
//     S = 2   7 814 waves, 80 VGPRs (6 waves per SIMD: two rounds)                         20.5 us
//     S = 3   5 209 waves, 80 VGPRs + 28 bytes of scratch to fit one round                 25.1 us
//     S = 4   3 907 waves, 98 VGPRs (4 waves per SIMD: one round)                          21.8 us
//     64 consecutive triangles per set instead of eight spread runs: S = 2 / 4             24.6 / 38.5 us
// and with THREE FRAMES IN FLIGHT (lsbench --pipeline 2, the headline's mode; k_project: 15.8 us per frame): S = 2 16.5 us,
// S = 4 20.4 us per frame -- 40 % fewer instructions buy nothing there either: the overlapped frame is not issue-bound.
// Ablation (S = 4 / S = 2): loads + band + compaction alone 10.9 / 10.6 us; + the dense pass without trips 15.7 / 14.2 us;
// + trips 21.8 / 20.5 us.  The streaming half does not get shorter with fewer, fatter waves: the load phase is a throughput
// (4.5 us whatever S, tools/micro/loads_probe.hip), and with four waves on a SIMD the band test's 680 dependent instructions
// run at ~10 cycles each (2.8 us) instead of hiding behind seven other waves; then every wave of the chip is in the dense pass
// at once, with the same few waves per SIMD.  Instructions per 256 triangles 2 280 -> ~1 300; kernel + 30 %.  Together with the
// block-tail form (k_project_block_tail.hip): the kernel is not bound by its instruction count but by how long the chain of
// ONE wave is times how many rounds of waves the chip holds, and every way of making lanes dense makes that chain longer.


// ------------------------------------------------------------------------------------------
// k_project_ms ("multi-set", round 4): S sets of 64 triangles per WAVE, the expensive half once per wave on dense lanes.
// k_project's wave pays ~570 VALU instructions for 64 triangles of which ~12 pass the band test: columns(), e1 / e2 / NgC
// and the staging (~270 instructions) run with a fifth of the lanes alive, the trips round ~69 cells up to whole trips of 64.
// The workgroup-level form of the cure (one wave of four finishes for all: tools/experiments/k_project_block_tail.hip) made
// the instruction count 40 % smaller and the kernel 70 % longer: the finishing wave's chain was the workgroup's lifetime.
// Here every wave finishes for ITSELF, but over S times the triangles: all S sets' loads go out first (the load phase is a
// throughput, tools/micro/loads_probe.hip: 4.5 us whatever S), the band test runs S times (transformed corners stay in
// registers), the survivors of all sets -- ~12 S -- are compacted by ballot + mbcnt into a list of (set, lane, band) in LDS,
// and ONE dense pass fetches their corners from the owning lanes' registers (ds_bpermute: the LDS crossbar, no LDS
// memory), runs columns(), stages, scans, and walks the ~69 S cells in trips batched kTrips at a time (one owner search, the
// direction loads up front, the tests, then the atomics).  ntris / (64 S) waves: one resident round for S >= 3 at 1 M
// triangles, no second round that waits for the first.  Same arithmetic, same atomics: bit-identical keys.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMsTrips = 4;
template <int S>
struct MsLds {
    uint32_t src[kBlock / 64][64 * S];    // band survivors of the wave: set << 6 | lane
    uint32_t band[kBlock / 64][64 * S];   // i0 | nch << 16
    float tri[kBlock / 64][10][64];       // the current pass's staged footprints: v0, e1, e2, NgC ...
    uint32_t meta[kBlock / 64][6][64];    // ... gid, i0, h0a, na, h0b, nb
    uint32_t pref[kBlock / 64][64];
    uint8_t flag[kBlock / 64][64 * kMsTrips];
};

template <bool COUNT, bool LDS_TABLES, bool MULTI, int S>
__global__ __launch_bounds__(kBlock, S == 2 ? 8 : (S == 3 ? 6 : 4)) void k_project_ms(ProjectParams pp, GeomBatch batch, unsigned long long *__restrict__ best,
                                                       BigItem *__restrict__ big, uint32_t big_capacity, uint32_t *__restrict__ big_count,
                                                       unsigned long long *__restrict__ stats)
{
    __shared__ MsLds<S> lds;
    extern __shared__ float s_chan[];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t gi = 0;
    if (MULTI)
        while (gi + 1u < batch.n && blockIdx.x >= batch.block_first[gi + 1u]) ++gi;
    const GeomSource &src = batch.g[gi];
    const uint32_t block = MULTI ? blockIdx.x - batch.block_first[gi] : blockIdx.x;
    // ---- the wave's S sets (set = 64 triangles: eight runs of eight a stride of the set count apart, as k_project's spread
    //      runs; or 64 consecutive ones) and ALL their loads
    const uint32_t n_sets = (src.ntris + 63u) / 64u, rank = block * (kBlock / 64) + w;
    uint32_t k[S];
    float v[S][9];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t set = rank * S + (uint32_t)s;
        k[s] = 0xFFFFFFFFu;
        if (set < n_sets) k[s] = pp.spread ? ((lane >> 3) * n_sets + set) * 8u + (lane & 7u) : set * 64u + lane;
        if (k[s] >= src.ntris) k[s] = 0xFFFFFFFFu;
    }
    uint32_t ia[S], ib[S], ic[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        ia[s] = ib[s] = ic[s] = 0;
        if (k[s] != 0xFFFFFFFFu) { ia[s] = src.idx[3 * (size_t)k[s] + 0]; ib[s] = src.idx[3 * (size_t)k[s] + 1]; ic[s] = src.idx[3 * (size_t)k[s] + 2]; }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int j = 0; j < 9; ++j) v[s][j] = 0.f;
        if (k[s] != 0xFFFFFFFFu) {
            const float *pa = reinterpret_cast<const float *>(src.verts + (size_t)ia[s] * src.stride);
            const float *pb = reinterpret_cast<const float *>(src.verts + (size_t)ib[s] * src.stride);
            const float *pc = reinterpret_cast<const float *>(src.verts + (size_t)ic[s] * src.stride);
            v[s][0] = pa[0]; v[s][1] = pa[1]; v[s][2] = pa[2];
            v[s][3] = pb[0]; v[s][4] = pb[1]; v[s][5] = pb[2];
            v[s][6] = pc[0]; v[s][7] = pc[1]; v[s][8] = pc[2];
        }
    }
    ChanTables ct = {pp.chan_tan_up, pp.chan_tan_dn, pp.tb.sin_theta, pp.tb.cos_theta, pp.chan_perm};
    if (LDS_TABLES) {
        const uint32_t V = pp.tb.V;
        for (uint32_t i = threadIdx.x; i < V; i += kBlock) {
            s_chan[i] = pp.chan_tan_up[i];
            s_chan[V + i] = pp.chan_tan_dn[i];
            s_chan[2 * V + i] = pp.tb.sin_theta[i];
            s_chan[3 * V + i] = pp.tb.cos_theta[i];
            s_chan[4 * V + i] = __uint_as_float(pp.chan_perm[i]);
        }
        ct = {s_chan, s_chan + V, s_chan + 2 * V, s_chan + 3 * V, reinterpret_cast<const uint32_t *>(s_chan + 4 * V)};
        __syncthreads();   // the only workgroup barrier
    }
    // ---- streaming half, S times: transform in place, sector test, band; survivors to the wave's list
    uint32_t n_keep = 0;   // uniform
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint32_t i0 = 0, nch = 0;
        if (k[s] != 0xFFFFFFFFu) {
            V3 v0, v1, v2;
            if (src.xform == 1) {
                v0 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(&v[s][0]));
                v1 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(&v[s][3]));
                v2 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(&v[s][6]));
            } else if (src.xform == 2) {
                v0 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(&v[s][0]));
                v1 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(&v[s][3]));
                v2 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(&v[s][6]));
            } else {
                v0 = {v[s][0], v[s][1], v[s][2]}; v1 = {v[s][3], v[s][4], v[s][5]}; v2 = {v[s][6], v[s][7], v[s][8]};
            }
            v[s][0] = v0.x; v[s][1] = v0.y; v[s][2] = v0.z; v[s][3] = v1.x; v[s][4] = v1.y; v[s][5] = v1.z;
            v[s][6] = v2.x; v[s][7] = v2.y; v[s][8] = v2.z;
            bool outside = false;
            if (pp.sector_on) {
                const float a0 = pp.sec_a[0] * v0.y - pp.sec_a[1] * v0.x, a1 = pp.sec_a[0] * v1.y - pp.sec_a[1] * v1.x,
                            a2 = pp.sec_a[0] * v2.y - pp.sec_a[1] * v2.x;
                const float b0 = v0.x * pp.sec_b[1] - v0.y * pp.sec_b[0], b1 = v1.x * pp.sec_b[1] - v1.y * pp.sec_b[0],
                            b2 = v2.x * pp.sec_b[1] - v2.y * pp.sec_b[0];
                outside = (a0 < 0.0f && a1 < 0.0f && a2 < 0.0f) || (b0 < 0.0f && b1 < 0.0f && b2 < 0.0f);
            }
            if (!outside) band(pp, ct, v0, v1, v2, i0, nch);
        }
        const bool keep = nch != 0u;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t at = n_keep + lanes_below(mask);
            lds.src[w][at] = ((uint32_t)s << 6) | lane;
            lds.band[w][at] = i0 | (nch << 16);
        }
        n_keep += (uint32_t)__popcll(mask);
    }
    wave_lds_fence();
    if (pp.debug == 3) { if (n_keep == 0xFFFFFFFFu) best[0] = 0; return; }
    // ---- the dense half: 64 survivors to a pass
    unsigned long long cells_total = 0;
    for (uint32_t c0 = 0; c0 < n_keep; c0 += 64u) {
        const uint32_t e = c0 + lane;
        const bool have = e < n_keep;
        const uint32_t sl = have ? lds.src[w][e] : lane, bd = have ? lds.band[w][e] : 0u;
        const uint32_t from = sl & 63u, set_of = sl >> 6;
        // the survivor's transformed corners, out of the owning lane's registers (every lane takes part in the exchange)
        float c[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float x = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float y = __shfl(v[s][j], (int)from, 64);
                x = set_of == (uint32_t)s ? y : x;
            }
            c[j] = x;
        }
        uint32_t cells = 0, slot = 0;
        if (have) {
            const V3 v0 = {c[0], c[1], c[2]}, v1 = {c[3], c[4], c[5]}, v2 = {c[6], c[7], c[8]};
            const uint32_t set = rank * S + set_of;
            const uint32_t kk = pp.spread ? ((from >> 3) * n_sets + set) * 8u + (from & 7u) : set * 64u + from;
            const uint32_t gid = src.gid_first + (src.perm ? src.perm[kk] : kk);
            Foot f = {bd & 0xFFFFu, bd >> 16, 0, 0, 0, 0};
            columns(pp, v0, v1, v2, f.h0a, f.na, f.h0b, f.nb);
            cells = f.nch * (f.na + f.nb);
            if (cells) {
                const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
                const float NgC = dot_fma(cross_fma(e2, e1), v0);
                bool queued = false;
                if (cells > pp.big_cells) {
                    const uint32_t qs = atomicAdd(big_count, 1u);
                    if (qs < big_capacity) {
                        BigItem it;
                        it.v0[0] = v0.x; it.v0[1] = v0.y; it.v0[2] = v0.z;
                        it.e1[0] = e1.x; it.e1[1] = e1.y; it.e1[2] = e1.z;
                        it.e2[0] = e2.x; it.e2[1] = e2.y; it.e2[2] = e2.z;
                        it.NgC = NgC; it.gid = gid; it.i0 = f.i0; it.nch = f.nch;
                        it.h0a = f.h0a; it.na = f.na; it.h0b = f.h0b; it.nb = f.nb;
                        it.pad[0] = it.pad[1] = it.pad[2] = 0;
                        big[qs] = it;
                        queued = true;
                    }
                }
                if (queued) {
                    cells = 0;
                } else {
                    slot = lanes_below(__ballot(true));
                    lds.tri[w][0][slot] = v0.x; lds.tri[w][1][slot] = v0.y; lds.tri[w][2][slot] = v0.z;
                    lds.tri[w][3][slot] = e1.x; lds.tri[w][4][slot] = e1.y; lds.tri[w][5][slot] = e1.z;
                    lds.tri[w][6][slot] = e2.x; lds.tri[w][7][slot] = e2.y; lds.tri[w][8][slot] = e2.z;
                    lds.tri[w][9][slot] = NgC;
                    lds.meta[w][0][slot] = gid; lds.meta[w][1][slot] = f.i0; lds.meta[w][2][slot] = f.h0a;
                    lds.meta[w][3][slot] = f.na; lds.meta[w][4][slot] = f.h0b; lds.meta[w][5][slot] = f.nb;
                }
            }
        }
        const uint32_t incl = wave_inclusive_scan(cells);
        uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (pp.debug == 4) { if (total == 0xFFFFFFFFu) best[0] = 0; total = 0; }
        const uint32_t n_slots = (uint32_t)__popcll(__ballot(cells != 0));
        if (cells) lds.pref[w][slot] = incl - cells;
        wave_lds_fence();
        const uint32_t first = lane < n_slots ? lds.pref[w][lane] : 0xFFFFFFFFu;
        // trips of 64 cells, kMsTrips to a batch: one owner search (a flag per first cell over the batch's cells, one ballot per
        // trip), then every trip's (channel, column) and direction load, then the tests, then the atomics
        for (uint32_t jb = 0; jb < total; jb += 64u * kMsTrips) {
            const uint32_t nb = min(kMsTrips, (total - jb + 63u) / 64u);   // trips of this batch (uniform)
            reinterpret_cast<uint32_t *>(lds.flag[w])[lane] = 0u;         // (64 lanes x 4 bytes = the 256 flags)
            wave_lds_fence();
            if (first - jb < 64u * kMsTrips) lds.flag[w][first - jb] = 1;
            wave_lds_fence();
            uint32_t own[kMsTrips], cv[kMsTrips], ch[kMsTrips];
            float2 cs[kMsTrips];
            bool live[kMsTrips];
#pragma unroll
            for (uint32_t b = 0; b < kMsTrips; ++b) {
                live[b] = false; own[b] = 0; cv[b] = 0; ch[b] = pp.tb.az0; cs[b] = make_float2(0.f, 0.f);
                if (b < nb) {   // uniform
                    const uint32_t j = jb + 64u * b + lane;
                    const uint32_t before = (uint32_t)__popcll(__ballot(first < jb + 64u * b));
                    const unsigned long long starts = __ballot(lds.flag[w][64u * b + lane] != 0);
                    live[b] = j < total;
                    if (live[b]) {
                        const uint32_t lo = before + lanes_below(starts) + (uint32_t)((starts >> lane) & 1ull) - 1u;
                        own[b] = lo;
                        foot_cell(ct, lds.meta[w][1][lo], lds.meta[w][2][lo], lds.meta[w][3][lo], lds.meta[w][4][lo], lds.meta[w][5][lo],
                                  j - lds.pref[w][lo], cv[b], ch[b]);
                    }
                    cs[b] = pp.tb.cs_phi[ch[b]];   // (idle lanes read a valid column too: no branch around the load)
                }
            }
            unsigned long long key[kMsTrips];
#pragma unroll
            for (uint32_t b = 0; b < kMsTrips; ++b) {
                key[b] = ~0ull;
                if (live[b]) {
                    const uint32_t lo = own[b];
                    const float st = ct.sin_theta[cv[b]];
                    const V3 d = {st * cs[b].x, st * cs[b].y, ct.cos_theta[cv[b]]};   // LidarDevice.cpp:310-316
                    float t;
                    if (tri_test(d, {lds.tri[w][0][lo], lds.tri[w][1][lo], lds.tri[w][2][lo]}, {lds.tri[w][3][lo], lds.tri[w][4][lo], lds.tri[w][5][lo]},
                                 {lds.tri[w][6][lo], lds.tri[w][7][lo], lds.tri[w][8][lo]}, lds.tri[w][9][lo], t))
                        key[b] = ((unsigned long long)__float_as_uint(t) << 32) | lds.meta[w][0][lo];
                }
            }
#pragma unroll
            for (uint32_t b = 0; b < kMsTrips; ++b)
                if (key[b] != ~0ull) atomicMin(&best[(size_t)cv[b] * pp.tb.naz + (ch[b] - pp.tb.az0)], key[b]);
            wave_lds_fence();
        }
        cells_total += total;
    }
    if (COUNT && lane == 0 && cells_total) atomicAdd(&stats[0], cells_total);
}


// ---- launch_project's branch (inside the `launch` lambda, before the plain launch)
#if 0
        // the multi-set form (k_project_ms) where it applies: no culling, every geometry of the launch at 64 triangles per wave
        const int ms = culled ? 0 : multi_sets();
        bool use_ms = ms >= 2;
        for (uint32_t i = 0; i < batch.n && use_ms; ++i) use_ms = batch.tris_per_wave[i] == 64u;
        if (use_ms) {
            // the same geometries, S sets to a wave: the workgroups of geometry i shrink by S
            GeomBatch mb = batch;
            uint32_t mblocks = 0;
            for (uint32_t i = 0; i < mb.n; ++i) {
                const uint32_t per_block = 64u * (uint32_t)ms * (kBlock / 64);
                mb.block_first[i] = mblocks;
                mblocks += (mb.g[i].ntris + per_block - 1) / per_block;
            }
            mb.block_first[mb.n] = mblocks;
            const dim3 mgrid(mblocks);
            const bool timed_ms = ev_start || ev_stop;
            hipEvent_t e0m = ev_start;
            ev_start = nullptr;
#define LS_LAUNCH_MS(C, L, M, SS) do { \
            if (timed_ms) hipExtLaunchKernelGGL((k_project_ms<C, L, M, SS>), mgrid, dim3(kBlock), (L) ? (uint32_t)lds : 0u, s, e0m, ev_stop, 0u, pp, mb, best, bq, big_capacity, big_count, stats); \
            else launch_k(k_project_ms<C, L, M, SS>, mgrid, dim3(kBlock), (L) ? (uint32_t)lds : 0u, s, pp, mb, best, bq, big_capacity, big_count, stats); } while (0)
#define LS_LAUNCH_MS_S(C, L, M) do { if (ms == 2) LS_LAUNCH_MS(C, L, M, 2); else if (ms == 3) LS_LAUNCH_MS(C, L, M, 3); else LS_LAUNCH_MS(C, L, M, 4); } while (0)
            if (lt) {
                if (stats) { if (multi) LS_LAUNCH_MS_S(true, true, true); else LS_LAUNCH_MS_S(true, true, false); }
                else { if (multi) LS_LAUNCH_MS_S(false, true, true); else LS_LAUNCH_MS_S(false, true, false); }
            } else {
                if (stats) { if (multi) LS_LAUNCH_MS_S(true, false, true); else LS_LAUNCH_MS_S(true, false, false); }
                else { if (multi) LS_LAUNCH_MS_S(false, false, true); else LS_LAUNCH_MS_S(false, false, false); }
            }
#undef LS_LAUNCH_MS_S
#undef LS_LAUNCH_MS
            return;
        }
#endif
