// k_project_two_launches.hip -- EXPERIMENT (round 4), kept for the record; not compiled into the library (it was, behind
// LS_OPT_SPLIT_PROJECT / LS_PROJECT_SPLIT, while it was measured: the text below is that code -- the two kernels, then the
// launcher's branch; the handle kept one scratch buffer per frame in flight, 12 KB + 4 bytes per 256 triangles).
// The cross-KERNEL form of "dense lanes for the expensive half" (the cross-workgroup form is k_project_block_tail.hip, the
// per-wave form k_project_multi_set.hip): k_band = loads + transform + band test of every triangle, survivors packed at the
// front of the workgroup's slot; k_cells = workgroup b takes the survivors of k_band's workgroup b on its first waves and
// runs column intervals, staging and the cell tests; waves behind the survivors leave at once.  No atomics, no scan across
// workgroups, no chain made longer.  Parity-green (tools/exp_run.sh: 37 passed) and slower:
//     one frame in flight (rocprofv3, SYN-128 x SYN-1M):  k_band 10.96 + k_cells 13.76 = 24.7 us   against k_project 16.49 us
//     three frames in flight (tools/variance_probe.py):   18.5 us per frame                        against 15.55 us
// k_project's own phases, cut off one by one (LS_PROJECT_DEBUG): loads + transform 5.9 us, + band test 8.5, + columns 10.4,
// + staging and cells 16.5 -- so k_band is the first 8.5 us plus 2.5 us of compaction and item stores, and k_cells, which
// replaces 8 us of k_project's timeline, takes 13.8: every one of its 3 907 workgroups still stages the channel tables and
// starts at least one wave (with spread runs the survivors are uniform: ~49 per workgroup, one wave of four, 77 % full, that
// walks 4 - 5 trips where k_project's four waves walked 1 - 2 each side by side), and it pays a dependent launch's ~5 us.
// A second launch costs more than the idle lanes it removes.


// ------------------------------------------------------------------------------------------
// k_project in two launches (one big mesh, no culling; LS_OPT_SPLIT_PROJECT).  Four of five triangles of a large scene
// end at the band test, and in k_project their lanes idle through everything behind it.  k_band runs k_project's first
// half on every triangle -- loads, transform, band test -- and leaves the survivors of a workgroup packed at the front of
// the workgroup's slot of `items` (three float4 per survivor: the transformed corners, the triangle id, the channel
// range) with their number in `counts`; k_cells gives workgroup b the survivors of k_band's workgroup b: its first
// counts[b] lanes -- whole waves but the last -- take one each and run k_project's second half (column intervals,
// staging, cell tests); waves behind them leave at once.  No atomics, no scan across workgroups.
// ------------------------------------------------------------------------------------------
template <bool LDS_TABLES>
__global__ __launch_bounds__(kBlock) void k_band(ProjectParams pp, GeomBatch batch, float4 *__restrict__ items, uint32_t *__restrict__ counts)
{
    extern __shared__ float s_chan[];   // LDS_TABLES: tan_up, tan_dn
    __shared__ uint32_t s_wcnt[kBlock / 64];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const GeomSource &src = batch.g[0];
    uint32_t block = blockIdx.x;
    if (pp.xcd_remap) {
        const uint32_t nb = batch.block_first[1];
        const uint32_t x = block & 7u, i = block >> 3, q = nb >> 3, r = nb & 7u;
        block = x * q + min(x, r) + i;
    }
    uint32_t k = 0xFFFFFFFFu;
    if (pp.spread) {
        const uint32_t n_waves = (src.ntris + 63u) / 64u, rank = block * (kBlock / 64) + w;
        if (rank < n_waves) k = ((lane >> 3) * n_waves + rank) * 8u + (lane & 7u);
    } else {
        k = (block * (kBlock / 64) + w) * 64u + lane;
    }
    float raw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < src.ntris) {
        const uint32_t a = src.idx[3 * (size_t)k + 0], b = src.idx[3 * (size_t)k + 1], c = src.idx[3 * (size_t)k + 2];
        const float *pa = reinterpret_cast<const float *>(src.verts + (size_t)a * src.stride);
        const float *pb = reinterpret_cast<const float *>(src.verts + (size_t)b * src.stride);
        const float *pc = reinterpret_cast<const float *>(src.verts + (size_t)c * src.stride);
        raw[0] = pa[0]; raw[1] = pa[1]; raw[2] = pa[2];
        raw[3] = pb[0]; raw[4] = pb[1]; raw[5] = pb[2];
        raw[6] = pc[0]; raw[7] = pc[1]; raw[8] = pc[2];
    }
    ChanTables ct = {pp.chan_tan_up, pp.chan_tan_dn, pp.tb.sin_theta, pp.tb.cos_theta, pp.chan_perm};
    if (LDS_TABLES) {
        const uint32_t V = pp.tb.V;
        for (uint32_t i = threadIdx.x; i < V; i += kBlock) {
            s_chan[i] = pp.chan_tan_up[i];
            s_chan[V + i] = pp.chan_tan_dn[i];
        }
        ct.tan_up = s_chan;
        ct.tan_dn = s_chan + V;
        __syncthreads();
    }
    V3 v0 = {0.f, 0.f, 0.f}, v1 = v0, v2 = v0;
    uint32_t i0 = 0, nch = 0;
    if (k < src.ntris) {
        if (src.xform == 1) {
            v0 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw));
            v1 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw + 3));
            v2 = xform_vertex(src.m, reinterpret_cast<const uint8_t *>(raw + 6));
        } else if (src.xform == 2) {
            v0 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw));
            v1 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw + 3));
            v2 = xform_vertex_sensor_only(src.m, reinterpret_cast<const uint8_t *>(raw + 6));
        } else {
            v0 = {raw[0], raw[1], raw[2]}; v1 = {raw[3], raw[4], raw[5]}; v2 = {raw[6], raw[7], raw[8]};
        }
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
    if (lane == 0) s_wcnt[w] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t at = lanes_below(mask), total = 0;
#pragma unroll
    for (uint32_t q = 0; q < kBlock / 64; ++q) {
        const uint32_t c = s_wcnt[q];
        at += q < w ? c : 0u;
        total += c;
    }
    if (keep) {
        const uint32_t gid = src.gid_first + (src.perm ? src.perm[k] : k);
        float4 *it = items + 3 * ((size_t)blockIdx.x * kBlock + at);
        it[0] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(gid));
        it[1] = make_float4(v1.x, v1.y, v1.z, __uint_as_float(i0));
        it[2] = make_float4(v2.x, v2.y, v2.z, __uint_as_float(nch));
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

template <bool LDS_TABLES>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_cells(ProjectParams pp, const float4 *__restrict__ items,
                                                    const uint32_t *__restrict__ counts, unsigned long long *__restrict__ best,
                                                    BigItem *__restrict__ big, uint32_t big_capacity, uint32_t *__restrict__ big_count)
{
    __shared__ ProjectLds lds;
    extern __shared__ float s_chan[];
    auto &s_tri = lds.tri;
    auto &s_meta = lds.meta;
    auto &s_pref = lds.pref;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)counts[blockIdx.x]);
    if (!n) return;   // (uniform: before the barrier)
    const bool have = threadIdx.x < n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (have) {
        const float4 *it = items + 3 * ((size_t)blockIdx.x * kBlock + threadIdx.x);
        a = it[0]; b = it[1]; c = it[2];
    }
    ChanTables ct = {pp.chan_tan_up, pp.chan_tan_dn, pp.tb.sin_theta, pp.tb.cos_theta, pp.chan_perm};
    if (LDS_TABLES) {
        const uint32_t V = pp.tb.V;
        for (uint32_t i = threadIdx.x; i < V; i += kBlock) {
            s_chan[2 * V + i] = pp.tb.sin_theta[i];
            s_chan[3 * V + i] = pp.tb.cos_theta[i];
            s_chan[4 * V + i] = __uint_as_float(pp.chan_perm[i]);
        }
        ct = {s_chan, s_chan + V, s_chan + 2 * V, s_chan + 3 * V, reinterpret_cast<const uint32_t *>(s_chan + 4 * V)};
        __syncthreads();
    }
    if (w * 64u >= n) return;   // (no barrier follows)
    uint32_t cells = 0, slot = 0;
    if (have) {
        const V3 v0 = {a.x, a.y, a.z}, v1 = {b.x, b.y, b.z}, v2 = {c.x, c.y, c.z};
        const uint32_t gid = __float_as_uint(a.w);
        Foot f = {__float_as_uint(b.w), __float_as_uint(c.w), 0, 0, 0, 0};
        columns(pp, v0, v1, v2, f.h0a, f.na, f.h0b, f.nb);
        cells = f.nch * (f.na + f.nb);
        if (cells) {
            const V3 e1 = sub(v0, v1), e2 = sub(v2, v0);
            const float NgC = dot_fma(cross_fma(e2, e1), v0);
            bool queued = false;
            if (cells > pp.big_cells) {
                const uint32_t qslot = atomicAdd(big_count, 1u);
                if (qslot < big_capacity) {
                    BigItem it;
                    it.v0[0] = v0.x; it.v0[1] = v0.y; it.v0[2] = v0.z;
                    it.e1[0] = e1.x; it.e1[1] = e1.y; it.e1[2] = e1.z;
                    it.e2[0] = e2.x; it.e2[1] = e2.y; it.e2[2] = e2.z;
                    it.NgC = NgC; it.gid = gid; it.i0 = f.i0; it.nch = f.nch;
                    it.h0a = f.h0a; it.na = f.na; it.h0b = f.h0b; it.nb = f.nb;
                    it.pad[0] = it.pad[1] = it.pad[2] = 0;
                    big[qslot] = it;
                    queued = true;
                }
            }
            if (queued) {
                cells = 0;
            } else {
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
    const uint32_t incl = wave_inclusive_scan(cells);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t n_slots = (uint32_t)__popcll(__ballot(cells != 0));
    if (cells) s_pref[w][slot] = incl - cells;
    wave_lds_fence();
    const uint32_t first = lane < n_slots ? s_pref[w][lane] : 0xFFFFFFFFu;
    for (uint32_t jb = 0; jb < total; jb += 64u) {
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
}


// ---- the launcher's branch (launch_project, before the single-launch path) ----
        // the two-launch form: one big mesh, nothing culled, no counting
        if (split_items && !multi && !culled && !stats && batch.tris_per_wave[0] == 64u &&
            (size_t)blocks * (3 * kBlock * sizeof(float4) + sizeof(uint32_t)) <= split_bytes) {
            float4 *items = static_cast<float4 *>(split_items);
            uint32_t *cnt = reinterpret_cast<uint32_t *>(items + (size_t)3 * kBlock * blocks);
            const bool timed = ev_start || ev_stop;
            hipEvent_t e0 = ev_start;
            ev_start = nullptr;
            const uint32_t lds_a = lt ? (uint32_t)(2 * (size_t)pp.tb.V * sizeof(float)) : 0u;
            if (lt) {
                if (timed) hipExtLaunchKernelGGL((k_band<true>), grid, dim3(kBlock), lds_a, s, e0, nullptr, 0u, pp, batch, items, cnt);
                else launch_k(k_band<true>, grid, dim3(kBlock), lds_a, s, pp, batch, items, cnt);
                if (timed) hipExtLaunchKernelGGL((k_cells<true>), grid, dim3(kBlock), (uint32_t)lds, s, nullptr, ev_stop, 0u, pp, (const float4 *)items, (const uint32_t *)cnt, best, bq, big_capacity, big_count);
                else launch_k(k_cells<true>, grid, dim3(kBlock), (uint32_t)lds, s, pp, (const float4 *)items, (const uint32_t *)cnt, best, bq, big_capacity, big_count);
            } else {
                if (timed) hipExtLaunchKernelGGL((k_band<false>), grid, dim3(kBlock), 0u, s, e0, nullptr, 0u, pp, batch, items, cnt);
                else launch_k(k_band<false>, grid, dim3(kBlock), 0u, s, pp, batch, items, cnt);
                if (timed) hipExtLaunchKernelGGL((k_cells<false>), grid, dim3(kBlock), 0u, s, nullptr, ev_stop, 0u, pp, (const float4 *)items, (const uint32_t *)cnt, best, bq, big_capacity, big_count);
                else launch_k(k_cells<false>, grid, dim3(kBlock), 0u, s, pp, (const float4 *)items, (const uint32_t *)cnt, best, bq, big_capacity, big_count);
            }
            return;
        }
