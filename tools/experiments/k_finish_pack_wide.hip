// k_finish_pack_wide.hip -- EXPERIMENT (round 5), kept for the record; not compiled into the library.
// Finish + pack of a frame that has the device to itself as ONE launch of wide workgroups (4 x 256 rays each: 512 workgroups
// for the 128 x 4096 raster instead of 2 048), with finish_pack_body's chained prefix.  Dropped into ls_project.hip next to
// k_finish_pack, launched from ls_trace.cpp's single-frame path (grid = ceil(ray blocks / 4)), it passed the parity suites
// (tests/test_gpu_parity.py, test_gpu_cull.py, test_gpu_dropin.py: 140 tests with the path forced, big-footprint queue
// included) and was SLOWER: one frame in flight 25.4 - 25.6 us per frame against 24.3 - 24.9 for k_project_finish + k_pack
// (same box, alternating processes, tools/exp_wide.sh).  With the 2 048-workgroup form (E7.3: 26.6 against 25.4) and the
// 256-workgroup form inside frame graphs (+ 1.0 us) that makes three sizes at which a kernel whose workgroups publish a word
// and read each other's through memory loses against a dependent launch (~4.5 us) on this device.
// (ls_project.hip's types and helpers -- ProjectParams, FinishPackArgs, BigItem, tri_test, kCullChunk -- are used as they are.)
// ------------------------------------------------------------------------------------------
// A frame that has the device to itself (one frame in flight, ls_trace_scene): finish + pack as ONE launch of WIDE
// workgroups -- RPT x 256 consecutive rays each (thread t takes rays t, t + 256, ...: coalesced), so that the full raster's
// 2 048 ray blocks are 512 workgroups.  Same chained prefix as finish_pack_body (a word per workgroup, tagged with an epoch
// that lives in device memory), a quarter of the words to look back over and a quarter of the pollers: at 2 048 workgroups the
// look-back cost more than the second read of the keys it saves (E7.3), and with other frames' kernels on the device a
// chain of workgroups that wait for each other stretches -- hence: alone only.
// ------------------------------------------------------------------------------------------
template <uint32_t RPT>
__global__ __launch_bounds__(kBlock) void k_finish_pack_wide(ProjectParams pp, FinishPackArgs fa)
{
    __shared__ uint32_t s_cnt[RPT][kBlock / 64];
    __shared__ uint32_t s_part[kBlock / 64];
    __shared__ uint32_t s_hint[kBlock / 64];
    __shared__ uint32_t s_box[4][kBlock / 64];   // rank min / max, column min / max per wave (queue gather)
    __shared__ uint16_t s_list[kCullChunk];
    __shared__ uint32_t s_n;
    const SensorTables &tb = pp.tb;
    unsigned long long *__restrict__ best = fa.best;
    const BigItem *__restrict__ big = static_cast<const BigItem *>(fa.big);
    const uint32_t n = tb.V * tb.naz;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t wg = blockIdx.x, n_wg = gridDim.x;
    const uint32_t q0 = wg * RPT * kBlock + threadIdx.x;
    const uint32_t n_big = min(*fa.big_count, fa.big_capacity);
    const uint32_t epoch = *fa.epoch_word;
    // every load of the kernel up front: the rays' keys, then their table entries
    unsigned long long key[RPT];
    uint32_t v[RPT], h[RPT];
    V3 d[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t q = q0 + j * kBlock;
        key[j] = q < n ? best[q] : ~0ull;
    }
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t q = min(q0 + j * kBlock, n - 1u);
        v[j] = q / tb.naz;
        h[j] = tb.az0 + (q - v[j] * tb.naz);
        // LidarDevice.cpp:310-316: d = (sin(theta)cos(phi), sin(theta)sin(phi), cos(theta))
        const float st = tb.sin_theta[v[j]];
        const float2 cs = tb.cs_phi[h[j]];
        d[j] = {st * cs.x, st * cs.y, tb.cos_theta[v[j]]};
    }
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j)
        if (key[j] != ~0ull) best[q0 + j * kBlock] = ~0ull;   // re-armed (a key nobody touched is armed already; key != ~0 implies q < n)
    if (n_big) {   // uniform: the (normally empty) queue of footprints too large for a wave, folded in 256 rays at a time as k_project_finish does
        for (uint32_t j = 0; j < RPT; ++j) {
            const uint32_t q = q0 + j * kBlock;
            uint32_t rank = 0, rmin = 0xFFFFFFFFu, rmax = 0, cmin = 0xFFFFFFFFu, cmax = 0;
            if (q < n) {
                rank = pp.chan_rank[v[j]];
                rmin = rmax = rank;
                cmin = cmax = h[j];
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                rmin = min(rmin, (uint32_t)__shfl_xor(rmin, off)); rmax = max(rmax, (uint32_t)__shfl_xor(rmax, off));
                cmin = min(cmin, (uint32_t)__shfl_xor(cmin, off)); cmax = max(cmax, (uint32_t)__shfl_xor(cmax, off));
            }
            __syncthreads();   // (the previous round's readers of s_box / s_list are done)
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
                        if (h[j] - it.h0a >= it.na && h[j] - it.h0b >= it.nb) continue;
                        float t;
                        if (tri_test(d[j], {it.v0[0], it.v0[1], it.v0[2]}, {it.e1[0], it.e1[1], it.e1[2]}, {it.e2[0], it.e2[1], it.e2[2]},
                                     it.NgC, t)) {
                            const unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | it.gid;
                            key[j] = k2 < key[j] ? k2 : key[j];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    unsigned long long m[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        m[j] = __ballot(key[j] != ~0ull);
        if (lane == 0) s_cnt[j][w] = (uint32_t)__popcll(m[j]);
    }
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) mine += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
    if (threadIdx.x == 0)
        __hip_atomic_store(&fa.status[wg], ((unsigned long long)epoch << 32) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // hits of all workgroups before this one (finish_pack_body's look-back: the workgroups before this one were started earlier)
    uint32_t acc = 0;
    bool stuck = false;
    for (uint32_t i = threadIdx.x; i < wg; i += kBlock) {
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
    if (wg == n_wg - 1u) {
        // every workgroup before this one has published, hence read the queue length and the tag: the counters are re-armed
        // here, the fullest survivor segment goes to the host, the tag steps on (0 is what fresh status words carry: skipped)
        if (threadIdx.x == 0) *fa.n_points = bad ? 0u : base + mine;
        if (threadIdx.x == 0) fa.rearm_big_count[0] = 0u;
        uint32_t fullest = 0;
        for (uint32_t i = threadIdx.x; i < kCullCounters; i += kBlock) {
            fullest = max(fullest, fa.rearm_big_count[kCullCountAt + i * 16u]);
            fa.rearm_big_count[kCullCountAt + i * 16u] = 0u;
        }
        if (fa.cull_hint && fa.rearm_big_count == fa.big_count) {   // (uniform)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) fullest = max(fullest, (uint32_t)__shfl_xor((int)fullest, off));
            if (lane == 0) s_hint[w] = fullest;
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(fa.cull_hint, 1u + max(max(s_hint[0], s_hint[1]), max(s_hint[2], s_hint[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (threadIdx.x == 0) *fa.epoch_word = epoch + 1u ? epoch + 1u : 1u;
    }
    if (bad) {
        if (threadIdx.x == 0) __hip_atomic_fetch_or(fa.device_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    float4 *__restrict__ points = reinterpret_cast<float4 *>(fa.points32);
    uint4 *__restrict__ hits = reinterpret_cast<uint4 *>(fa.hits);
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        // ray order: the 256 rays of round j come before those of round j + 1, wave w's before wave w + 1's
        uint32_t before = base;
        for (uint32_t k = 0; k < w; ++k) before += s_cnt[j][k];
        base += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        if (key[j] == ~0ull) continue;
        const uint32_t dst = before + (uint32_t)__popcll(m[j] & ((1ull << lane) - 1ull));
        const uint32_t gid = (uint32_t)key[j];
        const float t = __uint_as_float((uint32_t)(key[j] >> 32));
        // EmbreeTracer.cpp:341-345: xyz = tfar*dir, intensity 64.0; ring = channel (LidarDeviceKernels.cu:51)
        if (fa.compact == 3u) {
            // LS_OPT_EMIT_POINTS = 0: hit records only
        } else if (fa.compact) {
            points[dst] = make_float4(t * d[j].x, t * d[j].y, t * d[j].z, __int_as_float((int)v[j]));
        } else {
            points[2 * (size_t)dst] = make_float4(t * d[j].x, t * d[j].y, t * d[j].z, 0.0f);
            points[2 * (size_t)dst + 1] = make_float4(64.0f, __int_as_float((int)v[j]), 0.0f, 0.0f);
        }
        uint32_t lo = 0, hi = fa.gt.n;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (fa.gt.tri_first[mid] <= gid) lo = mid; else hi = mid;
        }
        if (hits) hits[dst] = make_uint4(v[j] * tb.H + h[j], fa.gt.geom_ids[lo], (gid - fa.gt.tri_first[lo]) >> fa.gt.prim_shift[lo], __float_as_uint(t));
    }
}

