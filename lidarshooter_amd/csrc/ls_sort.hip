// ls_sort.hip -- stable LSD radix sort of (30-bit Morton key, triangle index) pairs for the (re)build path: the step
// between k_morton / k_mesh_morton and k_leaves / k_permute_indices (OptixTracer::buildAccelStructure's sort is inside
// optixAccelBuild, OptixTracer.cpp:517-571; Embree's inside rtcCommitScene, EmbreeTracer.cpp:290-295).  Hand-written for
// gfx950 (wave64): THREE passes of ten bits (round 3: four of eight), each
//     k_sort_count    one workgroup per tile of 4 096 keys: digit histogram of the tile (LDS atomics) -> counts[tile][digit]
//     k_sort_scan     sixteen digits per workgroup: exclusive scan of every digit's column of counts, and the digit's total
//     k_sort_scatter  the tile again: stable rank of every key inside its tile and digit, scatter
// A wave owns 512 CONSECUTIVE keys of the tile (eight rows of 64) and ranks them against its own row of LDS counters: ten
// ballots split a row's 64 keys by digit ("which lanes hold my digit"), a key's rank is the wave's count so far plus the
// number of peers below it, the lowest peer moves the count on -- LDS operations of one wave execute in order, so the eight
// rows need no workgroup barrier between them (round 3's kernel: three per row of 256 keys); one barrier, one pass over the
// eight waves' counts per digit, and every key knows where it goes.  No LDS atomics in the ranking, no sorting network,
// and the order inside a digit is the input order (stable), which is what makes the Morton order of equal keys -- and with
// it the hierarchy and the BVH test suite's golden structure -- reproducible.
// Between the passes the pairs travel as eight-byte (key, value) words (one scattered store per key instead of two).
// Algorithmic bytes per pass and pair: 4 (count) + 8 read + 8 written (scatter); 60 B per pair for the three passes.
#include "ls_kernels.h"
#include "ls_device.h"

namespace ls {

namespace {

constexpr uint32_t kDigits = kSortDigits;
constexpr uint32_t kSortPasses = 3;                      // 10 + 10 + 10 bits
constexpr uint32_t kSortThreads = 512;                   // eight waves
constexpr uint32_t kSortWaves = kSortThreads / 64;
constexpr uint32_t kSortRows = 8;                        // rows of 64 keys a wave holds in registers
constexpr uint32_t kSortWaveKeys = kSortRows * 64;       // 512
static_assert(kSortTile == kSortWaves * kSortWaveKeys, "4 096 keys per workgroup");
constexpr uint32_t kScanDigits = 16, kScanChunks = 16;   // k_sort_scan: a workgroup is 16 digits x 16 runs of tiles

inline uint32_t sort_tiles(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }

// lanes of the wave that are active and hold the same ten-bit digit as this lane
__device__ __forceinline__ unsigned long long digit_peers(uint32_t d, bool active)
{
    unsigned long long m = __ballot(active);
#pragma unroll
    for (uint32_t k = 0; k < kSortBits; ++k) {
        const bool bit = (d >> k) & 1u;
        const unsigned long long b = __ballot(bit);
        m &= bit ? b : ~b;
    }
    return m;
}

__device__ __forceinline__ uint32_t below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool PAIRS /* keys: the first words of (key, value) pairs */>
__global__ __launch_bounds__(kSortThreads) void k_sort_count(const uint32_t *__restrict__ keys, uint32_t n, uint32_t shift,
                                                            uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_cnt[kDigits];
    const uint32_t tid = threadIdx.x;
    s_cnt[tid] = 0u;
    s_cnt[tid + kSortThreads] = 0u;
    const size_t base = (size_t)blockIdx.x * kSortTile;
    uint32_t key[kSortRows];
#pragma unroll
    for (uint32_t r = 0; r < kSortRows; ++r) {
        const size_t idx = base + (size_t)r * kSortThreads + tid;
        key[r] = idx < n ? keys[PAIRS ? 2 * idx : idx] : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSortRows; ++r)
        if (base + (size_t)r * kSortThreads + tid < n) atomicAdd(&s_cnt[(key[r] >> shift) & (kDigits - 1u)], 1u);
    __syncthreads();
    uint32_t *row = counts + (size_t)blockIdx.x * kDigits;
    row[tid] = s_cnt[tid];
    row[tid + kSortThreads] = s_cnt[tid + kSortThreads];
}

// A workgroup takes sixteen digits; its 256 threads are 16 digits x 16 runs of consecutive tiles.  Every thread sums its run
// of counts[tile][digit] (sixteen neighbouring digits = one 64-byte piece of a tile's row), the sixteen runs of a digit are
// scanned through LDS, and the run is walked again to leave, in place, where inside the digit's span each tile's keys start.
// The digit's total goes to totals[digit].
__global__ __launch_bounds__(kScanDigits *kScanChunks) void k_sort_scan(uint32_t *__restrict__ counts, uint32_t ntiles,
                                                                        uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_sum[kScanChunks][kScanDigits];
    const uint32_t dg = threadIdx.x & (kScanDigits - 1u), ch = threadIdx.x / kScanDigits;
    const uint32_t d = blockIdx.x * kScanDigits + dg;
    const uint32_t per = (ntiles + kScanChunks - 1u) / kScanChunks;
    const uint32_t t0 = ch * per < ntiles ? ch * per : ntiles, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t *col = counts + d;
    uint32_t sum = 0;
    uint32_t t = t0;
    for (; t + 8u <= t1; t += 8u) {   // eight loads in flight
        uint32_t c[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) c[k] = col[(size_t)(t + k) * kDigits];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) sum += c[k];
    }
    for (; t < t1; ++t) sum += col[(size_t)t * kDigits];
    s_sum[ch][dg] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanChunks; ++k) {
        const uint32_t v = s_sum[k][dg];
        run += k < ch ? v : 0u;
        total += v;
    }
    if (ch == 0u) totals[d] = total;
    t = t0;
    for (; t + 8u <= t1; t += 8u) {
        uint32_t c[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) c[k] = col[(size_t)(t + k) * kDigits];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) { col[(size_t)(t + k) * kDigits] = run; run += c[k]; }
    }
    for (; t < t1; ++t) { const uint32_t c = col[(size_t)t * kDigits]; col[(size_t)t * kDigits] = run; run += c; }
}

// IN_PAIRS: keys_in holds (key, value) pairs, vals_in is not read; OUT_PAIRS: keys_out takes pairs, vals_out is not written.
// A pass that writes pairs writes ONE eight-byte word per key where separate arrays take two four-byte words in two places:
// the scattered stores are what a pass costs beyond its 7.5 us (12.0 with them), and they cost per line touched.
template <bool IN_PAIRS, bool OUT_PAIRS>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                              uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n,
                                                              uint32_t shift, const uint32_t *__restrict__ offsets,
                                                              const uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_cnt[kSortWaves][kDigits];   // ranking: keys of digit d seen so far by wave w; then: where wave w's first key of digit d goes
    __shared__ uint32_t s_base[kDigits];              // where this tile's first key of digit d goes
    __shared__ uint32_t s_part[kSortWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const size_t wave_base = (size_t)blockIdx.x * kSortTile + (size_t)w * kSortWaveKeys;
    // the wave's 512 keys and values: sixteen loads in flight before anything else
    uint32_t key[kSortRows], val[kSortRows];
#pragma unroll
    for (uint32_t r = 0; r < kSortRows; ++r) {
        const size_t idx = wave_base + (size_t)r * 64u + lane;
        if (IN_PAIRS) {
            const uint2 kv = idx < n ? reinterpret_cast<const uint2 *>(keys_in)[idx] : make_uint2(0u, 0u);
            key[r] = kv.x;
            val[r] = kv.y;
        } else {
            key[r] = idx < n ? keys_in[idx] : 0u;
            val[r] = !vals_in ? (uint32_t)idx : idx < n ? vals_in[idx] : 0u;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < kSortWaves; ++k) { s_cnt[k][tid] = 0u; s_cnt[k][tid + kSortThreads] = 0u; }
    {
        // where digit d's span starts: exclusive scan of the 1 024 digit totals (every workgroup does its own: 4 KB), thread
        // t takes digits 2t and 2t + 1; plus where this tile's keys start inside the span (k_sort_scan)
        const uint2 c = reinterpret_cast<const uint2 *>(totals)[tid];
        const uint2 o = reinterpret_cast<const uint2 *>(offsets + (size_t)blockIdx.x * kDigits)[tid];
        const uint32_t both = c.x + c.y;
        uint32_t incl = both;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += v;
        }
        if (lane == 63u) s_part[w] = incl;
        __syncthreads();   // (also: the counters are zero)
        uint32_t before = incl - both;
#pragma unroll
        for (uint32_t k = 0; k < kSortWaves; ++k) before += k < w ? s_part[k] : 0u;
        s_base[2u * tid] = before + o.x;
        s_base[2u * tid + 1u] = before + c.x + o.y;
    }
    // rank of every key among the keys of its digit that this wave has seen so far (rows in order, lanes in order = input order)
    uint32_t rank[kSortRows];
    uint32_t *mine = s_cnt[w];
#pragma unroll
    for (uint32_t r = 0; r < kSortRows; ++r) {
        const bool active = wave_base + (size_t)r * 64u + lane < n;
        const uint32_t d = (key[r] >> shift) & (kDigits - 1u);
        const unsigned long long peers = digit_peers(d, active);
        const uint32_t lower = below(peers);
        const uint32_t seen = mine[d];
        rank[r] = seen + lower;
        // the lowest peer moves the wave's count on; the next row's read of it comes later in this wave's LDS queue
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (active && lower == 0u) mine[d] = seen + (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    __syncthreads();
    // digit d: the waves' counts become the waves' starting positions (the waves hold consecutive pieces of the tile)
#pragma unroll
    for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t d = tid + h * kSortThreads;
        uint32_t run = s_base[d];
#pragma unroll
        for (uint32_t k = 0; k < kSortWaves; ++k) {
            const uint32_t c = s_cnt[k][d];
            s_cnt[k][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSortRows; ++r) {
        if (wave_base + (size_t)r * 64u + lane < n) {
            const uint32_t at = mine[(key[r] >> shift) & (kDigits - 1u)] + rank[r];
            if (OUT_PAIRS) {
                reinterpret_cast<uint2 *>(keys_out)[at] = make_uint2(key[r], val[r]);
            } else {
                keys_out[at] = key[r];
                vals_out[at] = val[r];
            }
        }
    }
}

}  // namespace

// temp: two arrays of n (key, value) pairs, then the counts (tiles x 1 024 words) and the digit totals
size_t sort_temp_bytes(uint32_t n)
{
    return ((size_t)4 * n + (size_t)kDigits * sort_tiles(n) + kDigits + 64) * sizeof(uint32_t);
}

uint32_t *sort_first_counts(void *temp, uint32_t n) { return static_cast<uint32_t *>(temp) + (size_t)4 * n; }

// keys_in / vals_in are left as they are; the sorted pairs end in keys_out / vals_out (in -> pairs -> pairs -> out)
// false: the scratch is smaller than sort_temp_bytes(n) -- nothing was launched, keys_out / vals_out are NOT sorted (the
// callers turn that into LS_ERR_OUT_OF_RANGE instead of building on garbage: ADVICE round 3)
bool launch_sort(hipStream_t s, void *temp, size_t temp_bytes, uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_in,
                 uint32_t *vals_out, uint32_t n, bool first_counted)
{
    if (!n) return true;
    if (!temp || temp_bytes < sort_temp_bytes(n)) return false;
    uint32_t *p0 = static_cast<uint32_t *>(temp), *p1 = p0 + (size_t)2 * n, *counts = p1 + (size_t)2 * n;
    const uint32_t ntiles = sort_tiles(n);
    uint32_t *totals = counts + (size_t)kDigits * ntiles;
    const dim3 grid(ntiles), block(kSortThreads), sgrid(kDigits / kScanDigits), sblock(kScanDigits * kScanChunks);
    static_assert(kSortPasses == 3, "separate arrays -> pairs -> pairs -> separate arrays");
    if (!first_counted) hipLaunchKernelGGL(k_sort_count<false>, grid, block, 0, s, keys_in, n, 0u, counts);
    hipLaunchKernelGGL(k_sort_scan, sgrid, sblock, 0, s, counts, ntiles, totals);
    hipLaunchKernelGGL((k_sort_scatter<false, true>), grid, block, 0, s, keys_in, vals_in, p0, nullptr, n, 0u, counts, totals);
    hipLaunchKernelGGL(k_sort_count<true>, grid, block, 0, s, p0, n, kSortBits, counts);
    hipLaunchKernelGGL(k_sort_scan, sgrid, sblock, 0, s, counts, ntiles, totals);
    hipLaunchKernelGGL((k_sort_scatter<true, true>), grid, block, 0, s, p0, nullptr, p1, nullptr, n, kSortBits, counts, totals);
    hipLaunchKernelGGL(k_sort_count<true>, grid, block, 0, s, p1, n, 2u * kSortBits, counts);
    hipLaunchKernelGGL(k_sort_scan, sgrid, sblock, 0, s, counts, ntiles, totals);
    hipLaunchKernelGGL((k_sort_scatter<true, false>), grid, block, 0, s, p1, nullptr, keys_out, vals_out, n, 2u * kSortBits, counts, totals);
    return true;
}

}  // namespace ls
