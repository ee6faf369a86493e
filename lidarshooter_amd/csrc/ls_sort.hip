// ls_sort.hip -- stable LSD radix sort of (30-bit Morton key, triangle index) pairs for the (re)build path: the step
// between k_morton / k_mesh_morton and k_leaves / k_permute_indices (OptixTracer::buildAccelStructure's sort is inside
// optixAccelBuild, OptixTracer.cpp:517-571; Embree's inside rtcCommitScene, EmbreeTracer.cpp:290-295).  Hand-written for
// gfx950 (wave64): four passes of eight bits, each
//     k_sort_count    one workgroup per tile: digit histogram of the tile -> counts[digit][tile]
//     k_sort_scan     one workgroup per digit: exclusive scan of its row of counts, and the digit's total
//     k_sort_scatter  the tile again: stable rank of every key inside its tile and digit, scatter
// The rank comes from wave ballots: eight ballots split a wave's 64 keys by digit ("which lanes hold my digit"), the lowest
// of those lanes publishes the count, a key's rank is the number of peers below it -- no LDS atomics, no sorting network,
// and the order inside a digit is the input order (stable), which is what makes the Morton order of equal keys -- and with
// it the hierarchy and the BVH test suite's golden structure -- reproducible.
// Algorithmic bytes per pass and pair: 4 (count) + 8 read + 8 written (scatter); 80 B per pair for the four passes.
#include "ls_kernels.h"
#include "ls_device.h"

namespace ls {

namespace {

constexpr uint32_t kDigits = 256;
constexpr uint32_t kSortPasses = 4;   // 8 + 8 + 8 + 6 bits
constexpr uint32_t kSortBatch = 8;    // rows whose loads are issued together (a tile holds a multiple of it)

// rows of 256 keys a tile holds: 8 (2 048 keys) up to 2 M pairs, 32 (8 192 keys) above (a digit's row of counts, which one
// workgroup scans, stays around a thousand entries at 10 M pairs)
inline uint32_t sort_rows(uint32_t n) { return n <= (2u << 20) ? 8u : 32u; }
inline uint32_t sort_tiles(uint32_t n) { const uint32_t t = sort_rows(n) * kBlock; return (n + t - 1) / t; }

// lanes of the wave that are active and hold the same 8-bit digit as this lane
__device__ __forceinline__ unsigned long long digit_peers(uint32_t d, bool active)
{
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
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

__global__ __launch_bounds__(kBlock) void k_sort_count(const uint32_t *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t ntiles,
                                                       uint32_t rows, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_wave[kBlock / 64][kDigits];
    const uint32_t tid = threadIdx.x, w = tid >> 6;
#pragma unroll
    for (uint32_t k = 0; k < kBlock / 64; ++k) s_wave[k][tid] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * rows * kBlock;
    for (uint32_t r0 = 0; r0 < rows; r0 += kSortBatch) {   // kSortBatch rows at a time: their loads go out together
        uint32_t key[kSortBatch];
#pragma unroll
        for (uint32_t r = 0; r < kSortBatch; ++r) {
            const size_t idx = base + (size_t)(r0 + r) * kBlock + tid;
            key[r] = idx < n ? keys[idx] : 0u;
        }
#pragma unroll
        for (uint32_t r = 0; r < kSortBatch; ++r) {
            const bool active = base + (size_t)(r0 + r) * kBlock + tid < n;
            const uint32_t d = (key[r] >> shift) & 0xFFu;
            const unsigned long long peers = digit_peers(d, active);
            // one lane per digit present in the wave adds the wave's count (a wave owns its row of s_wave: no atomics)
            if (active && below(peers) == 0u) s_wave[w][d] += (uint32_t)__popcll(peers);
        }
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBlock / 64; ++k) c += s_wave[k][tid];
    counts[(size_t)tid * ntiles + blockIdx.x] = c;
}

// One workgroup per digit: exclusive scan, in place, of the digit's row counts[d][0 .. ntiles) (-> where inside the digit's
// run each tile's keys start) and the row's total -> totals[d].  Coalesced, 256 entries at a time.
__global__ __launch_bounds__(kBlock) void k_sort_scan(uint32_t *__restrict__ counts, uint32_t ntiles, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_part[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t *row = counts + (size_t)blockIdx.x * ntiles;
    uint32_t run = 0;   // total of the entries before this chunk (uniform)
    for (uint32_t first = 0; first < ntiles; first += kBlock) {
        const uint32_t i = first + tid;
        const uint32_t c = i < ntiles ? row[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += v;
        }
        __syncthreads();   // (the previous chunk's s_part has been read)
        if (lane == 63u) s_part[w] = incl;
        __syncthreads();
        uint32_t before = run;
        for (uint32_t k = 0; k < w; ++k) before += s_part[k];
        if (i < ntiles) row[i] = before + incl - c;
        run += s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
    if (tid == 0) totals[blockIdx.x] = run;
}

__global__ __launch_bounds__(kBlock) void k_sort_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n,
                                                         uint32_t shift, uint32_t ntiles, uint32_t rows, const uint32_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_base[kDigits];               // where this tile's next key of digit d goes
    __shared__ uint32_t s_wave[kBlock / 64][kDigits];  // the current row: keys of digit d in wave w
    __shared__ uint32_t s_part[kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    {
        // where digit d's run starts: exclusive scan of the 256 digit totals (every workgroup does its own: 256 words)
        const uint32_t c = totals[tid];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += v;
        }
        if (lane == 63u) s_part[w] = incl;
        __syncthreads();
        uint32_t before = incl - c;
        for (uint32_t k = 0; k < w; ++k) before += s_part[k];
        s_base[tid] = before + offsets[(size_t)tid * ntiles + blockIdx.x];
    }
#pragma unroll
    for (uint32_t k = 0; k < kBlock / 64; ++k) s_wave[k][tid] = 0u;
    const size_t base = (size_t)blockIdx.x * rows * kBlock;
    for (uint32_t r0 = 0; r0 < rows; r0 += kSortBatch) {   // kSortBatch rows' loads go out together (the barriers below would hold each row's back)
        uint32_t bkey[kSortBatch], bval[kSortBatch];
#pragma unroll
        for (uint32_t r = 0; r < kSortBatch; ++r) {
            const size_t idx = base + (size_t)(r0 + r) * kBlock + tid;
            bkey[r] = idx < n ? keys_in[idx] : 0u;
            bval[r] = idx < n ? vals_in[idx] : 0u;
        }
#pragma unroll
        for (uint32_t rr = 0; rr < kSortBatch; ++rr) {
            const bool active = base + (size_t)(r0 + rr) * kBlock + tid < n;
            const uint32_t key = bkey[rr], val = bval[rr];
            const uint32_t d = (key >> shift) & 0xFFu;
            const unsigned long long peers = digit_peers(d, active);
            const uint32_t rank = below(peers);
            __syncthreads();   // s_base / s_wave are ready (first row: initialised; later rows: advanced and cleared)
            if (active && rank == 0u) s_wave[w][d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (active) {
                uint32_t at = s_base[d] + rank;   // rows before this one, then the waves before this one in the row (input order)
                for (uint32_t k = 0; k < w; ++k) at += s_wave[k][d];
                keys_out[at] = key;
                vals_out[at] = val;
            }
            __syncthreads();
            // thread d moves digit d's base past this row and clears the row's counts (its own four words)
            uint32_t c = 0;
#pragma unroll
            for (uint32_t k = 0; k < kBlock / 64; ++k) { c += s_wave[k][tid]; s_wave[k][tid] = 0u; }
            s_base[tid] += c;
        }
    }
}

}  // namespace

// temp: two ping-pong arrays of n words each for keys and values, then the counts (256 x tiles words) and the digit totals
size_t sort_temp_bytes(uint32_t n)
{
    return ((size_t)2 * n + (size_t)kDigits * sort_tiles(n) + kDigits + 64) * sizeof(uint32_t);
}

// keys_in / vals_in are left as they are; the sorted pairs end in keys_out / vals_out (in -> temp -> out -> temp -> out)
// false: the scratch is smaller than sort_temp_bytes(n) -- nothing was launched, keys_out / vals_out are NOT sorted (the
// callers turn that into LS_ERR_OUT_OF_RANGE instead of building on garbage: ADVICE round 3)
bool launch_sort(hipStream_t s, void *temp, size_t temp_bytes, uint32_t *keys_in, uint32_t *keys_out, uint32_t *vals_in,
                 uint32_t *vals_out, uint32_t n)
{
    if (!n) return true;
    if (!temp || temp_bytes < sort_temp_bytes(n)) return false;
    uint32_t *tk = static_cast<uint32_t *>(temp), *tv = tk + n, *counts = tv + n;
    uint32_t *totals = counts + (size_t)kDigits * sort_tiles(n);
    const uint32_t rows = sort_rows(n), ntiles = sort_tiles(n);
    const uint32_t *ki = keys_in, *vi = vals_in;
    for (uint32_t p = 0; p < kSortPasses; ++p) {
        uint32_t *ko = (p & 1u) ? keys_out : tk, *vo = (p & 1u) ? vals_out : tv;
        const uint32_t shift = 8u * p;
        hipLaunchKernelGGL(k_sort_count, dim3(ntiles), dim3(kBlock), 0, s, ki, n, shift, ntiles, rows, counts);
        hipLaunchKernelGGL(k_sort_scan, dim3(kDigits), dim3(kBlock), 0, s, counts, ntiles, totals);
        hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(kBlock), 0, s, ki, vi, ko, vo, n, shift, ntiles, rows, counts, totals);
        ki = ko;
        vi = vo;
    }
    return true;
}

}  // namespace ls
