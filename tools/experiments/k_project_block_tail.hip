// k_project_block_tail.hip -- EXPERIMENT (round 4), kept for the record; not compiled into the library.
// "Run the expensive half of k_project on dense lanes": all four waves of a workgroup do the streaming half (loads,
// transform, band) and append their band survivors to ONE list in LDS; the first three waves to arrive retire, the wave
// that arrives last runs columns(), staging and the expansion trips for the whole workgroup's survivors (~49 of 256
// triangles: one dense pass; ~276 cells: trips batched four at a time -- one owner search, the four direction loads up
// front, the tests, then the atomics).  VALU wave-instructions per workgroup 2 280 -> ~1 330.
// Parity-green (tests/test_gpu_parity.py subset of tools/exp_run.sh) in all three forms, and slower in all three
// (rocprofv3, SYN-128 x SYN-1M, one frame in flight; k_project 16.0 - 16.8 us on the same boxes):
//     survivors' corners in LDS (48 B per entry, 18 KB per workgroup), trips one at a time       27.5 us
//     survivors as (triangle, band) only, corners gathered again by the finishing wave (9 KB)     31.2 us
//     corners in LDS, trips batched four at a time (this file)                                    27.0 us
// Ablation of this file's form: streaming half alone 8.7 us; + columns / staging of the finishing waves 13.3 us; + trips
// 27.0 us.  Why: a workgroup's lifetime is its critical path, and the finishing wave's path is ~900 mostly dependent
// instructions in ONE wave (a dependent instruction every ~11.5 cycles with few waves on the SIMD, an LDS round trip
// per step of the owner search, a global load and an atomic per batch): ~8 us against the ~3 us the same work takes
// as four waves side by side; the kernel is (workgroups per CU / resident workgroups) x workgroup lifetime, and the
// retired waves' slots cannot be used while the workgroup's LDS is held.  The instruction count went down by 40 %;
// the kernel is not bound by the instruction count of a workgroup but by the length of its longest chain.
// (ls_project.hip's helpers -- band, columns, foot_cell, tri_test, wave_inclusive_scan, lanes_below -- are used as they are.)
constexpr uint32_t kTailTrips = 4;
struct TailLds {
    float v[9][kBlock];          // transformed corners of the band survivors (v0 xyz, v1 xyz, v2 xyz)
    uint32_t gid[kBlock];
    uint32_t band[kBlock];       // i0 | nch << 16
    float tri[10][64];           // staging of the finishing wave: v0, e1, e2, NgC ...
    uint32_t meta[6][64];        // ... gid, i0, h0a, na, h0b, nb
    uint32_t pref[64];
    uint8_t flag[64 * kTailTrips];
    uint32_t count, arrived;
};

template <bool COUNT, bool LDS_TABLES, bool MULTI>
__global__ __launch_bounds__(kBlock) void k_project_bt(ProjectParams pp, GeomBatch batch, unsigned long long *__restrict__ best,
                                                       BigItem *__restrict__ big, uint32_t big_capacity, uint32_t *__restrict__ big_count,
                                                       unsigned long long *__restrict__ stats)
{
    __shared__ TailLds lds;
    extern __shared__ float s_chan[];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    // ... workgroup -> geometry, triangle of this lane (spread runs), index load, vertex gather, table staging: as k_project ...
    // __syncthreads();   // tables staged, lds.count = lds.arrived = 0: the only workgroup barrier
    // ---- streaming half: transform, sector test, band(); survivors go to the workgroup's list
    //     const unsigned long long mask = __ballot(nch != 0);
    //     if (mask) { base = readfirstlane(lane == 0 ? atomicAdd(&lds.count, popc(mask)) : 0); at = base + lanes_below(mask);
    //                 lds.v[0..8][at] = v0, v1, v2; lds.gid[at] = gid; lds.band[at] = i0 | nch << 16; }
    // ---- arrival: __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); seen = atomicAdd(&lds.arrived, 1) (lane 0, readfirstlane);
    //     if (seen != kBlock / 64 - 1) return;                 // three of four waves retire here
    //     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- the finishing wave, 64 survivors to a pass: columns(), e1 / e2 / NgC, big-footprint queue, staging packed to the front,
    //     wave_inclusive_scan of the cell counts, then the trips kTailTrips at a time:
    //         flags over the batch's 256 cells, one ballot per trip -> owner; foot_cell -> (channel, column); cs_phi loads for
    //         all four trips; the four tests; the atomics
}
