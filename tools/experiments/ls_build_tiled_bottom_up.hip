// EXPERIMENT, not part of the library (round 3): the hierarchy bottom-up in LDS-sized tiles, one launch
// per level.  Built to parity -- the whole BVH suite passed on it, root at node 0, same tree as k_hierarchy's -- and measured
// on SYN-1M (rocprofv3): level 0 56 us, levels 1 and 2 15.6 us each, tail 10 us = 98 us against the 85 us of k_range_bottom /
// k_range_top / k_hierarchy that it would replace.  A tile is a chain of latencies (item loads, keys, the climb's ~20 levels
// of four LDS round trips each, scan, the look-back over the tiles' status words with agent-scope loads that every waiting
// lane repeats, the root stores): ~15 us per tile at six tiles per CU.  Kept for the record; to build it, add it to KERN_OBJ
// and call launch_build_hierarchy where ls_commit.cpp calls launch_range_tree + launch_hierarchy.
// ls_build.hip -- the hierarchy over Morton-sorted leaves, bottom up in LDS-sized tiles (OptixTracer::buildAccelStructure's
// optixAccelBuild, OptixTracer.cpp:517-571; rtcCommitScene, EmbreeTracer.cpp:290-295).  The radix tree over sorted keys
// is the Cartesian tree of the prefix lengths delta(j) of adjacent keys (Karras 2012; the bottom-up formulation is
// Apetrei 2014): the parent of a subtree over leaves [lo, hi] splits at whichever of the two gaps next to it, lo - 1 or hi,
// has the LONGER common prefix, and that gap's index is the parent's node index.  One kernel per LEVEL:
//
//   level 0      a workgroup takes 256 consecutive leaves (boxes from k_leaves, keys from the sort).  Every leaf's lane
//                climbs: it parks its subtree (box, range, reference) at the parent gap's slot in LDS and marks the gap
//                with an LDS atomic; the first to arrive at a gap ends there, the second reads its sibling, WRITES THE
//                NODE (both child boxes, 64 bytes) and climbs on.  A climb also ends when the parent gap lies on the
//                tile's edge.  What is left when all lanes have ended -- the edge subtrees and the parked ones whose
//                sibling never came, at most two per tree level -- are the tile's ROOTS, written in leaf order to the next
//                level's item list; where in that list comes from a chained prefix over the tiles' root counts (every
//                tile publishes (epoch, count) and adds up the words of the tiles before it: one hop, no serial chain).
//   level 1, 2   the same kernel over the previous level's roots as items (a million leaves leave ~60 000 items after
//                level 0, ~4 000 after level 1 ...); a level that finds a single tile finishes the tree.
//   k_build_tail one workgroup: the levels that are left (a few hundred items), then the root changes places with the node
//                that sat at 0 -- the traversal kernels start there -- and the one reference to that node is redirected.
//
// No range tree (64 MB at a million leaves), no box is read back from global memory except the leaves' own: algorithmic
// bytes per leaf 4 (key) + 32 (leaf box) read, 64 (node) written.
#include "ls_kernels.h"
#include "ls_device.h"

namespace ls {

namespace {

constexpr uint32_t kTile = kBlock;        // items per workgroup, one per lane
constexpr uint32_t kBuildLevels = 8;      // launches per build at most (a level that finds nothing to do ends at once)
constexpr uint32_t kSideL = 1u, kSideR = 2u;

struct BuildItems {      // a level's items (subtrees), in leaf order
    uint32_t *lo, *hi, *ref;   // first / last leaf, child reference (leaf bit | leaf, or node index)
    float4 *box;               // 2 per item: (lo.xyz, -), (hi.xyz, -)
};

struct BuildCtl {        // device words of one build
    uint32_t count[kBuildLevels + 1];   // items of level k ([0] is not used: level 0's items are the leaves)
    uint32_t root;                      // reference of the finished tree's root
    uint32_t parent_of_zero;            // node whose child is (internal) node 0 | side << 31; 0xFFFFFFFF: none (node 0 is the root)
    uint32_t overflow;                  // a level's roots did not fit (cannot happen below 2 x 64 roots per tile; checked by the host in debug builds)
};

__device__ __forceinline__ int gap_delta(const uint32_t *__restrict__ keys, uint32_t g, uint32_t L, long j)
{
    // common prefix of leaves j and j + 1 (Karras: equal keys are told apart by their indices); -1 outside
    if (j < 0 || j + 1 >= (long)L) return -1;
    const uint32_t x = keys[(size_t)j * g] ^ keys[(size_t)(j + 1) * g];
    return x ? __clz(x) : 32 + __clz((uint32_t)j ^ (uint32_t)(j + 1));
}

struct Sub {   // a subtree on its way up
    uint32_t lo, hi, ref;
    float b[6];
    uint32_t l, r;   // its items inside the tile
};

// One tile of one level.  Workgroup-uniform control flow; ends with every lane past its last use of the LDS arrays' contents
// that the next tile overwrites (the caller puts a barrier between tiles).
struct BuildLds {
    int d[kTile + 1];                  // d[k + 1] = delta of the gap between items k and k + 1; [0]: left of the tile
    uint32_t flag[kTile];              // gap k: which sides have arrived
    uint32_t u[2][5][kTile];           // parked subtree per gap and side: lo, hi, ref, l, r
    float b[2][6][kTile];              // its box
    uint32_t start[kTile];             // 1 where a root starts, then the root's rank
    uint32_t wsum[kTile / 64];
};

template <bool LEAVES>
__device__ __forceinline__ void build_tile(BuildLds &lds, const uint32_t *__restrict__ keys, uint32_t g, uint32_t L, uint32_t level, uint32_t m,
                                           uint32_t tile, uint32_t ntiles, const float4 *__restrict__ leaf_boxes, const BuildItems &in,
                                           const BuildItems &out, uint32_t out_capacity, FatNode *__restrict__ nodes, BuildCtl *__restrict__ ctl,
                                           unsigned long long *__restrict__ status, uint32_t epoch)
{
    const uint32_t k = threadIdx.x, lane = k & 63u, w = k >> 6;
    const uint32_t i = tile * kTile + k;
    const uint32_t cnt = min(kTile, m - tile * kTile);   // items of this tile
    const bool valid = k < cnt;
    Sub cur = {0, 0, 0, {0, 0, 0, 0, 0, 0}, k, k};
    if (valid) {
        float4 a, b;
        if (LEAVES) {
            cur.lo = cur.hi = i;
            cur.ref = kLeafBit | i;
            a = leaf_boxes[2 * (size_t)i]; b = leaf_boxes[2 * (size_t)i + 1];
        } else {
            cur.lo = in.lo[i]; cur.hi = in.hi[i]; cur.ref = in.ref[i];
            a = in.box[2 * (size_t)i]; b = in.box[2 * (size_t)i + 1];
        }
        cur.b[0] = a.x; cur.b[1] = a.y; cur.b[2] = a.z; cur.b[3] = b.x; cur.b[4] = b.y; cur.b[5] = b.z;
        lds.d[k + 1] = gap_delta(keys, g, L, (long)cur.hi);           // the gap to the right of item k (-1 behind the last leaf)
        if (k == 0) lds.d[0] = gap_delta(keys, g, L, (long)cur.lo - 1);
    }
    lds.flag[k] = 0u;
    lds.start[k] = 0u;
    __syncthreads();
    // ---- the climb
    bool root_here = false;   // this lane ended holding a root of the tile (its parent gap is on the tile's edge)
    if (valid) {
        for (;;) {
            const int dl = lds.d[cur.l], dr = lds.d[cur.r + 1];
            if (dl < 0 && dr < 0) {   // every leaf: the tree's root (only a level of one tile gets here)
                ctl->root = cur.ref;
                break;
            }
            const bool left = dl > dr;                               // the parent splits at the gap on this side
            const uint32_t gap = left ? cur.l - 1u : cur.r;          // local: between items gap and gap + 1
            if ((left && cur.l == 0u) || (!left && cur.r + 1u == cnt)) { root_here = true; break; }   // the tile's edge
            const uint32_t side = left ? 1u : 0u;                    // slot: 0 = left child of the gap, 1 = right child
            lds.u[side][0][gap] = cur.lo; lds.u[side][1][gap] = cur.hi; lds.u[side][2][gap] = cur.ref;
            lds.u[side][3][gap] = cur.l; lds.u[side][4][gap] = cur.r;
#pragma unroll
            for (int a = 0; a < 6; ++a) lds.b[side][a][gap] = cur.b[a];
            // the slot is in LDS before the mark is (a wave's LDS operations complete in order; the wait is for THEM only: a
            // workgroup-scope fence would also wait for the node stores of the iteration before -- a microsecond per level
            // of the climb, which is what this kernel took at first)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t before = atomicOr(&lds.flag[gap], left ? kSideR : kSideL);
            if (before == 0u) break;                                 // first at the gap: parked
            asm volatile("" ::: "memory");                           // (the sibling's slot is read after the mark: LDS order)
            const uint32_t o = side ^ 1u;
            Sub sib;
            sib.lo = lds.u[o][0][gap]; sib.hi = lds.u[o][1][gap]; sib.ref = lds.u[o][2][gap]; sib.l = lds.u[o][3][gap]; sib.r = lds.u[o][4][gap];
#pragma unroll
            for (int a = 0; a < 6; ++a) sib.b[a] = lds.b[o][a][gap];
            const Sub &lc = left ? sib : cur, &rc = left ? cur : sib;
            const uint32_t node = lc.hi;                             // the gap's global index = its left subtree's last leaf
            float4 *nd = nodes[node].q;
            nd[0] = make_float4(lc.b[0], lc.b[1], lc.b[2], __uint_as_float(lc.ref));
            nd[1] = make_float4(lc.b[3], lc.b[4], lc.b[5], __uint_as_float(rc.ref));
            nd[2] = make_float4(rc.b[0], rc.b[1], rc.b[2], 0.0f);
            nd[3] = make_float4(rc.b[3], rc.b[4], rc.b[5], 0.0f);
            if (lc.ref == 0u) ctl->parent_of_zero = node;                  // (reference 0 without the leaf bit: node 0)
            if (rc.ref == 0u) ctl->parent_of_zero = node | 0x80000000u;
            Sub up;
            up.lo = lc.lo; up.hi = rc.hi; up.ref = node; up.l = lc.l; up.r = rc.r;
#pragma unroll
            for (int a = 0; a < 3; ++a) { up.b[a] = fminf(lc.b[a], rc.b[a]); up.b[3 + a] = fmaxf(lc.b[3 + a], rc.b[3 + a]); }
            cur = up;
        }
    }
    __syncthreads();
    if (ntiles == 1u) {   // nothing is left over: the tree is complete
        if (k == 0) __hip_atomic_store(&ctl->count[level + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- the tile's roots, in leaf order: the lanes that ended at the edge, and the subtrees parked at a gap whose
    //      other side never came (lane k looks after gap k)
    const uint32_t fl = (valid && k + 1u < cnt) ? lds.flag[k] : 0u;
    const bool parked = fl == kSideL || fl == kSideR;
    const uint32_t pside = fl == kSideL ? 0u : 1u;
    if (root_here) lds.start[cur.l] = 1u;
    if (parked) lds.start[lds.u[pside][3][k]] = 1u;
    __syncthreads();
    const uint32_t mine = lds.start[k];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63u) lds.wsum[w] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t q = 0; q < w; ++q) before += lds.wsum[q];
    const uint32_t n_roots = lds.wsum[0] + lds.wsum[1] + lds.wsum[2] + lds.wsum[3];
    lds.start[k] = before;      // rank of the root that starts at item k
    // ---- where the tile's roots go: the counts of the tiles before it (published with the launch's epoch tag; a tile
    //      only ever waits for lower-numbered ones, which the dispatcher started earlier or which this workgroup has done).
    //      Eight words per lane go out together: one memory round trip per 2 048 tiles, not one per tile
    if (k == 0) __hip_atomic_store(&status[tile], ((unsigned long long)epoch << 32) | n_roots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc = 0;
    for (uint32_t base = 0; base < tile; base += 8u * kTile) {
        unsigned long long st[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t t = base + u * kTile + k;
            st[u] = t < tile ? __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t t = base + u * kTile + k;
            while ((uint32_t)(st[u] >> 32) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                st[u] = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc += (uint32_t)st[u];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __syncthreads();          // (wsum has been read by everyone)
    if (lane == 0) lds.wsum[w] = acc;
    __syncthreads();
    const uint32_t offset = lds.wsum[0] + lds.wsum[1] + lds.wsum[2] + lds.wsum[3];
    if (tile == ntiles - 1u && k == 0) __hip_atomic_store(&ctl->count[level + 1], offset + n_roots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto emit = [&](uint32_t at, uint32_t lo, uint32_t hi, uint32_t ref, const float *b) {
        if (at >= out_capacity) { ctl->overflow = 1u; return; }
        out.lo[at] = lo; out.hi[at] = hi; out.ref[at] = ref;
        out.box[2 * (size_t)at] = make_float4(b[0], b[1], b[2], 0.0f);
        out.box[2 * (size_t)at + 1] = make_float4(b[3], b[4], b[5], 0.0f);
    };
    if (root_here) emit(offset + lds.start[cur.l], cur.lo, cur.hi, cur.ref, cur.b);
    if (parked) {
        float b[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) b[a] = lds.b[pside][a][k];
        emit(offset + lds.start[lds.u[pside][3][k]], lds.u[pside][0][k], lds.u[pside][1][k], lds.u[pside][2][k], b);
    }
}

// one level, its tiles dealt round the grid (a workgroup takes its tiles in ascending order: the chained prefix only ever
// waits for lower-numbered tiles); the grid is an estimate, any size is correct
template <bool LEAVES>
__global__ __launch_bounds__(kTile) void k_build_level(const uint32_t *__restrict__ keys, uint32_t g, uint32_t L, uint32_t level,
                                                       const float4 *__restrict__ leaf_boxes, BuildItems in, BuildItems out,
                                                       uint32_t out_capacity, FatNode *__restrict__ nodes, BuildCtl *__restrict__ ctl,
                                                       unsigned long long *__restrict__ status, uint32_t epoch)
{
    __shared__ BuildLds lds;
    const uint32_t m = LEAVES ? L : __hip_atomic_load(&ctl->count[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m <= 1u && !LEAVES) return;                // the tree is finished
    const uint32_t ntiles = (m + kTile - 1) / kTile;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        build_tile<LEAVES>(lds, keys, g, L, level, m, tile, ntiles, leaf_boxes, in, out, out_capacity, nodes, ctl, status, epoch);
        __syncthreads();
    }
}

// the rest, one workgroup: level after level until a single tile finishes the tree (the levels before it have left a few
// hundred items); then the root changes places with whatever node the gap numbering put at 0 -- the traversal kernels
// start there -- and the control words are made ready for the next build
__global__ __launch_bounds__(kTile) void k_build_tail(const uint32_t *__restrict__ keys, uint32_t g, uint32_t L, uint32_t first_level,
                                                      BuildItems it0, BuildItems it1, uint32_t capacity, FatNode *__restrict__ nodes,
                                                      BuildCtl *__restrict__ ctl, unsigned long long *__restrict__ status, uint32_t epoch0,
                                                      uint32_t *__restrict__ device_status)
{
    __shared__ BuildLds lds;
    uint32_t epoch = epoch0;
    for (uint32_t level = first_level; level < kBuildLevels; ++level, ++epoch) {
        const uint32_t m = __hip_atomic_load(&ctl->count[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m <= 1u) break;
        const uint32_t ntiles = (m + kTile - 1) / kTile;
        const BuildItems &in = (level & 1u) ? it0 : it1, &out = (level & 1u) ? it1 : it0;   // level k reads what level k - 1 wrote
        for (uint32_t tile = 0; tile < ntiles; ++tile) {
            build_tile<false>(lds, keys, g, L, level, m, tile, ntiles, nullptr, in, out, capacity, nodes, ctl, status, epoch);
            __syncthreads();
        }
        // this workgroup reads next what it has just written (and read the level before that from the same addresses)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t lane = threadIdx.x;
    const uint32_t r = ctl->root, p = ctl->parent_of_zero, failed = (r & kLeafBit) || ctl->overflow;
    __syncthreads();
    if (lane < 4u) {
        if (failed) {
            // the levels did not finish the tree (more levels than kBuildLevels, or more roots than room: neither can happen
            // below 62 tree levels, see layout()): node 0 becomes a node no ray enters and the host is told at its next wait
            nodes[0].q[lane] = lane == 0 ? make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(kLeafBit))
                             : lane == 1 ? make_float4(-INFINITY, -INFINITY, -INFINITY, __uint_as_float(kLeafBit))
                             : lane == 2 ? make_float4(INFINITY, INFINITY, INFINITY, 0.0f) : make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
            if (lane == 0) __hip_atomic_fetch_or(device_status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (r != 0u) {
            const float4 a = nodes[0].q[lane], b = nodes[r].q[lane];
            nodes[0].q[lane] = b;
            nodes[r].q[lane] = a;
        }
    }
    __syncthreads();
    if (lane == 0) {
        if (!failed && r != 0u && p != 0xFFFFFFFFu) {
            // the one reference to the node that was at 0 now has to say r (its parent may be the root itself, now at 0)
            uint32_t at = p & 0x7FFFFFFFu;
            if (at == r) at = 0u;
            float4 *q = nodes[at].q;
            if (p & 0x80000000u) q[1].w = __uint_as_float(r); else q[0].w = __uint_as_float(r);
        }
        // ready for the next build on this scratch
        for (uint32_t q = 0; q <= kBuildLevels; ++q) ctl->count[q] = 0u;
        ctl->root = kLeafBit;
        ctl->parent_of_zero = 0xFFFFFFFFu;
        ctl->overflow = 0u;
    }
}

// the control words of a fresh (zeroed) scratch
__global__ void k_build_begin(BuildCtl *__restrict__ ctl)
{
    if (threadIdx.x == 0) { ctl->root = kLeafBit; ctl->parent_of_zero = 0xFFFFFFFFu; ctl->overflow = 0u; }
}

struct Layout {
    size_t items[2], status, ctl, total;
    uint32_t capacity;
};
// items of level >= 1: at most 2 x (tree depth <= 62) per tile of 256 -- half of the leaves is room for every input
Layout layout(uint32_t n_leaves)
{
    Layout l;
    l.capacity = n_leaves / 2u + kTile;
    const size_t per = (size_t)l.capacity * (3 * 4 + 32);
    const size_t ntiles = (n_leaves + kTile - 1) / kTile;
    l.items[0] = 0;
    l.items[1] = (per + 255) / 256 * 256;
    l.status = 2 * l.items[1];
    l.ctl = l.status + (ntiles * 8 + 255) / 256 * 256;
    l.total = l.ctl + 256;
    return l;
}

BuildItems items_at(uint8_t *base, uint32_t capacity)
{
    BuildItems it;
    it.box = reinterpret_cast<float4 *>(base);
    it.lo = reinterpret_cast<uint32_t *>(base + (size_t)capacity * 32);
    it.hi = it.lo + capacity;
    it.ref = it.hi + capacity;
    return it;
}

}  // namespace

size_t build_scratch_bytes(uint32_t n_leaves) { return layout(n_leaves).total; }

// after the scratch has been zeroed (allocation, epoch wrap): the control words' initial values
void launch_build_init(hipStream_t s, void *scratch, uint32_t scratch_leaves)
{
    const Layout l = layout(scratch_leaves);
    hipLaunchKernelGGL(k_build_begin, dim3(1), dim3(64), 0, s, reinterpret_cast<BuildCtl *>(static_cast<uint8_t *>(scratch) + l.ctl));
}

// `scratch`: build_scratch_bytes(scratch_leaves) for some scratch_leaves >= n_leaves, zeroed and then launch_build_init'ed
// once when it is allocated (the status words carry epoch tags: *epoch is the caller's counter for this scratch, advanced
// here; it must never repeat a value that is still in the words -- the caller zeroes the scratch again before it wraps)
void launch_build_hierarchy(hipStream_t s, const uint32_t *sorted_keys, uint32_t n_leaves, uint32_t leaf_size, const float4 *leaf_boxes,
                            FatNode *nodes, void *scratch, uint32_t scratch_leaves, uint32_t *epoch, uint32_t *device_status)
{
    if (n_leaves < 2 || n_leaves > scratch_leaves) return;
    // the layout is that of the scratch, not of this build: the status words stay where they are from build to build (a
    // smaller build's words must not land on what a larger one used for items -- leaf numbers look like epoch tags)
    const Layout l = layout(scratch_leaves);
    uint8_t *base = static_cast<uint8_t *>(scratch);
    const BuildItems it0 = items_at(base + l.items[0], l.capacity), it1 = items_at(base + l.items[1], l.capacity);
    unsigned long long *status = reinterpret_cast<unsigned long long *>(base + l.status);
    BuildCtl *ctl = reinterpret_cast<BuildCtl *>(base + l.ctl);
    // level 0 over the leaves, then up to two more levels on grids sized for 32 roots per tile of the level before (a dozen
    // is usual; more only means that a workgroup takes several tiles), then one workgroup for whatever is left
    const uint32_t tiles0 = (n_leaves + kTile - 1) / kTile;
    ++*epoch;
    hipLaunchKernelGGL(k_build_level<true>, dim3(tiles0), dim3(kTile), 0, s, sorted_keys, leaf_size, n_leaves, 0u, leaf_boxes, it1, it0, l.capacity,
                       nodes, ctl, status, *epoch);
    uint32_t level = 1, tiles = tiles0;
    for (; level <= 2u && tiles > 8u; ++level) {
        tiles = (tiles * 32u + kTile - 1) / kTile;
        ++*epoch;
        // (a workgroup that takes a second tile waits for tiles of workgroups that may not have started yet: the grid must
        // fit on the chip at once -- 25 KB of LDS each, two per CU at the very least -- or the waiters could hold every place)
        hipLaunchKernelGGL(k_build_level<false>, dim3(std::min(tiles, 512u)), dim3(kTile), 0, s, sorted_keys, leaf_size, n_leaves, level, leaf_boxes,
                           (level & 1u) ? it0 : it1, (level & 1u) ? it1 : it0, l.capacity, nodes, ctl, status, *epoch);
    }
    hipLaunchKernelGGL(k_build_tail, dim3(1), dim3(kTile), 0, s, sorted_keys, leaf_size, n_leaves, level, it0, it1, l.capacity, nodes, ctl, status,
                       *epoch + 1u, device_status);
    *epoch += kBuildLevels;   // (the tail's levels take one tag each)
}

}  // namespace ls
