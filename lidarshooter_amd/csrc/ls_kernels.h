// ls_kernels.h -- launch wrappers of the gfx950 kernels (ls_kernels.hip) used by ls_tracer.cpp.
// Internal to the library; the public surface is include/lidarshooter_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ls {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr int kMaxRangeLevels = 32;
constexpr int kMaxGeoms = 1024;

// 32-byte BVH node slot; see DESIGN.md "BVH layout".
struct alignas(16) Node {
    float lo[3];
    uint32_t left;  // internal (odd slot): slot of the first child; leaf (even slot): record count
    float hi[3];
    uint32_t skip;  // next slot in depth-first order once this subtree is done / culled
};
static_assert(sizeof(Node) == 32, "node slot must be 32 bytes");

// 48-byte triangle record, in Morton-sorted order (leaf k owns records [k*g, k*g+count)).
struct alignas(16) TriRecord {
    float v0[3];
    uint32_t gid;   // global triangle id: orders triangles by (geomID, primID)
    float e1[3];    // v0 - v1
    float NgC;      // dot(cross(e2,e1), v0): the ray-independent numerator (origin is 0)
    float e2[3];    // v2 - v0
    uint32_t pad;
};
static_assert(sizeof(TriRecord) == 48, "triangle record must be 48 bytes");

// Aligned-range tree over the leaf boxes: level 0 = the leaf node slots themselves, level l >= 1
// at boxes[offset[l] .. offset[l]+count[l]) where entry j bounds leaves [j<<l, (j+1)<<l).
struct RangeTree {
    uint32_t levels;                      // number of levels including level 0
    uint32_t count[kMaxRangeLevels];
    uint32_t offset[kMaxRangeLevels];     // in box entries (2 x float4 each); offset[0] unused
};

struct SensorTables {
    const float *sin_theta;  // [V]
    const float *cos_theta;  // [V]
    const float *sin_phi;    // [H]
    const float *cos_phi;    // [H]
    uint32_t V, H;
    uint32_t az0, naz;       // shard: azimuth columns [az0, az0+naz)
    uint32_t n_az_blocks;    // ceil(naz / 64)
};

struct GeomTable {
    uint32_t n;
    const uint32_t *tri_first;  // device, [n+1] ascending global triangle id offsets
    const uint32_t *geom_ids;   // device, [n]
};

// ---- build ---------------------------------------------------------------------------------
void launch_transform(hipStream_t s, const void *raw, uint32_t stride, uint32_t n, const float *affine12,
                      const float *rinv9, const float *t3, float *out_xyz, uint32_t *d_maxabs_bits);
void launch_rebase(hipStream_t s, const uint32_t *idx, uint32_t n_idx, uint32_t vbase, uint32_t *out);
void launch_morton(hipStream_t s, const float *verts, const uint32_t *tris, uint32_t ntris,
                   const uint32_t *d_maxabs_bits, uint32_t *keys, uint32_t *vals);
size_t sort_temp_bytes(uint32_t n);
void launch_sort(hipStream_t s, void *temp, size_t temp_bytes, uint32_t *keys_in, uint32_t *keys_out,
                 uint32_t *vals_in, uint32_t *vals_out, uint32_t n);
void launch_leaves(hipStream_t s, const float *verts, const uint32_t *tris, const uint32_t *sorted_vals,
                   uint32_t ntris, uint32_t leaf_size, TriRecord *records, Node *nodes);
void launch_range_tree(hipStream_t s, const Node *nodes, const RangeTree &rt, float4 *boxes);
void launch_hierarchy(hipStream_t s, const uint32_t *sorted_keys, uint32_t nleaves, uint32_t leaf_size,
                      const RangeTree &rt, const float4 *boxes, Node *nodes);

// ---- trace ---------------------------------------------------------------------------------
void launch_trace(hipStream_t s, const SensorTables &tb, const Node *nodes, const TriRecord *records,
                  uint32_t nslots, uint32_t leaf_size, float *t_out, uint32_t *gid_out, uint32_t *row_counts,
                  unsigned long long *visit_counts /* nullptr = do not count */);
void launch_pack(hipStream_t s, const SensorTables &tb, const float *t, const uint32_t *gid,
                 const uint32_t *row_counts, const GeomTable &gt, uint8_t *points32, void *hits,
                 uint32_t *n_points);
void launch_raygen(hipStream_t s, const SensorTables &tb, float *dx, float *dy, float *dz);
void launch_bruteforce(hipStream_t s, const SensorTables &tb, const float *verts, const uint32_t *tris,
                       uint32_t ntris, float *t_out, uint32_t *gid_out);

}  // namespace ls
