// ls_kernels.h -- launch wrappers of the gfx950 kernels (ls_kernels.hip) used by ls_tracer.cpp.
// Internal to the library; the public surface is include/lidarshooter_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ls {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kLeafBit = 0x80000000u;  // child reference: bit 31 set = leaf index, else node index
constexpr uint32_t kTreeletBit = 0x40000000u;   // instanced trace, one geometry: node index inside the LDS-staged top of the hierarchy
constexpr uint32_t kTreeletNodes = 511;         // 32 KB of 64-byte nodes (nine levels of a balanced tree), next to the 32 KB of stacks
constexpr int kMaxRangeLevels = 32;
constexpr int kMaxGeoms = 1024;
constexpr int kQueues = 8;           // one ray queue per XCD
constexpr int kStackLds = 32;        // per-lane traversal stack entries kept in LDS
constexpr int kStackSpill = 64;      // further entries in global memory (tree depth <= 62 by construction: the binary walk holds at most 62
                                     // pending references, the four-wide walk at most 3 per two levels: 93)

// 64-byte BVH2 node: both child boxes + both child references (DESIGN.md "BVH layout").
//   q[0] = (L.lo.x, L.lo.y, L.lo.z, bits(left ref))    q[1] = (L.hi.x, L.hi.y, L.hi.z, bits(right ref))
//   q[2] = (R.lo.x, R.lo.y, R.lo.z, first leaf)        q[3] = (R.hi.x, R.hi.y, R.hi.z, last leaf)   (the node's leaf range:
//   the traversal does not read it, a refit -- k_refit_nodes -- does instead of searching)
// Node i is Karras' internal node i; the left child comes first in Morton order = front to back.
struct alignas(64) FatNode {
    float4 q[4];
};
static_assert(sizeof(FatNode) == 64, "node must be 64 bytes");

// 128-byte four-wide node (k_widen, k_trace_inst<WIDE>): slot c = 0 .. 3 is a grandchild of the binary node of the same index (or a
// child that is a leaf): q[c] = (lo.x, lo.y, lo.z, bits(reference)), q[4 + c] = (hi.x, hi.y, hi.z, 0); an empty slot's reference is
// kInvalid.  References are those of the binary nodes (leaf bit, node index), so that both walks share leaves and records.
struct alignas(128) WideNode {
    float4 q[8];
};
static_assert(sizeof(WideNode) == 128, "wide node must be 128 bytes");

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

// Aligned-range tree over the leaf boxes: entry j of level l (at boxes[offset[l]+j], 2 x float4 =
// lo,hi) bounds leaves [j<<l, (j+1)<<l).  Level 0 = the leaf boxes themselves.
struct RangeTree {
    uint32_t levels;
    uint32_t count[kMaxRangeLevels];
    uint32_t offset[kMaxRangeLevels];  // in box entries
};

struct SensorTables {
    const float *sin_theta;  // [V]
    const float *cos_theta;  // [V]
    const float *sin_phi;    // [H]
    const float *cos_phi;    // [H]
    const float2 *cs_phi;    // [H] (cos_phi, sin_phi) interleaved: one 8-byte load per ray
    uint32_t V, H;
    uint32_t az0, naz;       // shard: azimuth columns [az0, az0+naz)
};

struct GeomTable {
    uint32_t n;
    const uint32_t *tri_first;  // device, [n+1] ascending global triangle id offsets
    const uint32_t *geom_ids;   // device, [n]
    const uint32_t *prim_shift; // device, [n]: 1 for quad geometries (two triangles per element, primID = triangle / 2), else 0
};

// Ray queues of the persistent trace kernel: the shard's azimuth columns are cut into kQueues
// sectors (one per XCD); queue x enumerates its sector channel by channel.  heads live in device
// memory, one per 64-byte line.
struct RayQueues {
    uint32_t *heads;               // device, kQueues * 16 words, zeroed before every launch
    uint32_t chan_mul;             // channels are visited in the order (j * chan_mul) % V (coprime with V)
    uint32_t refill_min;           // idle lanes of a wave that trigger a refill
    const uint32_t *chan_order;    // position in ascending elevation order -> channel (the projection engine's chan_perm); visited from the top ring down
    uint32_t leaf_wait;            // lanes of a wave that must be standing at a leaf before the wave runs its leaf tests (0: every trip)
};

// mesh transform A (row-major 3x4), sensor rotation inverse and sensor translation
struct Affine {
    float a[12];
    float rinv[9];
    float t[3];
};

// Sensor-space projection engine (ls_project.hip)
struct ProjectParams {
    SensorTables tb;
    const float *chan_tan_up;      // [V] tan(elevation + margin) of the channels, ascending elevation
    const float *chan_tan_dn;      // [V] tan(elevation - margin), same order
    const uint32_t *chan_perm;     // [V] position in that order -> channel index
    const uint32_t *chan_rank;     // [V] channel index -> position in that order (inverse of chan_perm)
    // k_cull's channel query as a table look-up: chan_lut[b] = first position i with chan_tan_up[i] >= lut_t0 + b / lut_scale,
    // kCullLutBuckets entries; lut_ok = the host verified that no two consecutive buckets hold more than two channels, so
    // that a look-up (started one bucket early for rounding) and two compare-and-step reach the first channel at or
    // above any value.  Sensors that fail the check use the binary search.
    const uint16_t *chan_lut;
    float lut_t0, lut_scale;
    int lut_ok;
    float begin_deg, step_deg;     // azimuth of column h = begin + step*h (LidarDevice.cpp:306)
    float inv_step_deg, inv_period; // 1/step and |step|/360 (0 when step is 0)
    float margin_deg;              // angular slack of the footprint bounds
    int sector_on;                 // the shard covers less than 180 degrees of azimuth: sec_a / sec_b are valid
    float sec_a[2], sec_b[2];      // unit vectors (cos, sin) of the padded sector's first / second boundary, counter-clockwise
    uint32_t big_cells;            // footprints above this many cells go to the gather queue
    int debug;                     // diagnostic: 1 = stop after the vertex loads, 3 = after the band test, 2 = after the footprints
    int spread;                    // big meshes: a wave takes its triangles in runs spread over the whole mesh (balances the cells per
                                   // wave: shorter kernel) instead of one contiguous run (fewer cache lines: better with frames overlapping)
    uint32_t xcd_remap;            // one geometry, no culling: workgroup -> triangles so that each XCD streams one contiguous eighth
    int cols_lds;                  // the shard is at most kColsLdsMax columns: k_project keeps their (cos, sin) in LDS (needs V <= 2048: the channel tables there too)
    int cull_deal;                 // group culling: a segment's survivors are dealt to its waves at a stride (azimuth shards), not taken in runs
};

// one geometry as uploaded (xform = 1: vertices still need A / Rinv / t) or the committed scene
struct GeomSource {
    const uint8_t *verts;
    uint32_t stride;
    const uint32_t *idx;           // 3 * ntris vertex indices into verts
    uint32_t ntris;
    uint32_t gid_first;            // global triangle id of triangle 0
    int xform;                     // 0: verts already in the sensor frame, 1: full transform, 2: A is the identity
    Affine m;
    // group culling (meshes large enough for 64 triangles per wave): idx is then the library's Morton-ordered
    // copy of the indices, perm[k] the original triangle of sorted position k, boxes[2g], boxes[2g+1] the
    // mesh-space sheared-box bound of sorted triangles [kCullGroup g, kCullGroup (g+1)).  nullptr: no culling.
    const uint32_t *perm;
    const float4 *boxes;
    // corners[3k .. 3k+2] = the three corners of sorted triangle k as uploaded, 16 bytes each, perm[k] in the first one's
    // fourth word (k_corners, with the bounds): a surviving group's 4 triangles are 192 contiguous bytes, one memory round
    // trip from the survivor list instead of the index -> vertex chain's two
    const float4 *corners;
};

// the geometries of one k_project launch (kernel argument: no upload)
constexpr int kGeomsPerLaunch = 16;
constexpr uint32_t kColsLdsMax = 1024;   // columns of a shard whose directions k_project stages in LDS (8 KB)
struct GeomBatch {
    uint32_t n;
    uint32_t block_first[kGeomsPerLaunch + 1];   // first workgroup of geometry i; [n] = grid size
    uint32_t tris_per_wave[kGeomsPerLaunch];     // 64, or less for a small mesh
    uint32_t list_first[kGeomsPerLaunch + 1];    // group culling: first entry of geometry i in the survivor list
    uint32_t cull_first[kGeomsPerLaunch + 1];    // group culling: first k_cull workgroup of geometry i; [n] = k_cull's grid
    uint32_t seg_cap[kGeomsPerLaunch];           // group culling: entries of each of geometry i's kCullSegs list segments
    uint32_t seg_blocks[kGeomsPerLaunch];        // group culling: k_project workgroups per list segment in this launch
    uint32_t cull_rounds;                        // group culling: k_cull's groups per workgroup / 256
    GeomSource g[kGeomsPerLaunch];
};

// group culling (ls_project.hip): sorted triangles are bounded and culled in groups of kCullGroup, and those in blocks
// of kCullBlockGroups groups (a coarse bound, tested first).  A geometry's survivors are collected in kCullSegs list
// segments, each with a counter of its own on its own 64-byte line: one hot address sustains ~90 atomics per
// microsecond on MI355X, and the 2 441 workgroups of k_cull at 10 M triangles spent 27 of their 36 us queueing on a
// single counter.  A frame's counter slot: word 0 = length of the big-footprint queue; counter (i, s) of geometry i,
// segment s at word kCullCountAt + (i * kCullSegs + s) * 16.
#ifndef LS_CULL_GROUP
#define LS_CULL_GROUP 4
#endif
constexpr uint32_t kCullGroup = LS_CULL_GROUP;
constexpr uint32_t kCullBlockGroups = 64;
constexpr uint32_t kCullSegs = 32;
constexpr uint32_t kCullLutBuckets = 2048;
constexpr uint32_t kCullCountAt = 16;
constexpr uint32_t kCullCounters = kGeomsPerLaunch * kCullSegs;
constexpr uint32_t kCounterSlotWords = kCullCountAt + kCullCounters * 16;
static_assert(64 % LS_CULL_GROUP == 0, "a wave takes a whole number of groups");

// ---- build ---------------------------------------------------------------------------------
void launch_transform(hipStream_t s, const void *raw, uint32_t stride, uint32_t n, const float *affine12,
                      const float *rinv9, const float *t3, float *out_xyz, uint32_t *d_maxabs_bits);
void launch_rebase(hipStream_t s, const uint32_t *idx, uint32_t n_idx, uint32_t vbase, uint32_t *out);
// *d_max = the largest of idx[0 .. n) (0 for n = 0), stream-ordered: commitScene reads it back and refuses a geometry whose
// triangles name vertices it does not have
void launch_index_max(hipStream_t s, const uint32_t *idx, uint32_t n, uint32_t *d_max);
// RTC_GEOMETRY_TYPE_QUAD (EmbreeTracer.cpp:179-198): quad (v0,v1,v2,v3) -> triangles (v0,v1,v3), (v2,v3,v1), Embree's split
void launch_quads_to_triangles(hipStream_t s, const uint32_t *quad_idx, uint32_t n_quads, uint32_t *tri_idx);
void launch_morton(hipStream_t s, const float *verts, const uint32_t *tris, uint32_t ntris,
                   const uint32_t *d_maxabs_bits, uint32_t *keys, uint32_t *vals, uint32_t *first_counts = nullptr);
// the radix sort of the build (ls_sort.hip): tiles of kSortTile keys, digits of ten bits
constexpr uint32_t kSortTile = 4096, kSortBits = 10, kSortDigits = 1u << kSortBits;
size_t sort_temp_bytes(uint32_t n);
uint32_t *sort_first_counts(void *temp, uint32_t n);   // [tiles][kSortDigits]: the first pass's tile histograms (digit = key & 1023)
// first_counted: sort_first_counts(temp, n) is already filled (k_morton did it); vals_in null: the values are 0 .. n - 1
bool launch_sort(hipStream_t s, void *temp, size_t temp_bytes, uint32_t *keys_in, uint32_t *keys_out,
                 uint32_t *vals_in, uint32_t *vals_out, uint32_t n, bool first_counted = false);
// mesh_records: the records hold the three corners as given (v0, sorted id | v1, 0 | v2, 0) instead of v0 / e1 / e2 / NgC
// of the transformed triangle -- the per-geometry hierarchies of the instanced mode, built once in mesh space
void launch_leaves_tree(hipStream_t s, const float *verts, const uint32_t *tris, const uint32_t *sorted_vals, uint32_t ntris,
                        uint32_t leaf_size, TriRecord *records, const RangeTree &rt, float4 *boxes, bool mesh_records);
void launch_hierarchy(hipStream_t s, const uint32_t *sorted_keys, uint32_t nleaves, uint32_t leaf_size,
                      const RangeTree &rt, const float4 *boxes, FatNode *nodes);
void launch_refit_nodes(hipStream_t s, uint32_t nleaves, const RangeTree &rt, const float4 *boxes, FatNode *nodes);

// ---- trace ---------------------------------------------------------------------------------
uint32_t trace_grid_blocks(int device);  // persistent grid: resident blocks of the device
size_t trace_spill_bytes(uint32_t grid_blocks);
void launch_trace(hipStream_t s, uint32_t grid_blocks, const SensorTables &tb, const RayQueues &rq,
                  const FatNode *nodes, const TriRecord *records, uint32_t nleaves, uint32_t leaf_size,
                  uint32_t ntris, float *t_out, uint32_t *gid_out, uint32_t *spill,
                  unsigned long long *visit_counts /* nullptr = do not count */);
// Instanced mode of the BVH engine (LS_OPT_BVH_INSTANCED): one hierarchy per geometry, built ONCE in mesh space; a frame
// carries every ray into each geometry's mesh space (origin o, linear part minv of the inverse of mesh -> sensor, so
// that the ray parameter t is the sensor-space one), walks that hierarchy with boxes widened by eps, and tests a leaf's
// triangles exactly as the other paths do: corners through the frame's transform `m`, then the Embree test against the
// sensor-space table direction.  A pose change -- the sensor's or a mesh's -- costs no build and no refit.
struct InstGeom {
    uint32_t node_first, rec_first;   // this geometry's nodes / records inside the shared arrays
    uint32_t n_leaves;                // 0: nothing to trace
    uint32_t n_tris;                  // its last leaf may be short
    uint32_t gid_first;               // global triangle id of its triangle 0
    int xform;                        // 1: full transform, 2: A is the identity (as GeomSource::xform)
    float o[3];                       // ray origin (the sensor) in mesh space
    float eps;                        // widening of every box (covers the rounding of o and of minv * d)
    float minv[9];                    // sensor-space direction -> mesh-space direction (row-major)
    Affine m;                         // mesh -> sensor, as everywhere else
};
struct InstBatch {
    uint32_t n;
    InstGeom g[kGeomsPerLaunch];
};
// treelet (nullable; used when the scene is one geometry): kTreeletNodes nodes, the top of that geometry's hierarchy in
// breadth-first order with the references among them rewritten to kTreeletBit | slot (launch_treelet); every block of
// the trace grid stages it in LDS once and walks the first levels of every ray there
void launch_treelet(hipStream_t s, const FatNode *nodes, uint32_t n_leaves, FatNode *treelet);
// wide: the four-wide nodes of the same hierarchies (launch_widen; same indexing as `nodes`), or nullptr for the binary walk
void launch_trace_instanced(hipStream_t s, uint32_t grid_blocks, const SensorTables &tb, const RayQueues &rq, const InstBatch &batch,
                            const FatNode *nodes, const WideNode *wide, const TriRecord *records, uint32_t leaf_size, const FatNode *treelet, float *t_out,
                            uint32_t *gid_out, uint32_t *spill, unsigned long long *visit_counts /* nullptr = do not count */);
// the four-wide twins of a hierarchy's n_leaves - 1 binary nodes (after every build or refit of it)
void launch_widen(hipStream_t s, const FatNode *nodes, uint32_t n_leaves, WideNode *wide);
void launch_rowcount(hipStream_t s, const uint32_t *gid, uint32_t nrays, uint32_t *row_counts, uint32_t *queue_heads = nullptr);   // queue_heads: zeroed for the next k_trace
// Progress of a synchronous frame whose compact points go straight to pinned host memory (ls_trace_scene_begin /
// ls_trace_scene_expand): the device publishes, with system-scope release, (1) the frame's hit count as the pack pass
// starts (its first workgroup adds up the finish pass's block counts), together with the number of points the ray blocks
// before `split` hold -- the caller sizes its cloud while the points are being packed --, then, in stream order behind
// each of the pack pass's two launches (ray blocks before / from `split`), (2) "the first part has arrived" and (3) "all
// have": the host expands the first part of the cloud while the second is still crossing PCIe.  The split that would have halved the POINTS comes back too (the
// upper rings of a LiDAR mostly see sky): the next frame uses it.  One 64-byte line per group of words.
struct HostProgress {
    uint32_t total, n_first, even_split, total_epoch, pad0[12];   // even_split: the ray block at which the points halve
    uint32_t half_epoch, pad1[15];
    uint32_t all_epoch, pad2[15];
};
struct ProgressArgs {
    HostProgress *host;   // pinned host memory; nullptr: no progress reporting
    uint32_t epoch;       // this frame's tag (never 0)
    uint32_t split;       // the pack pass's second launch starts at this ray block (the first one packs n_first points)
    uint32_t *cull_hint;  // pinned host word (nullable): 1 + the fullest survivor-list segment of this frame's k_cull, written by the
                          // workgroup that re-arms the counters -- what sizes the NEXT frames' k_project<CULLED> grid
};

// compact != 0: points are written as 16-byte records (x, y, z, ring) instead of the 32-byte PointCloud2 layout
void launch_pack(hipStream_t s, const SensorTables &tb, float *t, uint32_t *gid, const uint32_t *block_counts,
                 const GeomTable &gt, uint8_t *points32, void *hits, uint32_t *n_points, uint32_t compact = 0);
// projection engine: pack straight from the closest-hit keys (also re-arms keys, counters, queue)
void launch_pack_keys(hipStream_t s, const SensorTables &tb, unsigned long long *keys, float *t, uint32_t *gid,
                      const uint32_t *block_counts, uint32_t *next_block_counts, uint32_t *big_count,
                      const GeomTable &gt, uint8_t *points32, void *hits, uint32_t *n_points, uint32_t compact = 0,
                      const ProgressArgs *progress = nullptr, uint32_t rays_per_lane = 1);
// projection engine: per-geometry streaming kernel, big-footprint kernel, resolve (+ row counts)
size_t project_big_item_bytes();
void launch_project_init(hipStream_t s, const ProjectParams &pp, unsigned long long *best, uint32_t *big_count,
                         uint32_t *block_counts2 /* two frame-parity arrays of ceil(rays/256) words */);
// LS_OPT_PIPELINE: finish + pack of one frame as one set of workgroups (alone, or riding in the launch
// of the next frame's k_project)
struct FinishPackArgs {
    unsigned long long *best;       // the frame's keys (re-armed here)
    const void *big;                // its big-footprint queue
    uint32_t big_capacity;
    const uint32_t *big_count;      // its queue counter
    uint32_t *rearm_big_count;      // the counter of the frame after the next: set to 0
    unsigned long long *status;     // per workgroup (epoch << 32) | hits, for the chained prefix
    uint32_t epoch;                 // tag this frame's workgroups wait for
    uint32_t *epoch_word;           // non-null: the tag lives in device memory instead (read at the start of every workgroup, stepped by
                                    // the last one when all have published) -- the arguments of a frame are then the same every
                                    // frame, and a captured frame graph needs no patch for it (three-stream mode, small shards)
    uint32_t publish_epoch;         // tag they publish (== epoch; LS_OPT_DEBUG_FAULT publishes another one)
    uint32_t spin_limit;            // polls before a waiting workgroup gives up and raises device_status
    uint32_t *device_status;        // sticky status word in pinned host memory (bit 0: a chained prefix gave up)
    GeomTable gt;
    uint8_t *points32;
    void *hits;
    uint32_t *n_points;
    uint32_t n_blocks;              // ceil(rays / 256)
    uint32_t compact;               // points as 16-byte (x, y, z, ring) records (LS_OPT_HOST_OUTPUT = 2)
    uint32_t *cull_hint;            // as ProgressArgs::cull_hint
};
// cull_list (nullable): room for project_cull_entries() words; the groups that survive k_cull are appended there per
// geometry (counts in big_count[kCullCountAt + i], re-armed wherever big_count[0] is) and the waves of the culled
// geometries' k_project launch take 64 / kCullGroup of them each.  Geometries without bounds go in a launch of their own.
void launch_project(hipStream_t s, const ProjectParams &pp, const GeomSource *srcs, uint32_t n_srcs, unsigned long long *best,
                    void *big, uint32_t big_capacity, uint32_t *big_count, unsigned long long *stats,
                    const FinishPackArgs *rider = nullptr, uint32_t *cull_list = nullptr, hipEvent_t ev_start = nullptr,
                    hipEvent_t ev_stop = nullptr,    // ev_*: ride on the k_project dispatch (its own begin / end timestamps)
                    uint32_t survivors_hint = 0);    // 1 + the fullest survivor segment of a recent frame (0: unknown): sizes the culled grid
uint32_t project_tris_per_wave(uint32_t ntris);   // 64 for big meshes, fewer for small ones (more waves than ntris / 64)
uint32_t project_cull_entries(const GeomSource *srcs, uint32_t n_srcs, bool sector);   // survivor-list words for the geometries with bounds (0: none, or too many for one launch)
// one-off per topology: Morton order of the triangles (centroids in mesh space) -> perm (sorted position -> triangle),
// idx_sorted; scratch: keys_a/keys_b/vals_a (ntris words each), aabb (6 words), sort temp
bool launch_mesh_order(hipStream_t s, const uint8_t *verts, uint32_t stride, uint32_t nverts, const uint32_t *idx, uint32_t ntris,
                       uint32_t *aabb6, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, void *sort_temp, size_t sort_temp_bytes,
                       uint32_t *perm, uint32_t *idx_sorted);
// per vertex upload: mesh-space sheared-box bound of every kCullGroup sorted triangles (2 float4 per group), followed
// in the same array by the bound of every kCullBlockGroups groups (project_box_entries() float4 pairs in all)
void launch_group_bounds(hipStream_t s, const uint8_t *verts, uint32_t stride, const uint32_t *idx_sorted, uint32_t ntris, float4 *boxes);
void launch_corners(hipStream_t s, const uint8_t *verts, uint32_t stride, const uint32_t *idx_sorted, const uint32_t *perm, uint32_t ntris,
                    float4 *corners);   // 3 * ntris entries
size_t project_box_entries(uint32_t ntris);
void launch_finish_pack(hipStream_t s, const ProjectParams &pp, const FinishPackArgs &fa, unsigned long long *stats);
// per ray: gather the queued big-footprint triangles, then hits per 256-ray block
void launch_project_finish(hipStream_t s, const ProjectParams &pp, unsigned long long *best, const void *big,
                           uint32_t big_capacity, const uint32_t *big_count, uint32_t *block_counts,
                           unsigned long long *stats, uint32_t rays_per_lane = 1);
// rays_per_lane (here and in launch_pack_keys; 1, 2, 4, pack also 8): frames that overlap on three streams are bound by wave
// slots, and these two passes then run as fewer, fatter waves -- project_rays_per_lane() says how fat for a raster
inline uint32_t project_rays_per_lane(uint32_t ray_blocks) { return ray_blocks >= 2048u ? 8u : ray_blocks >= 1024u ? 4u : ray_blocks >= 512u ? 2u : 1u; }
// one idle wave for `ticks` x 10 ns (stream / hardware-queue calibration)
void launch_spin(hipStream_t s, unsigned long long ticks);
// sensor -> world transform of a packed cloud (m.a = A, m.rinv = R, m.t = t); max_points bounds the grid
void launch_cloud_to_world(hipStream_t s, const Affine &m, const void *in, const uint32_t *n_points, void *out,
                           const uint32_t *out_base, uint32_t *out_total, uint32_t capacity, uint32_t max_points);
// dense per-ray (t, global triangle id) arrays rebuilt from the packed hit records (debug view)
void launch_dense_from_hits(hipStream_t s, const SensorTables &tb, const void *hits, const uint32_t *n_points,
                            const GeomTable &gt, float *t, uint32_t *gid);
// what a rebuild of gathered slots tells the host (pinned memory; the sized gather of include/lidarshooter_group.h)
struct GatherStat {
    uint32_t max_count;   // the largest rank's true hit count of the frame
    uint32_t truncated;   // 1: some rank had more hits than travelled -- the frame's cloud is incomplete
    uint32_t epoch;       // the caller's tag, released last
    uint32_t pad[13];
};
void launch_expand_slots(hipStream_t s, const SensorTables &tb, const uint32_t *gathered, uint32_t world, uint32_t cap,
                         uint32_t slot_words, uint8_t *points32, void *hits, uint32_t *n_points, GatherStat *stat = nullptr,
                         uint32_t epoch = 0);
void launch_raygen(hipStream_t s, const SensorTables &tb, float *dx, float *dy, float *dz);
// the reference's own buffers (LidarDeviceKernels.cu:38-51): Ray 32 B, Hit 24 B per ray; either may be nullptr
void launch_raygen_aos(hipStream_t s, const SensorTables &tb, void *rays32, void *hits24);
void launch_bruteforce(hipStream_t s, const SensorTables &tb, const float *verts, const uint32_t *tris,
                       uint32_t ntris, float *t_out, uint32_t *gid_out);

}  // namespace ls
