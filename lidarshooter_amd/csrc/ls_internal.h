// ls_internal.h -- the tracer handle (struct ls_tracer) and the functions the translation units of the host side share.
// Internal to liblidarshooter_hip.so; the public surface is include/lidarshooter_hip.h (+ lidarshooter_hip_debug.h).
//
//   ls_handle.cpp    lifetime, sensor tables, options, info, shard / stream / output buffers, the small stand-alone calls
//   ls_registry.cpp  ITracer's geometry bookkeeping: add / remove / update (EmbreeTracer.cpp:115-288), uploads
//   ls_commit.cpp    commitScene: layout, group-culling data, BVH build / refit / instanced hierarchies
//   ls_trace.cpp     traceScene: output buffers, frames in flight, the per-frame launch sequence, stage timings
//   ls_host_pool.cpp worker threads for host-side copies, point expansion
//   ls_debug.cpp     include/lidarshooter_hip_debug.h (tests and bench.py only)
#pragma once

#include "../../include/lidarshooter_hip.h"
#include "../../include/lidarshooter_hip_debug.h"   // (LS_OPT_DEBUG_FAULT)
#include "ls_kernels.h"
#include "ls_launch.h"
#include "ls_tuning.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace lsi {

struct Geometry {
    std::string name;
    int id = -1;
    uint32_t n_verts = 0, n_tris = 0;
    bool quad = false;          // RTC_GEOMETRY_TYPE_QUAD: n_elems quads, traced as n_tris = 2 n_elems triangles
    uint32_t n_elems = 0;       // elements as registered by addGeometry (= n_tris for triangle geometries)
    uint32_t *d_quad_idx = nullptr;   // 4*n_elems indices as handed over (converted into d_idx)
    void *d_raw = nullptr;      // vertex records as uploaded (n_verts * stride bytes)
    size_t raw_cap = 0;
    uint32_t stride = 0;
    uint32_t *d_idx = nullptr;  // 3*n_tris mesh-local vertex indices
    const void *shared_raw = nullptr;       // ls_update_geometry_device_shared: caller-owned device buffers
    const uint32_t *shared_idx = nullptr;   // read in place by the kernels, never copied or freed
    bool has_verts = false, has_idx = false, idx_dirty = true;
    // index validation: every index upload / hand-over launches k_index_max into d_idx_max; the next commit reads it back
    uint32_t *d_idx_max = nullptr;
    bool idx_unchecked = false;   // a reduction is in flight (or done) whose result no commit has looked at
    bool idx_bad = false;         // the last check found an index >= n_verts: every commit fails until new indices arrive
    uint32_t idx_bad_value = 0;
    // host uploads (ls_update_geometry): pinned staging, written by the copy pool, read by the DMA
    void *h_stage_v = nullptr, *h_stage_i = nullptr;
    size_t stage_v_cap = 0, stage_i_cap = 0;
    hipEvent_t ev_stage_v = nullptr, ev_stage_i = nullptr;   // recorded behind the last DMA that reads the staging buffer
    // group culling (projection engine, meshes with 64 triangles per wave): Morton order of the triangles, the
    // indices in that order, a mesh-space bound (sheared box) per kCullGroup sorted triangles
    uint32_t *d_perm = nullptr, *d_idx_sorted = nullptr;
    float4 *d_boxes = nullptr, *d_corners = nullptr;
    bool order_stale = true;    // the topology changed since d_perm / d_idx_sorted were made
    bool bounds_stale = true;   // vertices (may have) changed since d_boxes were made
    bool blas_dirty = true;     // BVH engine, instanced mode: vertices or topology changed since this geometry's hierarchy was built
    bool blas_topo_dirty = true;   // ... the topology did (vertices alone: the sorted order stays, the hierarchy is refitted)
    uint64_t blas_sorted_epoch = 0;   // key_scratch_epoch at which this geometry's sorted keys were written (0: never)
    float mesh_maxabs = 0.0f;   // largest |coordinate| of the mesh as uploaded (read back when that hierarchy is built)
    const void *raw() const { return shared_raw ? shared_raw : d_raw; }
    const uint32_t *idx() const { return shared_idx ? shared_idx : d_idx; }
    float affine[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
};

// LS_OPT_FRAME_GRAPH: the captured launches of one stream of the three-stream rotation
struct FrameGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t sig = 0;                       // what decides the launch SEQUENCE (trace_locked: frame_signature)
    std::vector<ls::LaunchRecord> recs;     // the launches as captured (arguments as of the last replay), in order
    std::vector<hipGraphNode_t> nodes;      // their kernel nodes
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
};

}  // namespace lsi

struct ls_tracer {
    std::mutex mu;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    // sensor (LidarDevice state needed by the path)
    std::vector<float> vertical;
    float h_begin = 0, h_end = 0, h_step = 0;
    std::vector<float> given_tables;   // ls_tracer_create_tables: sin_theta[V] cos_theta[V] sin_phi[H] cos_phi[H] as handed over
    uint32_t V = 0, H = 0;
    float rinv[9], t[3];
    float *d_tables = nullptr;  // sin_theta[V] cos_theta[V] sin_phi[H] cos_phi[H] ... (fill_tables)
    std::vector<float> host_tables;   // the same words on the host (ls_trace_scene_expand rebuilds points from (ray, t) records)
    float lut_t0 = 0.0f, lut_scale = 0.0f;   // k_cull's channel look-up table (ProjectParams::chan_lut)
    bool lut_ok = false;
    uint32_t az0 = 0, naz = 0;

    // geometry registry
    std::map<std::string, lsi::Geometry> geoms;
    long geometry_count = 0;
    bool layout_dirty = true;

    // committed scene
    std::vector<int> slot_geom_ids;       // geometry ids in layout order
    std::vector<uint32_t> slot_tri_first; // [n+1]
    uint32_t n_verts = 0, n_tris = 0, n_leaves = 0, n_slots = 0, leaf_size = 1, committed_leaf_size = 1;
    bool committed = false;
    lsi::DevBuf<float> verts;
    lsi::DevBuf<uint32_t> tris, keys_a, keys_b, vals_a, vals_b, geom_table;
    lsi::DevBuf<uint8_t> sort_temp;
    lsi::DevBuf<ls::TriRecord> records;
    lsi::DevBuf<ls::FatNode> nodes;
    lsi::DevBuf<float4> range_boxes;
    lsi::DevBuf<unsigned long long> best_keys;  // projection engine: per-ray (t bits, gid) closest-hit key
    lsi::DevBuf<uint8_t> big_queue;             // projection engine: triangles with very large footprints
    uint32_t big_capacity = 0;
    uint32_t *d_big_count = nullptr;
    bool keys_armed = false;               // best_keys all ~0, counters 0 (k_pack re-arms them every frame)
    uint32_t frame_parity = 0;             // which of the two block-counter arrays this frame adds into
    bool scene_materialized = false;       // verts / tris hold the transformed scene of the last commit
    struct LayoutEntry { std::string name; uint32_t vfirst, tfirst; };
    std::vector<LayoutEntry> layout;
    bool projection_ok = true;             // all channel elevations within [-90, 90] degrees
    int engine = 0;                        // LS_OPT_ENGINE: 0 auto, 1 BVH, 2 projection
    bool bvh_built = false;
    lsi::DevBuf<uint32_t> spill;       // traversal-stack overflow area of the persistent trace grid
    uint32_t *d_queue_heads = nullptr;
    bool queue_heads_armed = false;   // the last BVH frame's k_rowcount zeroed them again
    uint32_t trace_blocks = 0, chan_mul = 1, refill_min = 56;
    // LS_OPT_PIPELINE: the finish + pack workgroups of frame i ride in the launch of frame i+1's k_project;
    // everything a frame in flight touches exists twice (parity), the queue counter three times
    int opt_pipeline = 0;
    bool pipe_pending = false;              // a frame is projected, its finish + pack not launched yet
    uint32_t pipe_seq = 0;                  // frames issued in pipelined mode since the last flush
    ls::FinishPackArgs pipe_fa{};           // the pending frame's finish + pack
    lsi::DevBuf<unsigned long long> pack_status; // chained prefix: (epoch << 32) | hits of every 256-ray workgroup
    uint32_t pack_epoch = 0;
    lsi::DevBuf<unsigned long long> best_keys_b; // parity 1 twins of best_keys, big_queue, points, hits, d_n_points
    lsi::DevBuf<uint8_t> big_queue_b, points_b, hits_b;
    uint32_t *d_n_points_b = nullptr;
    bool keys_b_armed = false;
    // LS_OPT_PIPELINE = 2: whole frames rotate over three streams (three frames in flight); slot 0 / 1 use
    // the buffers above, slot 2 the ones below; every slot has its own block-count array
    hipStream_t slot_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_main = nullptr, ev_done[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_frame = nullptr;          // ls_tracer_order_after_last_frame on the handle's stream
    bool slot_pending[3] = {false, false, false};   // frames issued on slot_stream[s] since the last flush
    // the library enqueues mesh copies on the handle's stream; a slot stream whose epoch is behind orders itself
    // after that stream before its next frame (every slot, not only the first frame after the copy)
    uint64_t main_epoch = 0, slot_epoch[3] = {0, 0, 0};
    uint32_t ms_seq = 0;
    lsi::DevBuf<unsigned long long> best_keys_c;
    lsi::DevBuf<uint8_t> big_queue_c, points_c, hits_c;
    uint32_t *d_n_points_c = nullptr;
    bool keys_c_armed = false;
    // three-stream mode, small shards: finish + pack of a frame as ONE launch with the chained prefix (k_finish_pack); per slot
    // the status words of its ray blocks and, behind them, the tag word the kernel itself steps (FinishPackArgs::epoch_word)
    lsi::DevBuf<unsigned long long> pack_status_ms;
    bool traced_projection = false;        // the last trace ran on the projection engine (dense arrays on demand)
    const void *last_d_hits = nullptr;     // its hit records and count (device)
    const uint32_t *last_d_n = nullptr;
    std::vector<ls::GeomSource> project_srcs;  // scratch of trace_locked
    ls::RangeTree rt{};
    uint32_t range_entries = 0;
    uint32_t *d_maxabs = nullptr;
    unsigned long long *d_visits = nullptr;

    // trace outputs
    lsi::DevBuf<float> hit_t;
    lsi::DevBuf<uint32_t> hit_gid, row_counts;
    lsi::DevBuf<uint8_t> points;   // 32 B per ray
    lsi::DevBuf<uint8_t> hits;     // 16 B per ray
    uint32_t *d_n_points = nullptr;
    void *ext_points = nullptr, *ext_hits = nullptr;
    uint32_t *ext_n_points = nullptr;
    uint32_t ext_capacity = 0;
    bool ext_hits_only = false;   // ls_tracer_set_hit_buffers: ext_points is only a marker, no point record may be written
    uint8_t *h_points = nullptr;
    ls_hit *h_hits = nullptr;
    size_t h_cap = 0;  // records
    uint32_t *h_n_points = nullptr;
    bool traced = false;
    int opt_bvh_refit = 1;       // LS_OPT_BVH_REFIT
    int opt_bvh_instanced = 1;   // LS_OPT_BVH_INSTANCED: per-geometry hierarchies in mesh space, no build / refit for pose changes
    bool bvh_inst = false;       // the committed BVH is the instanced one
    bool inst_valid = false;     // records / nodes / inst_layout hold instanced hierarchies for the current layout and leaf size
    uint64_t key_scratch_epoch = 1;   // bumped whenever keys_b / vals_b are overwritten by something other than a geometry's own slice
    bool last_commit_built = false;
    struct InstSlot {
        uint32_t node_first, rec_first, n_leaves, range_first;
        ls::RangeTree rt;
        bool wide_made = false;   // its four-wide twins (wide_nodes) are those of its current binary nodes
        uint32_t wide_age = 0;    // frames traced over the current binary nodes while the twins were not made (kWidenAfterFrames)
    };
    std::vector<InstSlot> inst_layout;   // per layout entry
    uint32_t inst_leaf_size = 0;
    lsi::DevBuf<float> inst_verts;    // packed mesh-space vertices of all geometries (build input)
    uint32_t *d_inst_maxabs = nullptr;   // kGeomsPerLaunch words
    lsi::DevBuf<ls::FatNode> treelet; // one-geometry scenes: the top of that hierarchy, breadth-first (k_trace_inst stages it in LDS)
    bool treelet_valid = false;
    lsi::DevBuf<ls::WideNode> wide_nodes;   // instanced mode, LS_OPT_BVH_WIDE: the four-wide twins of `nodes` (k_widen), same indexing
    bool wide_valid = false;     // the twins' buffer exists and LS_OPT_BVH_WIDE was on at the last commit
    bool wide_in_use = false;    // the last trace walked them (every geometry's were made)
    int opt_bvh_wide = 1;
    bool bvh_order_valid = false;   // keys_b / vals_b hold the sorted Morton keys / order of the scene's triangles
    bool tris_rebased = false;          // tr->tris holds the rebased indices of the current layout and index uploads
    bool classic_nodes_valid = false;   // tr->nodes holds the classic hierarchy of the keys bvh_order_valid speaks of (k_refit_nodes may reuse its topology)
    uint32_t bvh_order_tris = 0;
    bool last_commit_refit = false;
    int opt_block_cull = 2;      // LS_OPT_BLOCK_CULL: 0 off, 1 on, 2 auto (geometries of 2 M triangles or more; from 512 k under an azimuth shard: cull_enabled)
    lsi::DevBuf<uint32_t> cull_list;  // three survivor lists (one per frame that can be in flight) of cull_chunks entries
    uint32_t cull_chunks = 0;
    uint32_t *d_aabb6 = nullptr; // scratch of launch_mesh_order
    int opt_host_output = 1;     // LS_OPT_HOST_OUTPUT: the pack kernel writes the pinned host buffers itself
    int opt_readback_hits = 1;   // LS_OPT_READBACK_HITS
    int opt_upload_mode = 1;     // LS_OPT_UPLOAD_MODE
    int opt_emit_points = 1;     // LS_OPT_EMIT_POINTS
    int opt_debug_fault = 0;     // LS_OPT_DEBUG_FAULT (one frame)
    uint32_t *h_status = nullptr;   // sticky device status word in pinned host memory (bit 0: chained prefix gave up)
    uint32_t cull_hint_in_use = 0;  // the survivor hint the culled grid is sized from (h_status[1] with hysteresis: trace_once)
    // ls_trace_scene_begin / ls_trace_scene_expand: the device tells the host how far the frame is (ls::HostProgress)
    ls::HostProgress *h_progress = nullptr;   // pinned host memory
    uint32_t progress_epoch = 0;
    bool progress_req = false;                // set by ls_trace_scene_begin around its trace_locked call
    bool progress_active = false;             // the frame begun last reports through h_progress (else it is complete already)
    bool begin_open = false;                  // a begun frame waits for its ls_trace_scene_expand
    uint32_t begin_points = 0, begin_first = 0, begin_blocks = 0;
    uint32_t pack_split = 0, pack_split_blocks = 0;   // ray block at which the pack pass's second launch starts, and the raster it was made for
    int concurrent_streams = 0;     // LS_OPT_PIPELINE = 2 calibration result (0 = not run yet)

    // LS_OPT_FRAME_GRAPH (three-stream mode): one captured graph per slot stream; fg_sink collects the launches of the
    // frame being built (capture: launched into the capturing stream and recorded; describe: only recorded)
    int opt_frame_graph = 0;
    bool fg_broken = false;      // the runtime refused a capture: plain launches from then on
    bool fg_bracket = false;     // ls_frame_graph_begin: the next frame's graph stays open for the caller's work
    bool fg_open = false;        // a frame is being captured / described (fg_sink.mode says which)
    uint64_t fg_tag = 0;
    uint32_t fg_slot = 0;
    lsi::FrameGraph fgraph[3];
    ls::LaunchSink fg_sink;
    uint64_t fg_captures = 0, fg_replays = 0, fg_patches = 0, fg_patch_waits = 0;   // (waits: a patch found its stream's previous frame still in flight)
    uint32_t fg_last_patched = 0;  // bit i: launch i of the frame replayed last went out with new arguments
    uint32_t last_slot = 0xFFFFFFFFu;   // the frame issued last: its stream of the three-stream rotation (none: it ran on `stream`)
    hipStream_t last_stream = nullptr;

    // options / measurement
    int opt_timing = 0;  // 0 off, 1 every stage, 2 only the trace kernel
    bool opt_count = false;
    // hipEvent records: one TimingRecord per frame (commit marks 0..6, trace marks 7..9), kept until
    // ls_get_timings averages and recycles them, so that timing a run never synchronises inside it.
    struct TimingRecord {
        hipEvent_t ev[LS_T_COUNT + 2];
        bool set[LS_T_COUNT + 2];
    };
    std::vector<TimingRecord> trec;
    size_t trec_used = 0;
    bool trec_open = false;
};

namespace lsi {

#define LS_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            tr->err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
            return LS_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

inline int fail(ls_tracer *tr, int code, const char *msg)
{
    tr->err = msg;
    return code;
}

template <typename T>
inline int ensure(ls_tracer *tr, DevBuf<T> &b, size_t need)
{
    if (need <= b.cap) return LS_OK;
    if (b.p) LS_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t cap = need + need / 8 + 64;
    LS_HIP(hipMalloc(reinterpret_cast<void **>(&b.p), cap * sizeof(T)));
    b.cap = cap;
    return LS_OK;
}

template <typename T>
inline void release(DevBuf<T> &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

constexpr float kProjectMarginDeg = 0.005f;  // AZIMUTH slack of the footprints, ~8.7e-5 rad: 5x the polynomial atan2 error (1e-3 deg) + table / test rounding
// ELEVATION slack of the channel tables (tan(chi +- margin), fill_tables): what it has to cover is the distance between a
// channel's nominal elevation and the elevation of the ray the kernels actually build for it -- theta = float((90 - chi) pi / 180)
// is off by half an ulp of ~1.5 rad (6e-8 rad), sinf / cosf by an ulp each, |(cos phi, sin phi)| by 1.2e-7 relative: under
// 3e-7 rad = 1.7e-5 degrees in all (the quotients of the band test and k_cull's remainder carry their own relative slack).
// Until round 5 the azimuth's 0.005 degrees stood here too: 1.7e-4 rad over both sides of every ring, 3 % of the ring
// spacing of SYN-128 -- a sixth of all the footprints k_project expanded at SYN-1M and a quarter of the groups k_cull kept at
// SYN-10M were that margin's.
constexpr float kProjectElevMarginDeg = 2e-4f;

inline bool use_projection(const ls_tracer *tr) { return tr->engine == 2 || (tr->engine == 0 && tr->projection_ok); }
inline uint32_t shard_rays(const ls_tracer *tr) { return tr->V * tr->naz; }

// ls_trace.cpp
ls::SensorTables tables(const ls_tracer *tr);
ls::ProjectParams project_params(const ls_tracer *tr);
ls::GeomTable geom_table(const ls_tracer *tr);
int ensure_slot_streams(ls_tracer *tr);
int check_device_status(ls_tracer *tr);
int flush_pipeline(ls_tracer *tr);   // order the handle's stream after every frame still in flight (no host wait)
inline int order_after_projects(ls_tracer *tr) { return flush_pipeline(tr); }
void mark(ls_tracer *tr, int i, hipEvent_t *ride = nullptr);
int trace_locked(ls_tracer *tr, uint32_t frame, ls_frame *out, bool readback);
void frame_graph_destroy(ls_tracer *tr);   // every cached graph (nothing may be open)
// installs the handle's launch sink for the calling thread while an entry point runs (LS_ENTER)
struct SinkScope {
    ls::LaunchSink *prev;
    explicit SinkScope(ls_tracer *tr) : prev(ls::thread_sink()) { ls::thread_sink() = tr->fg_open ? &tr->fg_sink : nullptr; }
    ~SinkScope() { ls::thread_sink() = prev; }
};
// ls_registry.cpp
void affine_from_components(const float *lin, const float *ang, float *A);
void free_geometry(Geometry &g);
// ls_commit.cpp
int materialize_scene(ls_tracer *tr, bool with_maxabs, bool keep_indices = false);
bool cull_enabled(const ls_tracer *tr, const Geometry &g);
bool shard_sector(const ls_tracer *tr, double &lo_deg, double &hi_deg);   // ls_trace.cpp
bool inst_inverse(const ls_tracer *tr, const Geometry &ge, double *minv9, double *o3, double *cond);
int commit_locked(ls_tracer *tr);
// ls_host_pool.cpp
void parallel_copy(void *dst, const void *src, size_t bytes);
int host_pool_threads();
constexpr size_t kExpandItem = 16384;   // points per work item of an expansion: 256 KB read, 512 KB written
void expand_points_range(uint8_t *dst_points32, const uint8_t *compact16, size_t count);
// 8-byte (ray, t) records in ascending ray order -> 32-byte points: xyz = t * direction from the factor tables, the float
// operations of k_pack in the same order
void expand_hits_range(uint8_t *dst_points32, const uint8_t *hits8, size_t count, const float *sin_theta, const float *cos_theta,
                       const float *cs_phi, uint32_t V, uint32_t H);
void pool_run(size_t n, const std::function<void(size_t)> &fn);   // fn(0) .. fn(n-1) on the worker threads and the caller

}  // namespace lsi

// every entry point that takes a handle: argument check, the handle's mutex, its device
#define LS_ENTER(tr)                                   \
    if (!(tr)) return LS_ERR_INVALID_ARGUMENT;         \
    std::lock_guard<std::mutex> lock_((tr)->mu);       \
    lsi::SinkScope sink_scope_(tr);                    \
    if (hipSetDevice((tr)->device) != hipSuccess) return lsi::fail((tr), LS_ERR_HIP, "hipSetDevice failed")
