// ls_tracer.cpp -- host side of the C ABI in include/lidarshooter_hip.h: the tracer handle, its
// geometry registry (ITracer semantics, EmbreeTracer.cpp:115-295), device memory and the per-frame
// launch sequence.  No compute happens on the host: if there is no usable HIP device the create
// call fails (LS_ERR_NO_DEVICE); there is no CPU fallback anywhere in this library.
#include "../../include/lidarshooter_hip.h"
#include "ls_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

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
    // host uploads (ls_update_geometry): pinned staging, written by the copy pool, read by the DMA
    void *h_stage_v = nullptr, *h_stage_i = nullptr;
    size_t stage_v_cap = 0, stage_i_cap = 0;
    hipEvent_t ev_stage_v = nullptr, ev_stage_i = nullptr;   // recorded behind the last DMA that reads the staging buffer
    // group culling (projection engine, meshes with 64 triangles per wave): Morton order of the triangles, the
    // indices in that order, a mesh-space bound (sheared box) per kCullGroup sorted triangles
    uint32_t *d_perm = nullptr, *d_idx_sorted = nullptr;
    float4 *d_boxes = nullptr;
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

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
};

// Worker threads for host-side copies (pageable caller memory <-> pinned staging).  One pool per process,
// created on first use; LS_HOST_THREADS overrides its size (default: half the cores, at most 8).
class HostPool {
public:
    static HostPool &get()
    {
        static HostPool pool;
        return pool;
    }
    int threads() const { return (int)workers_.size() + 1; }
    // run fn(0) .. fn(n-1); the calling thread works too.  `on_done(i)` (optional) is called on the CALLING thread,
    // in index order, as soon as item i is complete -- the caller enqueues item i's DMA there while later items copy.
    void run(size_t n, const std::function<void(size_t)> &fn, const std::function<void(size_t)> *on_done = nullptr)
    {
        if (!n) return;
        if (workers_.empty() || n == 1) {
            for (size_t i = 0; i < n; ++i) { fn(i); if (on_done) (*on_done)(i); }
            return;
        }
        std::unique_lock<std::mutex> run_lock(run_mu_);   // one job at a time
        std::vector<std::atomic<uint8_t>> done(n);
        for (auto &d : done) d.store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            done_ = done.data();
            n_ = n;
            next_.store(0);
            active_ = workers_.size();
            ++generation_;
        }
        cv_.notify_all();
        size_t reported = 0;
        auto report = [&]() {
            while (on_done && reported < n && done[reported].load(std::memory_order_acquire)) (*on_done)(reported++);
        };
        if (!on_done) {
            for (size_t i; (i = next_.fetch_add(1)) < n;) { fn(i); done[i].store(1, std::memory_order_release); }
        } else {
            // the caller only copies when nothing is waiting to be reported (its DMA calls are what the device waits for)
            while (reported < n) {
                report();
                if (reported == n) break;
                if (!done[reported].load(std::memory_order_acquire)) {
                    const size_t i = next_.fetch_add(1);
                    if (i < n) { fn(i); done[i].store(1, std::memory_order_release); }
                    else std::this_thread::yield();
                }
            }
        }
        std::unique_lock<std::mutex> lk(mu_);
        idle_cv_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    HostPool()
    {
        int n = 0;
        if (const char *e = getenv("LS_HOST_THREADS")) n = atoi(e);
        if (n <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = (int)std::min(8u, std::max(1u, hw / 2u));
        }
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            const std::function<void(size_t)> *fn = fn_;
            std::atomic<uint8_t> *done = done_;
            const size_t n = n_;
            lk.unlock();
            for (size_t i; (i = next_.fetch_add(1)) < n;) { (*fn)(i); done[i].store(1, std::memory_order_release); }
            lk.lock();
            if (--active_ == 0) idle_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, idle_cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    std::atomic<uint8_t> *done_ = nullptr;
    size_t n_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// host copy / DMA granularity of an upload (LS_COPY_CHUNK_KB, LS_DMA_RUN: tuning knobs of tools/dropin_bench.py)
static const size_t kCopyChunk = (getenv("LS_COPY_CHUNK_KB") ? (size_t)std::max(16, atoi(getenv("LS_COPY_CHUNK_KB"))) : 512u) << 10;
static const size_t kDmaRun = getenv("LS_DMA_RUN") ? (size_t)std::max(1, atoi(getenv("LS_DMA_RUN"))) : 4u;
// LS_UPLOAD_MODE: 1 (default) hipMemcpyAsync straight from the caller's pageable memory -- the runtime pins the pages
// for the duration of the call and the copy runs at PCIe rate (8 MB in 0.154 ms = 52 GB/s on the MI355X box);
// 0: copy pool -> pinned staging -> DMA, never waits for the device but costs 0.26-0.31 ms of host time for the same
// 8 MB; 2: one thread, one DMA (0.33 ms).  tools/upload_sweep.py.
static const int kUploadMode = getenv("LS_UPLOAD_MODE") ? atoi(getenv("LS_UPLOAD_MODE")) : 1;

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes <= kCopyChunk) { std::memcpy(dst, src, bytes); return; }
    const size_t n = (bytes + kCopyChunk - 1) / kCopyChunk;
    HostPool::get().run(n, [&](size_t i) {
        const size_t off = i * kCopyChunk;
        std::memcpy(static_cast<uint8_t *>(dst) + off, static_cast<const uint8_t *>(src) + off, std::min(kCopyChunk, bytes - off));
    });
}

}  // namespace

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
    float *d_tables = nullptr;  // sin_theta[V] cos_theta[V] sin_phi[H] cos_phi[H]
    uint32_t az0 = 0, naz = 0;

    // geometry registry
    std::map<std::string, Geometry> geoms;
    long geometry_count = 0;
    bool layout_dirty = true;

    // committed scene
    std::vector<int> slot_geom_ids;       // geometry ids in layout order
    std::vector<uint32_t> slot_tri_first; // [n+1]
    uint32_t n_verts = 0, n_tris = 0, n_leaves = 0, n_slots = 0, leaf_size = 1, committed_leaf_size = 1;
    bool committed = false;
    DevBuf<float> verts;
    DevBuf<uint32_t> tris, keys_a, keys_b, vals_a, vals_b, geom_table;
    DevBuf<uint8_t> sort_temp;
    DevBuf<ls::TriRecord> records;
    DevBuf<ls::FatNode> nodes;
    DevBuf<float4> range_boxes;
    DevBuf<unsigned long long> best_keys;  // projection engine: per-ray (t bits, gid) closest-hit key
    DevBuf<uint8_t> big_queue;             // projection engine: triangles with very large footprints
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
    DevBuf<uint32_t> spill;       // traversal-stack overflow area of the persistent trace grid
    uint32_t *d_queue_heads = nullptr;
    uint32_t trace_blocks = 0, chan_mul = 1, refill_min = 56;
    // LS_OPT_PIPELINE: the finish + pack workgroups of frame i ride in the launch of frame i+1's k_project;
    // everything a frame in flight touches exists twice (parity), the queue counter three times
    int opt_pipeline = 0;
    bool pipe_pending = false;              // a frame is projected, its finish + pack not launched yet
    uint32_t pipe_seq = 0;                  // frames issued in pipelined mode since the last flush
    ls::FinishPackArgs pipe_fa{};           // the pending frame's finish + pack
    DevBuf<unsigned long long> pack_status; // chained prefix: (epoch << 32) | hits of every 256-ray workgroup
    uint32_t pack_epoch = 0;
    DevBuf<unsigned long long> best_keys_b; // parity 1 twins of best_keys, big_queue, points, hits, d_n_points
    DevBuf<uint8_t> big_queue_b, points_b, hits_b;
    uint32_t *d_n_points_b = nullptr;
    bool keys_b_armed = false;
    // LS_OPT_PIPELINE = 2: whole frames rotate over three streams (three frames in flight); slot 0 / 1 use
    // the buffers above, slot 2 the ones below; every slot has its own block-count array
    hipStream_t slot_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_main = nullptr, ev_done[3] = {nullptr, nullptr, nullptr};
    bool slot_pending[3] = {false, false, false};   // frames issued on slot_stream[s] since the last flush
    // the library enqueues mesh copies on the handle's stream; a slot stream whose epoch is behind orders itself
    // after that stream before its next frame (every slot, not only the first frame after the copy)
    uint64_t main_epoch = 0, slot_epoch[3] = {0, 0, 0};
    uint32_t ms_seq = 0;
    DevBuf<unsigned long long> best_keys_c;
    DevBuf<uint8_t> big_queue_c, points_c, hits_c;
    uint32_t *d_n_points_c = nullptr;
    bool keys_c_armed = false;
    bool traced_projection = false;        // the last trace ran on the projection engine (dense arrays on demand)
    const void *last_d_hits = nullptr;     // its hit records and count (device)
    const uint32_t *last_d_n = nullptr;
    std::vector<ls::GeomSource> project_srcs;  // scratch of trace_locked
    ls::RangeTree rt{};
    uint32_t range_entries = 0;
    uint32_t *d_maxabs = nullptr;
    unsigned long long *d_visits = nullptr;

    // trace outputs
    DevBuf<float> hit_t;
    DevBuf<uint32_t> hit_gid, row_counts;
    DevBuf<uint8_t> points;   // 32 B per ray
    DevBuf<uint8_t> hits;     // 16 B per ray
    uint32_t *d_n_points = nullptr;
    void *ext_points = nullptr, *ext_hits = nullptr;
    uint32_t *ext_n_points = nullptr;
    uint32_t ext_capacity = 0;
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
    struct InstSlot { uint32_t node_first, rec_first, n_leaves, range_first; ls::RangeTree rt; };
    std::vector<InstSlot> inst_layout;   // per layout entry
    uint32_t inst_leaf_size = 0;
    DevBuf<float> inst_verts;    // packed mesh-space vertices of all geometries (build input)
    DevBuf<uint32_t> inst_tris;  // their indices, rebased
    uint32_t *d_inst_maxabs = nullptr;   // kGeomsPerLaunch words
    DevBuf<ls::FatNode> treelet; // one-geometry scenes: the top of that hierarchy, breadth-first (k_trace_inst stages it in LDS)
    bool treelet_valid = false;
    bool bvh_order_valid = false;   // keys_b / vals_b hold the sorted Morton keys / order of the scene's triangles
    uint32_t bvh_order_tris = 0;
    bool last_commit_refit = false;
    int opt_block_cull = 2;      // LS_OPT_BLOCK_CULL: 0 off, 1 on, 2 auto (geometries of 2 M triangles or more, cull_enabled)
    DevBuf<uint32_t> cull_list;  // three survivor lists (one per frame that can be in flight) of cull_chunks entries
    uint32_t cull_chunks = 0;
    uint32_t *d_aabb6 = nullptr; // scratch of launch_mesh_order
    int opt_host_output = 1;     // LS_OPT_HOST_OUTPUT: the pack kernel writes the pinned host buffers itself
    int opt_readback_hits = 1;   // LS_OPT_READBACK_HITS
    int opt_debug_fault = 0;     // LS_OPT_DEBUG_FAULT (one frame)
    uint32_t *h_status = nullptr;   // sticky device status word in pinned host memory (bit 0: chained prefix gave up)
    int concurrent_streams = 0;     // LS_OPT_PIPELINE = 2 calibration result (0 = not run yet)

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

namespace {

#define LS_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            tr->err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
            return LS_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

int fail(ls_tracer *tr, int code, const char *msg)
{
    tr->err = msg;
    return code;
}

template <typename T>
int ensure(ls_tracer *tr, DevBuf<T> &b, size_t need)
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
void release(DevBuf<T> &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

constexpr float kProjectMarginDeg = 0.005f;  // ~8.7e-5 rad: 5x the polynomial atan2 error (1e-3 deg) + table / test rounding

bool use_projection(const ls_tracer *tr) { return tr->engine == 2 || (tr->engine == 0 && tr->projection_ok); }

ls::ProjectParams project_params(const ls_tracer *tr);

ls::SensorTables tables(const ls_tracer *tr)
{
    ls::SensorTables tb;
    tb.sin_theta = tr->d_tables;
    tb.cos_theta = tr->d_tables + tr->V;
    tb.sin_phi = tr->d_tables + 2 * (size_t)tr->V;
    tb.cos_phi = tr->d_tables + 2 * (size_t)tr->V + tr->H;
    tb.cs_phi = reinterpret_cast<const float2 *>(tr->d_tables + 5 * (size_t)tr->V + 2 * (size_t)tr->H);
    tb.V = tr->V;
    tb.H = tr->H;
    tb.az0 = tr->az0;
    tb.naz = tr->naz;
    return tb;
}

ls::ProjectParams project_params(const ls_tracer *tr)
{
    ls::ProjectParams pp;
    pp.tb = tables(tr);
    pp.chan_tan_up = tr->d_tables + 2 * (size_t)tr->V + 2 * (size_t)tr->H;
    pp.chan_tan_dn = pp.chan_tan_up + tr->V;
    pp.chan_perm = reinterpret_cast<const uint32_t *>(pp.chan_tan_dn + tr->V);
    pp.chan_rank = reinterpret_cast<const uint32_t *>(tr->d_tables + 5 * (size_t)tr->V + 4 * (size_t)tr->H);
    pp.begin_deg = tr->h_begin;
    pp.step_deg = tr->h_step;  // LidarDevice.cpp:611
    pp.inv_step_deg = pp.step_deg != 0.0f ? 1.0f / pp.step_deg : 0.0f;
    pp.inv_period = std::fabs(pp.step_deg) / 360.0f;
    pp.margin_deg = kProjectMarginDeg;
    // azimuth sector of the shard, padded by the angular margin and 1.5 columns per side, as two boundary
    // directions in counter-clockwise order; used to reject triangles early when it spans less than 180 degrees
    pp.sector_on = 0;
    pp.sec_a[0] = pp.sec_a[1] = pp.sec_b[0] = pp.sec_b[1] = 0.0f;
    if (tr->naz < tr->H && pp.step_deg != 0.0f) {
        const double step = pp.step_deg, pad = kProjectMarginDeg + 1.5 * std::fabs(step);
        double lo = (double)tr->h_begin + step * (double)tr->az0, hi = (double)tr->h_begin + step * (double)(tr->az0 + tr->naz - 1u);
        if (lo > hi) std::swap(lo, hi);
        lo -= pad;
        hi += pad;
        if (hi - lo < 179.0) {
            pp.sector_on = 1;
            pp.sec_a[0] = (float)std::cos(lo * M_PI / 180.0); pp.sec_a[1] = (float)std::sin(lo * M_PI / 180.0);
            pp.sec_b[0] = (float)std::cos(hi * M_PI / 180.0); pp.sec_b[1] = (float)std::sin(hi * M_PI / 180.0);
        }
    }
    // tuning / ablation knobs of tools/sweep_*.sh, read once
    static const uint32_t big_cells = getenv("LS_PROJECT_BIG_CELLS") ? (uint32_t)atoi(getenv("LS_PROJECT_BIG_CELLS")) : 128u;
    static const int debug = getenv("LS_PROJECT_DEBUG") ? atoi(getenv("LS_PROJECT_DEBUG")) : 0;
    pp.big_cells = big_cells;
    pp.debug = debug;
    pp.spread = 1;   // trace_locked clears it for frames that overlap on the three slot streams
    return pp;
}

uint32_t shard_rays(const ls_tracer *tr) { return tr->V * tr->naz; }

// LS_OPT_PIPELINE = 2 needs three streams whose kernels really run side by side.  The runtime multiplexes
// streams onto a few hardware queues (which ones depends on every stream created before, by anybody in the
// process), and two streams on one queue serialise: 24 us per frame instead of 16.  So: candidates are created
// and tried pairwise with an idle 200 us wave each -- two on one queue take twice as long as two on two --
// until three mutually concurrent ones are found; the rest is destroyed.  A few milliseconds, once per handle.
// The number found is kept (LS_INFO_CONCURRENT_STREAMS); with fewer than three the handle runs mode 1 instead.
int pick_slot_streams(ls_tracer *tr)
{
    constexpr int kCandidates = 8;
    constexpr unsigned long long kTicks = 20000;   // 200 us
    hipStream_t cand[kCandidates] = {};
    for (auto &c : cand) LS_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    LS_HIP(hipEventCreate(&e0));
    LS_HIP(hipEventCreate(&ea));
    LS_HIP(hipEventCreate(&eb));
    // device-side timing: both waves are launched behind e0; on one hardware queue the second one ends ~400 us
    // after e0, on two queues ~200 us -- host scheduling noise does not enter
    auto pair_us = [&](hipStream_t a, hipStream_t b, double &us) -> int {
        LS_HIP(hipStreamSynchronize(a));
        LS_HIP(hipStreamSynchronize(b));
        LS_HIP(hipEventRecord(e0, a));
        ls::launch_spin(a, kTicks);
        ls::launch_spin(b, kTicks);
        LS_HIP(hipEventRecord(ea, a));
        LS_HIP(hipEventRecord(eb, b));
        LS_HIP(hipStreamSynchronize(a));
        LS_HIP(hipStreamSynchronize(b));
        float ma = 0.f, mb = 0.f;
        LS_HIP(hipEventElapsedTime(&ma, e0, ea));
        LS_HIP(hipEventElapsedTime(&mb, e0, eb));
        us = 1e3 * (double)std::max(ma, mb);
        return LS_OK;
    };
    int rc;
    double warm;
    if ((rc = pair_us(cand[0], cand[1], warm))) return rc;   // first launches: code object upload etc.
    int chosen[3] = {0, -1, -1}, n = 1;
    for (int c = 1; c < kCandidates && n < 3; ++c) {
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) {
            double us;
            if ((rc = pair_us(cand[chosen[k]], cand[c], us))) return rc;
            ok = us < 1.5 * (double)kTicks / 100.0;   // concurrent: ~200 us; serialised: ~400 us
        }
        if (ok) chosen[n++] = c;
    }
    tr->concurrent_streams = n;
    for (int i = 0; i < 3; ++i) tr->slot_stream[i] = cand[chosen[i] >= 0 ? chosen[i] : chosen[0]];
    for (int c = 0; c < kCandidates; ++c) {
        bool used = false;
        for (int i = 0; i < 3; ++i) used = used || tr->slot_stream[i] == cand[c];
        if (!used) (void)hipStreamDestroy(cand[c]);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return LS_OK;
}

int ensure_slot_streams(ls_tracer *tr)
{
    if (tr->slot_stream[0]) return LS_OK;
    int rc;
    LS_HIP(hipEventCreateWithFlags(&tr->ev_main, hipEventDisableTiming | hipEventDisableSystemFence));
    if ((rc = pick_slot_streams(tr))) return rc;
    for (int i = 0; i < 3; ++i)
        LS_HIP(hipEventCreateWithFlags(&tr->ev_done[i], hipEventDisableTiming | hipEventDisableSystemFence));
    return LS_OK;
}

// pinned host buffers of the synchronous ls_trace_scene (written by the pack kernel itself or by D2H copies)
int ensure_host_buffers(ls_tracer *tr, size_t records)
{
    if (records <= tr->h_cap) return LS_OK;
    if (tr->h_points) LS_HIP(hipHostFree(tr->h_points));
    if (tr->h_hits) LS_HIP(hipHostFree(tr->h_hits));
    tr->h_points = nullptr;
    tr->h_hits = nullptr;
    tr->h_cap = 0;
    LS_HIP(hipHostMalloc(reinterpret_cast<void **>(&tr->h_points), records * 32));
    LS_HIP(hipHostMalloc(reinterpret_cast<void **>(&tr->h_hits), records * 16));
    tr->h_cap = records;
    return LS_OK;
}

// the sticky device status word, read after a host wait: a frame whose chained prefix gave up is lost
int check_device_status(ls_tracer *tr)
{
    const uint32_t st = __atomic_exchange_n(tr->h_status, 0u, __ATOMIC_ACQ_REL);
    if (!st) return LS_OK;
    tr->err = "device status " + std::to_string(st) + ": the chained prefix of a pipelined finish + pack pass gave up waiting; "
              "that frame's cloud is incomplete";
    return LS_ERR_HIP;
}

int ensure_outputs(ls_tracer *tr)
{
    const size_t nr = shard_rays(tr);
    int rc;
    if ((rc = ensure(tr, tr->hit_t, nr))) return rc;
    if ((rc = ensure(tr, tr->hit_gid, nr))) return rc;
    {
        const size_t c0 = tr->row_counts.cap;
        if ((rc = ensure(tr, tr->row_counts, 4 * ((nr + 255) / 256) + 8))) return rc;  // two frame-parity arrays (+ two: three-stream mode)
        if (tr->row_counts.cap != c0) tr->keys_armed = false;
    }
    if (use_projection(tr)) {
        const size_t cap0 = tr->best_keys.cap;
        if ((rc = ensure(tr, tr->best_keys, nr))) return rc;
        if (tr->best_keys.cap != cap0) tr->keys_armed = false;
        if (!tr->big_queue.p) {
            tr->big_capacity = 2048u;  // culled per 256-ray workgroup in k_project_finish; overflow is expanded in place
            if ((rc = ensure(tr, tr->big_queue, (size_t)tr->big_capacity * ls::project_big_item_bytes()))) return rc;
        }
    } else {
        if ((rc = ensure(tr, tr->spill, ls::trace_spill_bytes(tr->trace_blocks) / 4))) return rc;
    }
    if (!tr->ext_points) {
        if ((rc = ensure(tr, tr->points, nr * 32))) return rc;
        if ((rc = ensure(tr, tr->hits, nr * 16))) return rc;
    }
    if (tr->opt_pipeline == 2 && use_projection(tr)) {
        const size_t cap0 = tr->best_keys_c.cap;
        if ((rc = ensure(tr, tr->best_keys_c, nr))) return rc;
        if (tr->best_keys_c.cap != cap0) tr->keys_c_armed = false;
        if (!tr->big_queue_c.p && (rc = ensure(tr, tr->big_queue_c, (size_t)tr->big_capacity * ls::project_big_item_bytes()))) return rc;
        if (!tr->ext_points) {
            if ((rc = ensure(tr, tr->points_c, nr * 32))) return rc;
            if ((rc = ensure(tr, tr->hits_c, nr * 16))) return rc;
        }
        if (!tr->d_n_points_c) LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_n_points_c), 4));
        if ((rc = ensure_slot_streams(tr))) return rc;
    }
    if ((tr->opt_pipeline || tr->pipe_seq) && use_projection(tr)) {   // twins: needed as long as the rotation may stand on parity 1
        const size_t cap0 = tr->best_keys_b.cap;
        if ((rc = ensure(tr, tr->best_keys_b, nr))) return rc;
        if (tr->best_keys_b.cap != cap0) tr->keys_b_armed = false;
        if (!tr->big_queue_b.p && (rc = ensure(tr, tr->big_queue_b, (size_t)tr->big_capacity * ls::project_big_item_bytes()))) return rc;
        if (!tr->ext_points) {
            if ((rc = ensure(tr, tr->points_b, nr * 32))) return rc;
            if ((rc = ensure(tr, tr->hits_b, nr * 16))) return rc;
        }
        if (!tr->d_n_points_b) LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_n_points_b), 4));
        {
            const size_t cap1 = tr->pack_status.cap;
            if ((rc = ensure(tr, tr->pack_status, (nr + 255) / 256 + 1))) return rc;
            if (tr->pack_status.cap != cap1) {   // fresh memory: no word may carry a live epoch tag
                LS_HIP(hipMemsetAsync(tr->pack_status.p, 0, tr->pack_status.cap * 8, tr->stream));
                tr->pack_epoch = 0;
            }
        }
    }
    return LS_OK;
}

ls::ProjectParams project_params(const ls_tracer *tr);

// LS_OPT_PIPELINE: launch the finish + pack of the frame still in flight on its own
// frames on one of the three streams (LS_OPT_PIPELINE = 2) may still read the meshes: the handle's stream, on
// which the next mesh copy is about to be enqueued, waits for them (same as a flush)
int flush_pipeline(ls_tracer *tr);
int order_after_projects(ls_tracer *tr) { return flush_pipeline(tr); }

int flush_pipeline(ls_tracer *tr)
{
    for (int i = 0; i < 3; ++i)
        if (tr->slot_pending[i]) {   // three-stream mode: no per-frame event; the one recorded now covers the stream's frames
            LS_HIP(hipEventRecord(tr->ev_done[i], tr->slot_stream[i]));
            LS_HIP(hipStreamWaitEvent(tr->stream, tr->ev_done[i], 0));
            tr->slot_pending[i] = false;
        }
    if (!tr->pipe_pending) return LS_OK;
    ls::launch_finish_pack(tr->stream, project_params(tr), tr->pipe_fa, nullptr);
    LS_HIP(hipGetLastError());
    tr->pipe_pending = false;   // pipe_seq goes on: the next frame, in any mode, takes the next twin and queue counter
    return LS_OK;
}

constexpr size_t kMaxTimingRecords = 4096;

// Marks 0..6 bracket the six commit stages, 7..9 bracket trace and pack.  A commit opens a new
// record; a trace without a preceding commit opens its own.
// marks: 0..6 bracket the six commit stages; 7..10 bracket trace, trace_aux and pack
// `ride` (optional): the event is not recorded on the stream here; the caller attaches it to a kernel dispatch
// (hipExtLaunchKernel), where it carries the kernel's own begin or end timestamp
void mark(ls_tracer *tr, int i, hipEvent_t *ride = nullptr)
{
    if (ride) *ride = nullptr;
    if (!tr->opt_timing) return;
    if (tr->opt_timing == 2 && i != 7 && i != 8) return;
    const bool opens = (i == 0) || (i == 7 && !tr->trec_open);
    if (opens) {
        if (tr->trec_used >= kMaxTimingRecords) { tr->trec_open = false; return; }
        if (tr->trec_used == tr->trec.size()) {
            ls_tracer::TimingRecord r;
            for (auto &e : r.ev) e = nullptr;
            tr->trec.push_back(r);
        }
        for (auto &b : tr->trec[tr->trec_used].set) b = false;
        ++tr->trec_used;
        tr->trec_open = true;
    }
    if (!tr->trec_open || tr->trec_used == 0) return;
    ls_tracer::TimingRecord &r = tr->trec[tr->trec_used - 1];
    if (!r.ev[i] && hipEventCreate(&r.ev[i]) != hipSuccess) return;
    if (ride) { *ride = r.ev[i]; r.set[i] = true; }
    else r.set[i] = hipEventRecord(r.ev[i], tr->stream) == hipSuccess;
    if (i == 10 || (tr->opt_timing == 2 && i == 8)) tr->trec_open = false;
}

// LidarDevice.cpp:306-316 on the host: the V+H distinct angles of a revolution go through libm
// (sinf/cosf, exactly like the reference's CPU path); the kernels only multiply table entries.
void fill_tables(const ls_tracer *tr, std::vector<float> &tab)
{
    const uint32_t V = tr->V, H = tr->H;
    tab.resize(2 * (size_t)V + 2 * (size_t)H + 3 * (size_t)V + 2 * (size_t)H + (size_t)V);
    const float step = tr->h_step;  // LidarDevice.cpp:611
    if (!tr->given_tables.empty()) {
        // ls_tracer_create_tables: the caller's factor tables, bit for bit
        std::memcpy(tab.data(), tr->given_tables.data(), (2 * (size_t)V + 2 * (size_t)H) * sizeof(float));
    } else {
        for (uint32_t v = 0; v < V; ++v) {
            const float preChi = tr->vertical[v];
            const float theta = static_cast<float>((90.0 - static_cast<double>(preChi)) * M_PI / 180.0);
            tab[v] = std::sin(theta);
            tab[V + v] = std::cos(theta);
        }
        for (uint32_t h = 0; h < H; ++h) {
            const float prePhi = tr->h_begin + step * static_cast<float>(h);
            const float phi = static_cast<float>(static_cast<double>(prePhi) * M_PI / 180.0);
            tab[2 * (size_t)V + h] = std::sin(phi);
            tab[2 * (size_t)V + H + h] = std::cos(phi);
        }
    }
    // projection engine: channels by ascending elevation; tan(elevation +- margin), nudged outwards
    std::vector<uint32_t> perm(V);
    for (uint32_t v = 0; v < V; ++v) perm[v] = v;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return tr->vertical[a] < tr->vertical[b]; });
    float *up = tab.data() + 2 * (size_t)V + 2 * (size_t)H, *dn = up + V;
    for (uint32_t i = 0; i < V; ++i) {
        const double chi = tr->vertical[perm[i]];
        const double hi = chi + kProjectMarginDeg, lo = chi - kProjectMarginDeg;
        float tu = hi >= 90.0 ? INFINITY : (hi <= -90.0 ? -INFINITY : static_cast<float>(std::tan(hi * M_PI / 180.0)));
        float td = lo <= -90.0 ? -INFINITY : (lo >= 90.0 ? INFINITY : static_cast<float>(std::tan(lo * M_PI / 180.0)));
        up[i] = std::nextafter(tu, INFINITY);
        dn[i] = std::nextafter(td, -INFINITY);
        std::memcpy(&dn[V + i], &perm[i], 4);
    }
    // (cos_phi, sin_phi) interleaved
    float *cs = tab.data() + 2 * (size_t)V + 2 * (size_t)H + 3 * (size_t)V;
    for (uint32_t h = 0; h < H; ++h) {
        cs[2 * (size_t)h] = tab[2 * (size_t)V + H + h];
        cs[2 * (size_t)h + 1] = tab[2 * (size_t)V + h];
    }
    // inverse channel permutation
    float *rank = cs + 2 * (size_t)H;
    for (uint32_t i = 0; i < V; ++i) {
        const uint32_t pos = i;
        std::memcpy(&rank[perm[i]], &pos, 4);
    }
}

// MeshTransformer.cpp:467-477 (Eigen: ((Translation*Rz)*Ry)*Rx, AngleAxis::toRotationMatrix)
void angle_axis_unit(float angle, int axis, float *m)
{
    float ax[3] = {0.f, 0.f, 0.f};
    ax[axis] = 1.0f;
    const float s = std::sin(angle), c = std::cos(angle);
    const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; m[1] = tmp - sa[2]; m[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; m[2] = tmp + sa[1]; m[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; m[5] = tmp - sa[0]; m[7] = tmp + sa[0];
    m[0] = ca[0] * ax[0] + c;
    m[4] = ca[1] * ax[1] + c;
    m[8] = ca[2] * ax[2] + c;
}

void mat3_mul(const float *a, const float *b, float *o)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[3 * i + j] = (a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}

void affine_from_components(const float *lin, const float *ang, float *A)
{
    float rx[9], ry[9], rz[9], zy[9], zyx[9];
    angle_axis_unit(ang[0], 0, rx);
    angle_axis_unit(ang[1], 1, ry);
    angle_axis_unit(ang[2], 2, rz);
    mat3_mul(rz, ry, zy);
    mat3_mul(zy, rx, zyx);
    for (int i = 0; i < 3; ++i) {
        A[4 * i + 0] = zyx[3 * i + 0];
        A[4 * i + 1] = zyx[3 * i + 1];
        A[4 * i + 2] = zyx[3 * i + 2];
        A[4 * i + 3] = lin[i];
    }
}

// Host memory -> device, without waiting for the device: the copy pool moves the caller's (pageable) bytes into
// a pinned staging buffer chunk by chunk, and the calling thread enqueues each chunk's DMA on the handle's
// stream as soon as the chunk is staged, so copying and DMA overlap.  When the call returns the caller's
// memory is free again (MeshProjector.cpp:448-461 reuses it); the staging buffer is protected by `ev`.
int stage_upload(ls_tracer *tr, void *&stage, size_t &stage_cap, hipEvent_t &ev, void *d_dst, const void *src, size_t bytes)
{
    if (!bytes) return LS_OK;
    if (kUploadMode == 1) {
        // the call returns once the caller's memory has been read (pageable source: the runtime waits for its own
        // staging / pinning); what follows on the stream is ordered behind the copy
        LS_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, tr->stream));
        LS_HIP(hipStreamSynchronize(tr->stream));
        return LS_OK;
    }
    if (bytes > stage_cap) {
        if (stage) { LS_HIP(hipStreamSynchronize(tr->stream)); LS_HIP(hipHostFree(stage)); }
        stage = nullptr;
        stage_cap = 0;
        LS_HIP(hipHostMalloc(&stage, bytes, hipHostMallocDefault));
        stage_cap = bytes;
    }
    if (!ev) LS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else LS_HIP(hipEventSynchronize(ev));   // a DMA of the previous upload may still read the buffer (asynchronous callers)
    const size_t n = (bytes + kCopyChunk - 1) / kCopyChunk;
    hipError_t err = hipSuccess;
    const std::function<void(size_t)> copy = [&](size_t i) {
        const size_t off = i * kCopyChunk;
        std::memcpy(static_cast<uint8_t *>(stage) + off, static_cast<const uint8_t *>(src) + off, std::min(kCopyChunk, bytes - off));
    };
    // DMAs are enqueued in runs of up to four staged chunks (2 MB): fewer API calls, still a fine-grained pipeline
    size_t run_first = 0, run_len = 0;
    auto flush_run = [&]() {
        if (!run_len) return;
        const size_t off = run_first * kCopyChunk, len = std::min(run_len * kCopyChunk, bytes - off);
        const hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + off, static_cast<uint8_t *>(stage) + off, len,
                                            hipMemcpyHostToDevice, tr->stream);
        if (e != hipSuccess && err == hipSuccess) err = e;
        run_len = 0;
    };
    const std::function<void(size_t)> dma = [&](size_t i) {
        if (!run_len) run_first = i;
        if (++run_len == kDmaRun || i + 1 == n) flush_run();
    };
    if (kUploadMode == 2) {   // ablation: one thread, one DMA
        std::memcpy(stage, src, bytes);
        run_first = 0;
        run_len = n;
        flush_run();
    } else {
        HostPool::get().run(n, copy, &dma);
        flush_run();
    }
    if (err != hipSuccess) { tr->err = std::string("hipMemcpyAsync (upload): ") + hipGetErrorString(err); return LS_ERR_HIP; }
    LS_HIP(hipEventRecord(ev, tr->stream));
    return LS_OK;
}

int update_common(ls_tracer *tr, const char *name, const float *affine, const void *verts, uint32_t stride,
                  const uint32_t *idx, hipMemcpyKind kind, bool shared = false)
{
    if (!name || !affine) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null argument");
    auto it = tr->geoms.find(name);
    if (it == tr->geoms.end()) return fail(tr, LS_ERR_UNKNOWN_GEOMETRY, "geometry key does not exist");
    Geometry &g = it->second;
    std::memcpy(g.affine, affine, sizeof(g.affine));
    if (shared) {
        if (stride < 12 || (stride & 3u)) return fail(tr, LS_ERR_INVALID_ARGUMENT, "vertex stride must be >= 12 and a multiple of 4");
        if (!g.has_verts || (idx && !g.has_idx)) tr->layout_dirty = true;
        g.shared_raw = verts;
        g.stride = stride;
        g.has_verts = true;
        g.bounds_stale = true;   // the caller's buffer may hold anything now
        g.blas_dirty = true;
        if (idx && g.quad) {
            // quads are traced as triangle pairs: the caller's indices are converted into a library-owned array
            if (!g.d_idx) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_idx), (size_t)g.n_tris * 12 + 4));
            const int rc = order_after_projects(tr);
            if (rc) return rc;
            ++tr->main_epoch;
            ls::launch_quads_to_triangles(tr->stream, idx, g.n_elems, g.d_idx);
            g.shared_idx = nullptr;
            if (!g.has_idx) tr->layout_dirty = true;
            g.has_idx = true; g.idx_dirty = true; g.order_stale = true; g.blas_dirty = g.blas_topo_dirty = true;
        } else if (idx) { g.shared_idx = idx; g.has_idx = true; g.idx_dirty = true; g.order_stale = true; g.blas_dirty = g.blas_topo_dirty = true; }
        return LS_OK;
    }
    if (!verts && !idx) return LS_OK;   // transform only: nothing is copied, nothing to order
    {   // three-stream mode: frames in flight may still read the mesh buffers this call overwrites
        const int rc = order_after_projects(tr);
        if (rc) return rc;
        ++tr->main_epoch;   // every slot stream must see the copies below before its next frame
    }
    const bool from_host = kind == hipMemcpyHostToDevice;
    if (verts) {
        g.shared_raw = nullptr;
        if (stride < 12 || (stride & 3u)) return fail(tr, LS_ERR_INVALID_ARGUMENT, "vertex stride must be >= 12 and a multiple of 4");
        const size_t bytes = (size_t)g.n_verts * stride;
        if (bytes > g.raw_cap) {
            if (g.d_raw) LS_HIP(hipFree(g.d_raw));
            g.d_raw = nullptr;
            g.raw_cap = 0;
            LS_HIP(hipMalloc(&g.d_raw, bytes ? bytes : 4));
            g.raw_cap = bytes;
        }
        if (bytes) {
            if (from_host) {
                const int rc = stage_upload(tr, g.h_stage_v, g.stage_v_cap, g.ev_stage_v, g.d_raw, verts, bytes);
                if (rc) return rc;
            } else {
                LS_HIP(hipMemcpyAsync(g.d_raw, verts, bytes, kind, tr->stream));
            }
        }
        g.stride = stride;
        if (!g.has_verts) tr->layout_dirty = true;
        g.has_verts = true;
        g.bounds_stale = true;
        g.blas_dirty = true;
    }
    if (idx) {
        g.shared_idx = nullptr;
        g.order_stale = true;
        g.blas_dirty = g.blas_topo_dirty = true;
        const size_t bytes = g.quad ? (size_t)g.n_elems * 16 : (size_t)g.n_tris * 12;
        if (!g.d_idx) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_idx), (size_t)g.n_tris * 12 + 4));
        if (g.quad && !g.d_quad_idx) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_quad_idx), bytes ? bytes : 4));
        uint32_t *dst = g.quad ? g.d_quad_idx : g.d_idx;
        if (bytes) {
            if (from_host) {
                const int rc = stage_upload(tr, g.h_stage_i, g.stage_i_cap, g.ev_stage_i, dst, idx, bytes);
                if (rc) return rc;
            } else {
                LS_HIP(hipMemcpyAsync(dst, idx, bytes, kind, tr->stream));
            }
            if (g.quad) ls::launch_quads_to_triangles(tr->stream, g.d_quad_idx, g.n_elems, g.d_idx);   // Embree's split of a quad
        }
        if (!g.has_idx) tr->layout_dirty = true;
        g.has_idx = true;
        g.idx_dirty = true;
    }
    return LS_OK;
}

void free_geometry(Geometry &g)
{
    if (g.d_raw) (void)hipFree(g.d_raw);
    if (g.d_idx) (void)hipFree(g.d_idx);
    if (g.d_quad_idx) (void)hipFree(g.d_quad_idx);
    g.d_quad_idx = nullptr;
    if (g.h_stage_v) (void)hipHostFree(g.h_stage_v);
    if (g.h_stage_i) (void)hipHostFree(g.h_stage_i);
    if (g.d_perm) (void)hipFree(g.d_perm);
    if (g.d_idx_sorted) (void)hipFree(g.d_idx_sorted);
    if (g.d_boxes) (void)hipFree(g.d_boxes);
    g.d_perm = g.d_idx_sorted = nullptr;
    g.d_boxes = nullptr;
    if (g.ev_stage_v) (void)hipEventDestroy(g.ev_stage_v);
    if (g.ev_stage_i) (void)hipEventDestroy(g.ev_stage_i);
    g.d_raw = nullptr;
    g.d_idx = nullptr;
    g.h_stage_v = g.h_stage_i = nullptr;
    g.stage_v_cap = g.stage_i_cap = 0;
    g.ev_stage_v = g.ev_stage_i = nullptr;
}

// (re)build the committed scene arrays (transformed vertices, rebased indices) on the device
int materialize_scene(ls_tracer *tr, bool with_maxabs)
{
    int rc;
    if ((rc = ensure(tr, tr->verts, (size_t)tr->n_verts * 3))) return rc;
    if ((rc = ensure(tr, tr->tris, (size_t)tr->n_tris * 3))) return rc;
    hipStream_t s = tr->stream;
    if (with_maxabs) LS_HIP(hipMemsetAsync(tr->d_maxabs, 0, 4, s));
    for (const auto &le : tr->layout) {
        auto it = tr->geoms.find(le.name);
        if (it == tr->geoms.end()) return fail(tr, LS_ERR_NOT_COMMITTED, "geometry removed since the last commit");
        Geometry &ge = it->second;
        ls::launch_transform(s, ge.raw(), ge.stride, ge.n_verts, ge.affine, tr->rinv, tr->t,
                             tr->verts.p + 3 * (size_t)le.vfirst, tr->d_maxabs);
        ls::launch_rebase(s, ge.idx(), ge.n_tris * 3, le.vfirst, tr->tris.p + 3 * (size_t)le.tfirst);
    }
    LS_HIP(hipGetLastError());
    tr->scene_materialized = true;
    return LS_OK;
}

// Group culling pays when k_project is bandwidth-bound.  Measured on MI355X (128 x 4096 rays, rocprofv3): at 10 M
// triangles the frame drops from 87 to 59 us (three frames in flight); at 1 M k_project itself drops from 21 to 15.6 us
// (43 % of the groups survive, its lanes are 2.3 x denser) but the cull pass in front of it takes ~10 us of pure
// latency (table staging, bound loads, one contended atomic per workgroup), and an 8-way azimuth shard's 8.5 + 9.7 us
// lose against 12.8 us without it.  auto = geometries of 2 M triangles or more.
bool cull_enabled(const ls_tracer *tr, const Geometry &g)
{
    if (tr->opt_block_cull != 2) return tr->opt_block_cull != 0;
    return g.n_tris >= 2000000u;
}

// Group-culling data of one geometry, brought up to date (stream-ordered on the handle's stream).
int prepare_blocks(ls_tracer *tr, Geometry &g)
{
    const bool want = cull_enabled(tr, g) && g.has_verts && g.has_idx && ls::project_tris_per_wave(g.n_tris) == 64u;
    if (!want) return LS_OK;
    if (!g.order_stale && !g.bounds_stale) return LS_OK;
    // frames in flight on the slot streams read d_idx_sorted / d_perm / d_boxes
    int rc;
    if ((rc = flush_pipeline(tr))) return rc;
    ++tr->main_epoch;
    const uint32_t nt = g.n_tris, ngroups = (nt + ls::kCullGroup - 1u) / ls::kCullGroup;
    if (!g.d_perm) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_perm), (size_t)nt * 4));
    if (!g.d_idx_sorted) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_idx_sorted), (size_t)nt * 12));
    if (!g.d_boxes) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_boxes), (size_t)ngroups * 32));
    if (g.order_stale) {
        if ((rc = ensure(tr, tr->keys_a, nt))) return rc;
        if ((rc = ensure(tr, tr->keys_b, nt))) return rc;
        if ((rc = ensure(tr, tr->vals_a, nt))) return rc;
        if ((rc = ensure(tr, tr->sort_temp, ls::sort_temp_bytes(nt)))) return rc;
        if (!tr->d_aabb6) LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_aabb6), 32));
        tr->bvh_order_valid = false;   // keys_a / keys_b / vals_a are the scratch of this pass
        ++tr->key_scratch_epoch;
        ls::launch_mesh_order(tr->stream, static_cast<const uint8_t *>(g.raw()), g.stride, g.n_verts, g.idx(), nt, tr->d_aabb6,
                              tr->keys_a.p, tr->keys_b.p, tr->vals_a.p, tr->sort_temp.p, tr->sort_temp.cap, g.d_perm, g.d_idx_sorted);
        g.order_stale = false;
        g.bounds_stale = true;
    }
    if (g.bounds_stale) {
        ls::launch_group_bounds(tr->stream, static_cast<const uint8_t *>(g.raw()), g.stride, g.d_idx_sorted, nt, g.d_boxes);
        g.bounds_stale = false;
    }
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- BVH engine, instanced mode (LS_OPT_BVH_INSTANCED) --------------------------------------------------------------
// mesh -> sensor of one geometry is p = Mlin v + Mtr with Mlin = Rinv A_lin, Mtr = Rinv (a - t).  Its inverse (double
// precision) gives the sensor origin and the direction map in mesh space; false if the matrix is (nearly) singular --
// such a scene takes the classic path (build / refit in the sensor frame).
bool inst_inverse(const ls_tracer *tr, const Geometry &ge, double *minv9, double *o3, double *cond)
{
    double M[9], tr3[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += (double)tr->rinv[3 * i + k] * (double)ge.affine[4 * k + j];
            M[3 * i + j] = acc;
        }
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) acc += (double)tr->rinv[3 * i + k] * ((double)ge.affine[4 * k + 3] - (double)tr->t[k]);
        tr3[i] = acc;
    }
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    double nM = 0.0;
    for (double v : M) nM += v * v;
    if (!(std::fabs(det) > 1e-12 * std::pow(nM, 1.5)) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    minv9[0] = c00 * id; minv9[1] = (M[2] * M[7] - M[1] * M[8]) * id; minv9[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    minv9[3] = c01 * id; minv9[4] = (M[0] * M[8] - M[2] * M[6]) * id; minv9[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    minv9[6] = c02 * id; minv9[7] = (M[1] * M[6] - M[0] * M[7]) * id; minv9[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    double nI = 0.0;
    for (int i = 0; i < 9; ++i) nI += minv9[i] * minv9[i];
    *cond = std::sqrt(nM * nI) / 3.0;   // 1 for a rotation
    for (int i = 0; i < 3; ++i) o3[i] = -(minv9[3 * i] * tr3[0] + minv9[3 * i + 1] * tr3[1] + minv9[3 * i + 2] * tr3[2]);
    for (int i = 0; i < 3; ++i) if (!std::isfinite(o3[i])) return false;
    return *cond < 1e3;
}

bool inst_possible(const ls_tracer *tr, const std::vector<Geometry *> &order)
{
    if (!tr->opt_bvh_instanced || order.size() > (size_t)ls::kGeomsPerLaunch) return false;
    for (const Geometry *ge : order) {
        double minv[9], o[3], cond;
        if (!inst_inverse(tr, *ge, minv, o, &cond)) return false;
    }
    return true;
}

// Hierarchies of the geometries whose vertices or topology changed (all of them after a layout change), each over its
// own slice of the shared key / record / node arrays, in MESH space: the same kernels as the classic build, fed with the
// vertices as uploaded.  A commit after which only poses differ finds nothing to do here.
int commit_instanced(ls_tracer *tr, const std::vector<Geometry *> &order, bool relayout)
{
    const uint32_t g = tr->leaf_size;
    const bool fresh = relayout || !tr->inst_valid || tr->inst_leaf_size != g || tr->inst_layout.size() != order.size();
    int rc;
    if (fresh) {
        tr->inst_layout.assign(order.size(), ls_tracer::InstSlot());
        uint32_t nodes = 0, recs = 0, range = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            ls_tracer::InstSlot &sl = tr->inst_layout[i];
            const uint32_t L = (order[i]->n_tris + g - 1) / g;
            sl.node_first = nodes; sl.rec_first = recs; sl.n_leaves = L; sl.range_first = range;
            std::memset(&sl.rt, 0, sizeof(sl.rt));
            uint32_t cnt = L, off = 0, lev = 0;
            while (true) {
                sl.rt.count[lev] = cnt; sl.rt.offset[lev] = off;
                off += cnt; ++lev;
                if (cnt <= 1) break;
                cnt = (cnt + 1) / 2;
            }
            sl.rt.levels = lev;
            nodes += L; recs += L * g; range += 2 * off + 2;
        }
        if ((rc = ensure(tr, tr->records, (size_t)recs))) return rc;
        if ((rc = ensure(tr, tr->nodes, (size_t)nodes + 1))) return rc;
        if ((rc = ensure(tr, tr->range_boxes, (size_t)range + 2))) return rc;
        tr->n_leaves = nodes;
        tr->inst_leaf_size = g;
    }
    bool any = false;
    for (const Geometry *ge : order) any = any || fresh || ge->blas_dirty;
    tr->last_commit_built = any;
    if (any) {
        if ((rc = ensure(tr, tr->inst_verts, (size_t)tr->n_verts * 3))) return rc;
        if ((rc = ensure(tr, tr->inst_tris, (size_t)tr->n_tris * 3))) return rc;
        if ((rc = ensure(tr, tr->keys_a, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->keys_b, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->vals_a, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->vals_b, tr->n_tris))) return rc;
        uint32_t biggest = 0;
        for (const Geometry *ge : order) biggest = std::max(biggest, ge->n_tris);
        if ((rc = ensure(tr, tr->sort_temp, ls::sort_temp_bytes(biggest)))) return rc;
        if (!tr->d_inst_maxabs) LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_inst_maxabs), ls::kGeomsPerLaunch * 4));
        tr->bvh_order_valid = false;   // the key arrays hold per-geometry slices now
        if (fresh) ++tr->key_scratch_epoch;   // the slices moved
        hipStream_t s = tr->stream;
        static const float kIdA[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, kIdR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, kZero[3] = {0, 0, 0};
        LS_HIP(hipMemsetAsync(tr->d_inst_maxabs, 0, ls::kGeomsPerLaunch * 4, s));
        for (size_t i = 0; i < order.size(); ++i) {
            Geometry &ge = *order[i];
            if (!fresh && !ge.blas_dirty) continue;
            const ls_tracer::LayoutEntry &le = tr->layout[i];
            const ls_tracer::InstSlot &sl = tr->inst_layout[i];
            float *verts = tr->inst_verts.p + 3 * (size_t)le.vfirst;
            uint32_t *tris = tr->inst_tris.p + 3 * (size_t)le.tfirst;
            uint32_t *ka = tr->keys_a.p + le.tfirst, *kb = tr->keys_b.p + le.tfirst, *va = tr->vals_a.p + le.tfirst, *vb = tr->vals_b.p + le.tfirst;
            // identity transform: the packed copy holds the vertices as uploaded (1 * x + 0 * y + 0 * z + 0 is x)
            ls::launch_transform(s, ge.raw(), ge.stride, ge.n_verts, kIdA, kIdR, kZero, verts, tr->d_inst_maxabs + i);
            ls::launch_rebase(s, ge.idx(), ge.n_tris * 3, 0u, tris);
            // vertices alone changed and the geometry's sorted keys are still in place: a refit (same order, same
            // topology, every box recomputed) -- what the classic path does with LS_OPT_BVH_REFIT
            const bool refit = tr->opt_bvh_refit && !ge.blas_topo_dirty && ge.blas_sorted_epoch == tr->key_scratch_epoch;
            if (!refit) {
                ls::launch_morton(s, verts, tris, ge.n_tris, tr->d_inst_maxabs + i, ka, va);
                ls::launch_sort(s, tr->sort_temp.p, tr->sort_temp.cap, ka, kb, va, vb, ge.n_tris);
                ge.blas_sorted_epoch = tr->key_scratch_epoch;
            }
            float4 *rb = tr->range_boxes.p + sl.range_first;
            ls::launch_leaves(s, verts, tris, vb, ge.n_tris, g, tr->records.p + sl.rec_first, rb, true);
            ls::launch_range_tree(s, sl.rt, rb);
            ls::launch_hierarchy(s, kb, sl.n_leaves, g, sl.rt, rb, tr->nodes.p + sl.node_first);
        }
        // a scene of one geometry: the top of its hierarchy for the trace grid's LDS (static with the hierarchy)
        tr->treelet_valid = false;
        static const bool no_treelet = getenv("LS_TRACE_NO_TREELET") != nullptr;
        if (order.size() == 1 && tr->inst_layout[0].n_leaves > 1u && !no_treelet) {
            if ((rc = ensure(tr, tr->treelet, (size_t)ls::kTreeletNodes))) return rc;
            LS_HIP(hipMemsetAsync(tr->treelet.p, 0, (size_t)ls::kTreeletNodes * sizeof(ls::FatNode), s));
            ls::launch_treelet(s, tr->nodes.p + tr->inst_layout[0].node_first, tr->inst_layout[0].n_leaves, tr->treelet.p);
            tr->treelet_valid = true;
        }
        LS_HIP(hipGetLastError());
        // the extent of every rebuilt mesh (the widening of its boxes at trace time is scaled by it)
        uint32_t bits[ls::kGeomsPerLaunch];
        LS_HIP(hipMemcpyAsync(bits, tr->d_inst_maxabs, sizeof(bits), hipMemcpyDeviceToHost, s));
        LS_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < order.size(); ++i) {
            Geometry &ge = *order[i];
            if (!fresh && !ge.blas_dirty) continue;
            std::memcpy(&ge.mesh_maxabs, &bits[i], 4);
            ge.blas_dirty = ge.blas_topo_dirty = false;
        }
    }
    tr->inst_valid = true;
    tr->bvh_inst = true;
    tr->bvh_built = true;
    tr->scene_materialized = false;   // tr->verts / tr->tris (sensor frame) were not made: the debug views make them on demand
    return LS_OK;
}

int commit_locked(ls_tracer *tr)
{
    tr->committed = false;
    tr->traced = false;
    tr->bvh_built = false;
    tr->scene_materialized = false;
    // layout: geometries with data, in geomID order, so that the global triangle id orders
    // triangles by (geomID, primID) -- the tie-break key of equal-t hits.
    std::vector<Geometry *> order;
    for (auto &kv : tr->geoms)
        if (kv.second.has_verts && kv.second.has_idx && kv.second.n_tris > 0) order.push_back(&kv.second);
    std::sort(order.begin(), order.end(), [](const Geometry *a, const Geometry *b) { return a->id < b->id; });
    if (order.empty()) {
        tr->n_tris = tr->n_verts = tr->n_leaves = tr->n_slots = 0;
        tr->slot_geom_ids.clear();
        tr->slot_tri_first.assign(1, 0u);
        tr->layout.clear();
        tr->layout_dirty = true;
        return -1;  // OptixTracer.cpp:266-267
    }
    if (order.size() > (size_t)ls::kMaxGeoms) return fail(tr, LS_ERR_OUT_OF_RANGE, "too many geometries");

    std::vector<uint32_t> vfirst(order.size() + 1, 0u), tfirst(order.size() + 1, 0u);
    std::vector<int> ids(order.size());
    tr->layout.clear();
    for (size_t k = 0; k < order.size(); ++k) {
        vfirst[k + 1] = vfirst[k] + order[k]->n_verts;
        tfirst[k + 1] = tfirst[k] + order[k]->n_tris;
        ids[k] = order[k]->id;
        tr->layout.push_back({order[k]->name, vfirst[k], tfirst[k]});
    }
    const bool relayout = tr->layout_dirty || ids != tr->slot_geom_ids || tfirst != tr->slot_tri_first ||
                          tr->leaf_size != tr->committed_leaf_size;
    const uint32_t nv = vfirst.back(), nt = tfirst.back();
    const uint32_t g = tr->leaf_size;
    const uint32_t L = (nt + g - 1) / g;
    tr->n_verts = nv;
    tr->n_tris = nt;

    int rc;
    if (relayout) {
        // frames still in flight on the second stream read the old table
        if ((rc = flush_pipeline(tr))) return rc;
        LS_HIP(hipStreamSynchronize(tr->stream));
        std::vector<uint32_t> table(tfirst);
        for (int id : ids) table.push_back((uint32_t)id);
        for (const Geometry *ge : order) table.push_back(ge->quad ? 1u : 0u);   // primID = triangle >> shift
        if ((rc = ensure(tr, tr->geom_table, table.size()))) return rc;
        LS_HIP(hipMemcpyAsync(tr->geom_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice, tr->stream));
        LS_HIP(hipStreamSynchronize(tr->stream));  // `table` is a stack temporary
        // aligned-range tree geometry
        ls::RangeTree &rt = tr->rt;
        std::memset(&rt, 0, sizeof(rt));
        uint32_t cnt = L, off = 0, lev = 0;
        while (true) {
            rt.count[lev] = cnt;
            rt.offset[lev] = off;
            off += cnt;
            ++lev;
            if (cnt <= 1) break;
            cnt = (cnt + 1) / 2;
        }
        rt.levels = lev;
        tr->range_entries = off;
    }

    hipStream_t s = tr->stream;
    const bool want_bvh = !use_projection(tr);
    mark(tr, 0);
    if (want_bvh && inst_possible(tr, order)) {
        // BVH engine, instanced mode: per-geometry hierarchies in mesh space; nothing to do when only poses changed
        if ((rc = commit_instanced(tr, order, relayout))) return rc;
        mark(tr, 6);
    } else if (want_bvh) {
        // BVH engine: transform every geometry into the sensor frame, then the LBVH: a full build (Morton keys, radix
        // sort, leaves, range tree, hierarchy), or -- when only vertices / poses changed since the last build, the case
        // OptixTracer handles with OPTIX_BUILD_OPERATION_UPDATE (OptixTracer.cpp:532-535) -- a REFIT: the triangles keep
        // their Morton order and the radix tree its topology (both come from the sorted keys, which stay), records and
        // every box are recomputed from the new vertices (leaf boxes, aligned-range tree, both child boxes of every
        // node by range query).  Always a valid BVH; the sort and the key pass (half of the build) are not run.
        tr->bvh_inst = false;
        tr->inst_valid = false;        // the shared record / node arrays are about to hold the sensor-frame hierarchy
        ++tr->key_scratch_epoch;
        tr->last_commit_built = true;
        bool any_idx_dirty = false;
        for (const Geometry *ge : order) any_idx_dirty = any_idx_dirty || ge->idx_dirty;
        const bool refit = tr->opt_bvh_refit && !relayout && !any_idx_dirty && tr->bvh_order_valid && tr->bvh_order_tris == nt;
        if ((rc = ensure(tr, tr->keys_a, nt))) return rc;
        if ((rc = ensure(tr, tr->keys_b, nt))) return rc;
        if ((rc = ensure(tr, tr->vals_a, nt))) return rc;
        if ((rc = ensure(tr, tr->vals_b, nt))) return rc;
        if ((rc = ensure(tr, tr->records, (size_t)L * g))) return rc;
        if ((rc = ensure(tr, tr->nodes, (size_t)L))) return rc;
        if ((rc = ensure(tr, tr->sort_temp, ls::sort_temp_bytes(nt)))) return rc;
        if ((rc = ensure(tr, tr->range_boxes, 2 * (size_t)tr->range_entries + 2))) return rc;
        if ((rc = materialize_scene(tr, true))) return rc;
        mark(tr, 1);
        if (!refit) ls::launch_morton(s, tr->verts.p, tr->tris.p, nt, tr->d_maxabs, tr->keys_a.p, tr->vals_a.p);
        mark(tr, 2);
        if (!refit) ls::launch_sort(s, tr->sort_temp.p, tr->sort_temp.cap, tr->keys_a.p, tr->keys_b.p, tr->vals_a.p, tr->vals_b.p, nt);
        mark(tr, 3);
        tr->bvh_order_valid = true;
        tr->bvh_order_tris = nt;
        tr->last_commit_refit = refit;
        ls::launch_leaves(s, tr->verts.p, tr->tris.p, tr->vals_b.p, nt, g, tr->records.p, tr->range_boxes.p);
        mark(tr, 4);
        ls::launch_range_tree(s, tr->rt, tr->range_boxes.p);
        mark(tr, 5);
        ls::launch_hierarchy(s, tr->keys_b.p, L, g, tr->rt, tr->range_boxes.p, tr->nodes.p);
        mark(tr, 6);
        LS_HIP(hipGetLastError());
        tr->bvh_built = true;
        for (Geometry *ge : order) ge->blas_dirty = ge->blas_topo_dirty = true;   // whatever instanced hierarchies there were are overwritten
    } else {
        // projection engine: no hierarchy to build -- the trace kernel streams the meshes as uploaded and applies
        // the vertex transform on the fly.  Big meshes keep a Morton order (per topology) and per-block bounds
        // (per vertex upload) so that k_cull can drop whole 64-triangle blocks before their indices are read.
        for (Geometry *ge : order)
            if ((rc = prepare_blocks(tr, *ge))) return rc;
        mark(tr, 1);
    }
    for (Geometry *ge : order) ge->idx_dirty = false;

    tr->n_leaves = L;
    tr->n_slots = L - 1;  // BVH2 nodes (64 B each)
    tr->committed_leaf_size = g;
    tr->slot_geom_ids = ids;
    tr->slot_tri_first = tfirst;
    tr->layout_dirty = false;
    tr->committed = true;
    return LS_OK;
}

int trace_locked(ls_tracer *tr, uint32_t frame, ls_frame *out, bool readback)
{
    if (!out) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null frame");
    std::memset(out, 0, sizeof(*out));
    out->frame = frame;
    out->n_rays = shard_rays(tr);
    tr->traced = false;
    if (!tr->committed || tr->n_tris == 0) return -1;  // OptixTracer.cpp:280-288: cleared cloud, -1
    int rc;
    if ((rc = ensure_outputs(tr))) return rc;
    if (tr->ext_points && tr->ext_capacity < shard_rays(tr))
        return fail(tr, LS_ERR_OUT_OF_RANGE, "external output buffers smaller than the shard's ray count");

    hipStream_t s = tr->stream;   // three-stream mode switches to the frame's own stream below
    const ls::SensorTables tb = tables(tr);
    uint8_t *d_points = tr->ext_points ? static_cast<uint8_t *>(tr->ext_points) : tr->points.p;
    void *d_hits = tr->ext_points ? tr->ext_hits : static_cast<void *>(tr->hits.p);
    uint32_t *d_n = tr->ext_points ? tr->ext_n_points : tr->d_n_points;
    // synchronous call with the library's own outputs: the pack kernel writes the pinned host buffers itself
    const bool hv = readback && tr->opt_host_output && !tr->ext_points;
    if (hv && (rc = ensure_host_buffers(tr, shard_rays(tr)))) return rc;
    const uint32_t compact = hv && tr->opt_host_output == 2 ? 1u : 0u;
    auto host_targets = [&]() {
        if (!hv) return;
        d_points = tr->h_points;
        if (tr->opt_readback_hits) d_hits = tr->h_hits;
        d_n = tr->h_n_points;
    };
    host_targets();
    if (tr->opt_count) LS_HIP(hipMemsetAsync(tr->d_visits, 0, 32, s));
    if (use_projection(tr)) {
        // sensor-space projection engine: stream the triangles once, test only the covered rays
        ls::ProjectParams pp = project_params(tr);
        unsigned long long *stats = tr->opt_count ? tr->d_visits + 1 : nullptr;  // counts[1] = triangle tests
        const uint32_t n_blocks = (shard_rays(tr) + 255u) / 256u;
        const bool pipelined = tr->opt_pipeline == 1 && !tr->opt_count && !tr->opt_timing;
        const bool multi = tr->opt_pipeline == 2 && !tr->opt_count && !tr->opt_timing;
        if (!pipelined && !multi && (rc = flush_pipeline(tr))) return rc;
        {   // spread runs balance the cells per wave (shorter kernel: 23.6 -> 21.2 us alone) but touch more cache lines,
            // which costs more than it gains once three frames overlap (16.9 -> 17.2 us per frame): LS_PROJECT_SPREAD overrides
            static const int spread_env = getenv("LS_PROJECT_SPREAD") ? atoi(getenv("LS_PROJECT_SPREAD")) : -1;
            pp.spread = spread_env >= 0 ? spread_env : (multi ? 0 : 1);
        }
        // which set of keys / queue / outputs and which of the three queue counters this frame uses.
        // Rider mode and frames that are not pipelined: pipe_seq counts the pipelined frames; a frame that is
        // not pipelined re-arms what it used itself and leaves pipe_seq alone, so the rotation stays
        // consistent across mode changes.  Three-stream mode: slot = frame number mod 3.
        const uint32_t slot = multi ? tr->ms_seq % 3u : (tr->pipe_seq & 1u);
        unsigned long long *keys = slot == 0 ? tr->best_keys.p : (slot == 1 ? tr->best_keys_b.p : tr->best_keys_c.p);
        void *bigq = slot == 0 ? static_cast<void *>(tr->big_queue.p)
                               : (slot == 1 ? static_cast<void *>(tr->big_queue_b.p) : static_cast<void *>(tr->big_queue_c.p));
        uint32_t *big_count = tr->d_big_count + ls::kCounterSlotWords * (multi ? slot : tr->pipe_seq % 3u);
        if (slot && !tr->ext_points) {
            d_points = slot == 1 ? tr->points_b.p : tr->points_c.p;
            d_hits = slot == 1 ? tr->hits_b.p : tr->hits_c.p;
            d_n = slot == 1 ? tr->d_n_points_b : tr->d_n_points_c;
        }
        host_targets();
        if (!tr->keys_armed) {
            // first frame (or a new shard / raster): key set 0, all queue counters, the block counts.  In three-
            // stream mode the other streams' frames use those counters too: the initialisation completes first
            ls::launch_project_init(s, pp, tr->best_keys.p, tr->d_big_count, tr->row_counts.p);
            if (multi) LS_HIP(hipStreamSynchronize(s));
            tr->keys_armed = true;
            tr->frame_parity = 0;
        }
        if (multi) {
            // the frame's stream first sees what is enqueued on the handle's stream: the library's own mesh copies,
            // or anything at all when the stream is the caller's (host API calls are not cheap: only when needed)
            const bool dep = tr->slot_epoch[slot] != tr->main_epoch || tr->stream != tr->own_stream;
            if (dep) LS_HIP(hipEventRecord(tr->ev_main, s));
            s = tr->slot_stream[slot];
            if (dep) LS_HIP(hipStreamWaitEvent(s, tr->ev_main, 0));
            tr->slot_epoch[slot] = tr->main_epoch;
        }
        if (slot == 1 && !tr->keys_b_armed) {
            LS_HIP(hipMemsetAsync(tr->best_keys_b.p, 0xFF, (size_t)shard_rays(tr) * 8, s));
            tr->keys_b_armed = true;
        }
        if (slot == 2 && !tr->keys_c_armed) {
            LS_HIP(hipMemsetAsync(tr->best_keys_c.p, 0xFF, (size_t)shard_rays(tr) * 8, s));
            tr->keys_c_armed = true;
        }
        uint32_t *counts = tr->row_counts.p + (size_t)(multi ? slot : tr->frame_parity) * n_blocks;
        uint32_t *next_counts = tr->row_counts.p + (size_t)(multi ? 3u : 1u - tr->frame_parity) * n_blocks;
        // LS_OPT_TIMING = 2 (the dominant kernel alone): the two events ride on the k_project dispatch
        hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
        const bool ride = tr->opt_timing == 2;
        mark(tr, 7, ride ? &ev_k0 : nullptr);
        std::vector<ls::GeomSource> &srcs = tr->project_srcs;
        srcs.clear();
        bool any_culled = false;
        for (const auto &le : tr->layout) {
            auto it = tr->geoms.find(le.name);
            if (it == tr->geoms.end()) return fail(tr, LS_ERR_NOT_COMMITTED, "geometry removed since the last commit");
            const Geometry &ge = it->second;
            ls::GeomSource src;
            src.verts = static_cast<const uint8_t *>(ge.raw());
            src.stride = ge.stride;
            src.idx = ge.idx();
            src.ntris = ge.n_tris;
            src.gid_first = le.tfirst;
            static const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            src.xform = std::memcmp(ge.affine, kIdentity, sizeof(kIdentity)) == 0 ? 2 : 1;
            std::memcpy(src.m.a, ge.affine, sizeof(src.m.a));
            std::memcpy(src.m.rinv, tr->rinv, sizeof(src.m.rinv));
            std::memcpy(src.m.t, tr->t, sizeof(src.m.t));
            // block culling data is current (prepare_blocks ran at the commit) unless an update came in since
            const bool culled = cull_enabled(tr, ge) && ge.d_boxes && !ge.order_stale && !ge.bounds_stale &&
                                ls::project_tris_per_wave(ge.n_tris) == 64u;
            src.perm = culled ? ge.d_perm : nullptr;
            src.boxes = culled ? ge.d_boxes : nullptr;
            if (culled) src.idx = ge.d_idx_sorted;
            any_culled = any_culled || culled;
            srcs.push_back(src);
        }
        // survivor list of this frame's k_cull: one of three (as many frames as can be in flight)
        uint32_t *cull_list = nullptr;
        if (any_culled) {
            const uint32_t entries = ls::project_cull_entries(srcs.data(), (uint32_t)srcs.size());
            if (entries) {
                if (entries > tr->cull_chunks) {
                    if ((rc = flush_pipeline(tr))) return rc;
                    LS_HIP(hipStreamSynchronize(tr->stream));
                    if ((rc = ensure(tr, tr->cull_list, 3 * (size_t)entries))) return rc;
                    tr->cull_chunks = (uint32_t)(tr->cull_list.cap / 3);
                }
                cull_list = tr->cull_list.p + (size_t)(multi ? slot : tr->pipe_seq % 3u) * tr->cull_chunks;
            }
        }
        ls::GeomTable gt;
        gt.n = (uint32_t)tr->slot_geom_ids.size();
        gt.tri_first = tr->geom_table.p;
        gt.geom_ids = tr->geom_table.p + gt.n + 1;
        gt.prim_shift = gt.geom_ids + gt.n;
        if (pipelined) {
            // one launch: this frame's k_project workgroups + the previous frame's finish + pack workgroups
            ls::launch_project(s, pp, srcs.data(), (uint32_t)srcs.size(), keys, bigq, tr->big_capacity, big_count, nullptr,
                               tr->pipe_pending ? &tr->pipe_fa : nullptr, cull_list);
            if (++tr->pack_epoch == 0u) {   // the epoch tag wrapped: no stale status word may match
                LS_HIP(hipMemsetAsync(tr->pack_status.p, 0, tr->pack_status.cap * 8, s));
                tr->pack_epoch = 1u;
            }
            ls::FinishPackArgs &fa = tr->pipe_fa;   // this frame's, launched with the next frame or by a flush
            fa.best = keys;
            fa.big = bigq;
            fa.big_capacity = tr->big_capacity;
            fa.big_count = big_count;
            fa.rearm_big_count = tr->d_big_count + ls::kCounterSlotWords * ((tr->pipe_seq + 2u) % 3u);
            fa.status = tr->pack_status.p;
            fa.epoch = tr->pack_epoch;
            fa.publish_epoch = tr->pack_epoch;
            fa.spin_limit = 1u << 18;   // ~1 s of backed-off polls
            fa.device_status = tr->h_status;
            if (tr->opt_debug_fault) {   // LS_OPT_DEBUG_FAULT: this frame publishes a tag nobody waits for
                fa.publish_epoch = tr->pack_epoch ^ 0x40000000u;
                fa.spin_limit = 1u << 6;
                tr->opt_debug_fault = 0;
            }
            fa.gt = gt;
            fa.points32 = d_points;
            fa.hits = d_hits;
            fa.n_points = d_n;
            fa.n_blocks = n_blocks;
            fa.compact = compact;
            tr->pipe_pending = true;
            ++tr->pipe_seq;
            if (readback && (rc = flush_pipeline(tr))) return rc;
        } else {
            // one launch per 16 geometries (the descriptors travel as kernel arguments)
            if (ride) mark(tr, 8, &ev_k1);
            ls::launch_project(s, pp, srcs.data(), (uint32_t)srcs.size(), keys, bigq, tr->big_capacity, big_count, stats, nullptr, cull_list,
                               ev_k0, ev_k1);
            if (!ride) mark(tr, 8);
            ls::launch_project_finish(s, pp, keys, bigq, tr->big_capacity, big_count, counts, stats);
            mark(tr, 9);
            ls::launch_pack_keys(s, tb, keys, tr->hit_t.p, tr->hit_gid.p, counts, next_counts, big_count, gt, d_points, d_hits, d_n, compact);
            mark(tr, 10);
            if (multi) {
                // no event per frame: a flush (or a mesh copy) records one per stream and orders the handle's stream after it
                tr->slot_pending[slot] = true;
                ++tr->ms_seq;
                s = tr->stream;
                if (readback && (rc = flush_pipeline(tr))) return rc;
            } else {
                tr->frame_parity ^= 1u;
            }
        }
        tr->traced_projection = true;
        tr->last_d_hits = d_hits;
        tr->last_d_n = d_n;
    } else {
        if (!tr->bvh_built) return fail(tr, LS_ERR_NOT_COMMITTED, "the BVH engine was selected after the last commit");
        if ((rc = flush_pipeline(tr))) return rc;
        LS_HIP(hipMemsetAsync(tr->d_queue_heads, 0, ls::kQueues * 16 * sizeof(uint32_t), s));
        ls::RayQueues rq;
        rq.heads = tr->d_queue_heads;
        rq.chan_mul = tr->chan_mul;
        rq.refill_min = tr->refill_min;
        rq.chan_order = reinterpret_cast<const uint32_t *>(tr->d_tables + 4 * (size_t)tr->V + 2 * (size_t)tr->H);   // chan_perm (fill_tables)
        { static const bool no_order = getenv("LS_TRACE_NO_ORDER") != nullptr; if (no_order) rq.chan_order = nullptr; }
        mark(tr, 7);
        if (tr->bvh_inst) {
            // this frame's poses: every geometry's ray map (inverse of mesh -> sensor) and exact transform
            ls::InstBatch batch;
            batch.n = (uint32_t)tr->layout.size();
            for (uint32_t i = 0; i < batch.n; ++i) {
                auto it = tr->geoms.find(tr->layout[i].name);
                if (it == tr->geoms.end()) return fail(tr, LS_ERR_NOT_COMMITTED, "geometry removed since the last commit");
                const Geometry &ge = it->second;
                const ls_tracer::InstSlot &sl = tr->inst_layout[i];
                ls::InstGeom &ig = batch.g[i];
                double minv[9], o[3], cond = 1.0;
                const bool ok = !ge.blas_dirty && inst_inverse(tr, ge, minv, o, &cond);
                if (!ok) return fail(tr, LS_ERR_NOT_COMMITTED, "a geometry or pose changed since the last commit");
                ig.node_first = sl.node_first;
                ig.rec_first = sl.rec_first;
                ig.n_leaves = sl.n_leaves;
                ig.n_tris = ge.n_tris;
                ig.gid_first = tr->layout[i].tfirst;
                static const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
                ig.xform = std::memcmp(ge.affine, kIdentity, sizeof(kIdentity)) == 0 ? 2 : 1;
                float omax = 0.0f;
                for (int k = 0; k < 3; ++k) { ig.o[k] = (float)o[k]; omax = std::max(omax, std::fabs(ig.o[k])); }
                for (int k = 0; k < 9; ++k) ig.minv[k] = (float)minv[k];
                // rounding of o (6e-8 |o|) and of minv * d along the way to any box (<= |o| + the mesh's extent), times
                // the conditioning of the map, with a factor of ten in hand
                ig.eps = 4e-6f * (float)std::max(1.0, cond) * (omax + 2.0f * ge.mesh_maxabs);
                std::memcpy(ig.m.a, ge.affine, sizeof(ig.m.a));
                std::memcpy(ig.m.rinv, tr->rinv, sizeof(ig.m.rinv));
                std::memcpy(ig.m.t, tr->t, sizeof(ig.m.t));
            }
            ls::launch_trace_instanced(s, tr->trace_blocks, tb, rq, batch, tr->nodes.p, tr->records.p, tr->inst_leaf_size,
                                       (batch.n == 1u && tr->treelet_valid) ? tr->treelet.p : nullptr, tr->hit_t.p, tr->hit_gid.p, tr->spill.p, tr->opt_count ? tr->d_visits : nullptr);
        } else {
            ls::launch_trace(s, tr->trace_blocks, tb, rq, tr->nodes.p, tr->records.p, tr->n_leaves, tr->committed_leaf_size,
                             tr->n_tris, tr->hit_t.p, tr->hit_gid.p, tr->spill.p, tr->opt_count ? tr->d_visits : nullptr);
        }
        mark(tr, 8);
        ls::launch_rowcount(s, tr->hit_gid.p, shard_rays(tr), tr->row_counts.p);
        tr->keys_armed = false;  // the counter array was just used with the BVH layout
        mark(tr, 9);
        ls::GeomTable gt;
        gt.n = (uint32_t)tr->slot_geom_ids.size();
        gt.tri_first = tr->geom_table.p;
        gt.geom_ids = tr->geom_table.p + gt.n + 1;
        gt.prim_shift = gt.geom_ids + gt.n;
        ls::launch_pack(s, tb, tr->hit_t.p, tr->hit_gid.p, tr->row_counts.p, gt, d_points, d_hits, d_n, compact);
        tr->traced_projection = false;
        mark(tr, 10);
    }
    LS_HIP(hipGetLastError());
    tr->traced = true;
    out->d_points32 = d_points;
    out->d_hits = d_hits;
    out->d_n_points = d_n;
    if (!readback) return LS_OK;

    if (hv) {
        LS_HIP(hipStreamSynchronize(s));   // the only host wait of the frame
        if ((rc = check_device_status(tr))) return rc;
        out->n_points = *tr->h_n_points;
        out->points32 = compact ? nullptr : tr->h_points;
        out->compact16 = compact ? tr->h_points : nullptr;
        out->hits = tr->opt_readback_hits ? tr->h_hits : nullptr;
        return LS_OK;
    }
    LS_HIP(hipMemcpyAsync(tr->h_n_points, d_n, 4, hipMemcpyDeviceToHost, s));
    LS_HIP(hipStreamSynchronize(s));
    if ((rc = check_device_status(tr))) return rc;
    const uint32_t n = *tr->h_n_points;
    if ((rc = ensure_host_buffers(tr, std::max<size_t>(shard_rays(tr), n)))) return rc;
    if (n) {
        LS_HIP(hipMemcpyAsync(tr->h_points, d_points, (size_t)n * 32, hipMemcpyDeviceToHost, s));
        if (tr->opt_readback_hits) LS_HIP(hipMemcpyAsync(tr->h_hits, d_hits, (size_t)n * 16, hipMemcpyDeviceToHost, s));
        LS_HIP(hipStreamSynchronize(s));
    }
    out->n_points = n;
    out->points32 = tr->h_points;
    out->hits = tr->opt_readback_hits ? tr->h_hits : nullptr;
    return LS_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int ls_abi_version(void) { return LS_ABI_VERSION; }

// shared tail of the two create calls: `tr` holds the sensor (V, H, vertical, h_begin / h_step, pose and, for
// ls_tracer_create_tables, the given factor tables)
static int create_device_state(ls_tracer *tr, int hip_device, ls_tracer **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hip_device < 0 || hip_device >= ndev) {
        delete tr;
        return LS_ERR_NO_DEVICE;
    }
    if (hipSetDevice(hip_device) != hipSuccess) {
        delete tr;
        return LS_ERR_NO_DEVICE;
    }
    tr->device = hip_device;
    tr->az0 = 0;
    tr->naz = tr->H;
    auto bail = [&](int code) {
        ls_tracer_destroy(tr);
        return code;
    };
    if (hipStreamCreateWithFlags(&tr->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(LS_ERR_HIP);
    tr->stream = tr->own_stream;
    std::vector<float> tab;
    fill_tables(tr, tab);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_tables), tab.size() * 4) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMemcpy(tr->d_tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_maxabs), 4) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_visits), 32) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_n_points), 4) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_queue_heads), ls::kQueues * 16 * sizeof(uint32_t)) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_big_count), 4 * ls::kCounterSlotWords * sizeof(uint32_t)) != hipSuccess) return bail(LS_ERR_HIP);
    for (float chi : tr->vertical)
        if (!(chi >= -90.0f && chi <= 90.0f)) tr->projection_ok = false;  // elevation == channel angle only there
    if (!std::isfinite(tr->h_begin) || !std::isfinite(tr->h_end)) tr->projection_ok = false;
    if (tr->V > 32767u) tr->projection_ok = false;  // channel range is packed into 16+15 bits
    tr->trace_blocks = ls::trace_grid_blocks(hip_device);
    {
        // Ray order of the persistent trace grid.  Round 1 visited the channels at a stride near 0.38 V (cheap sky channels
        // and expensive grazing ones alternate in every queue) and refilled 24 idle lanes at a time.  Measured again in
        // round 2 (bench.py --engine bvh, k_trace_inst and k_trace alike): rings one after the other, from the highest
        // elevation down (position (j * chan_mul) % V of the elevation order with chan_mul = 1: a wave's next 64 rays
        // pass the nodes its last 64 did), whole-wave refills (idle lanes wait until 56 of 64 are idle) and 2 resident
        // blocks per CU: 0.150 ms against 0.186.  The other direction, lowest ring first, takes 0.194 ms; tiles of
        // 4 x 16 or 8 x 8 rays instead of 64 x 1 take 0.156 / 0.159.
        auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
        tr->chan_mul = 1u;
        tr->refill_min = 56u;
        if (const char *e = getenv("LS_TRACE_CHAN_MUL")) { const uint32_t v = (uint32_t)atoi(e); if (v && gcd(v, tr->V) == 1u) tr->chan_mul = v; }
        if (const char *e = getenv("LS_TRACE_REFILL_MIN")) { const int v = atoi(e); if (v >= 1 && v <= 64) tr->refill_min = (uint32_t)v; }
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&tr->h_n_points), 16) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipHostMalloc(reinterpret_cast<void **>(&tr->h_status), 16) != hipSuccess) return bail(LS_ERR_HIP);
    *tr->h_status = 0u;
    tr->slot_tri_first.assign(1, 0u);
    *out = tr;
    return LS_OK;
}

int ls_tracer_create(const ls_sensor_desc *sd, int hip_device, ls_tracer **out)
{
    if (!out) return LS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!sd || !sd->vertical_deg || sd->n_vertical == 0 || sd->h_count < 2) return LS_ERR_INVALID_ARGUMENT;
    if ((unsigned long long)sd->n_vertical * sd->h_count > 0x7FFFFFFFull) return LS_ERR_OUT_OF_RANGE;  // ray indices are 32-bit
    ls_tracer *tr = new ls_tracer();
    tr->V = sd->n_vertical;
    tr->H = sd->h_count;
    tr->vertical.assign(sd->vertical_deg, sd->vertical_deg + sd->n_vertical);
    tr->h_begin = sd->h_begin;
    tr->h_end = sd->h_end;
    tr->h_step = (sd->h_end - sd->h_begin) / static_cast<float>(sd->h_count - 1u);  // LidarDevice.cpp:611
    std::memcpy(tr->rinv, sd->Rinv, sizeof(tr->rinv));
    std::memcpy(tr->t, sd->t, sizeof(tr->t));
    return create_device_state(tr, hip_device, out);
}

int ls_tracer_create_tables(const ls_sensor_tables *st, int hip_device, ls_tracer **out)
{
    if (!out) return LS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!st || !st->sin_theta || !st->cos_theta || !st->elevation_deg || !st->sin_phi || !st->cos_phi || st->n_vertical == 0 ||
        st->h_count < 2)
        return LS_ERR_INVALID_ARGUMENT;
    if ((unsigned long long)st->n_vertical * st->h_count > 0x7FFFFFFFull) return LS_ERR_OUT_OF_RANGE;
    ls_tracer *tr = new ls_tracer();
    const uint32_t V = tr->V = st->n_vertical, H = tr->H = st->h_count;
    tr->vertical.assign(st->elevation_deg, st->elevation_deg + V);
    tr->h_begin = st->h_begin_deg;
    tr->h_step = st->h_step_deg;
    tr->h_end = st->h_begin_deg + st->h_step_deg * static_cast<float>(H - 1u);
    tr->given_tables.resize(2 * (size_t)V + 2 * (size_t)H);
    std::memcpy(tr->given_tables.data(), st->sin_theta, V * sizeof(float));
    std::memcpy(tr->given_tables.data() + V, st->cos_theta, V * sizeof(float));
    std::memcpy(tr->given_tables.data() + 2 * (size_t)V, st->sin_phi, H * sizeof(float));
    std::memcpy(tr->given_tables.data() + 2 * (size_t)V + H, st->cos_phi, H * sizeof(float));
    std::memcpy(tr->rinv, st->Rinv, sizeof(tr->rinv));
    std::memcpy(tr->t, st->t, sizeof(tr->t));
    return create_device_state(tr, hip_device, out);
}

void ls_tracer_destroy(ls_tracer *tr)
{
    if (!tr) return;
    (void)hipSetDevice(tr->device);
    if (tr->stream) (void)hipStreamSynchronize(tr->stream);
    for (auto &kv : tr->geoms) free_geometry(kv.second);
    for (int i = 0; i < 3; ++i)
        if (tr->slot_stream[i]) (void)hipStreamSynchronize(tr->slot_stream[i]);
    release(tr->best_keys_b); release(tr->big_queue_b); release(tr->points_b); release(tr->hits_b); release(tr->pack_status);
    release(tr->best_keys_c); release(tr->big_queue_c); release(tr->points_c); release(tr->hits_c);
    if (tr->d_n_points_b) (void)hipFree(tr->d_n_points_b);
    if (tr->d_n_points_c) (void)hipFree(tr->d_n_points_c);
    if (tr->ev_main) (void)hipEventDestroy(tr->ev_main);
    for (int i = 0; i < 3; ++i) {
        if (tr->ev_done[i]) (void)hipEventDestroy(tr->ev_done[i]);
        bool dup = false;
        for (int k = 0; k < i; ++k) dup = dup || tr->slot_stream[k] == tr->slot_stream[i];
        if (tr->slot_stream[i] && !dup) (void)hipStreamDestroy(tr->slot_stream[i]);
    }
    release(tr->verts); release(tr->tris); release(tr->keys_a); release(tr->keys_b); release(tr->vals_a);
    release(tr->vals_b); release(tr->geom_table); release(tr->sort_temp); release(tr->records);
    release(tr->inst_verts); release(tr->inst_tris); release(tr->treelet);
    if (tr->d_inst_maxabs) (void)hipFree(tr->d_inst_maxabs);
    release(tr->nodes); release(tr->range_boxes); release(tr->hit_t); release(tr->hit_gid);
    release(tr->row_counts); release(tr->points); release(tr->hits);
    if (tr->d_tables) (void)hipFree(tr->d_tables);
    if (tr->d_maxabs) (void)hipFree(tr->d_maxabs);
    if (tr->d_visits) (void)hipFree(tr->d_visits);
    if (tr->d_n_points) (void)hipFree(tr->d_n_points);
    if (tr->d_queue_heads) (void)hipFree(tr->d_queue_heads);
    if (tr->d_big_count) (void)hipFree(tr->d_big_count);
    release(tr->cull_list);
    if (tr->d_aabb6) (void)hipFree(tr->d_aabb6);
    release(tr->best_keys);
    release(tr->big_queue);
    release(tr->spill);
    if (tr->h_points) (void)hipHostFree(tr->h_points);
    if (tr->h_hits) (void)hipHostFree(tr->h_hits);
    if (tr->h_n_points) (void)hipHostFree(tr->h_n_points);
    if (tr->h_status) (void)hipHostFree(tr->h_status);
    for (auto &r : tr->trec)
        for (auto &e : r.ev)
            if (e) (void)hipEventDestroy(e);
    if (tr->own_stream) (void)hipStreamDestroy(tr->own_stream);
    delete tr;
}

#define LS_ENTER(tr)                                   \
    if (!(tr)) return LS_ERR_INVALID_ARGUMENT;         \
    std::lock_guard<std::mutex> lock_((tr)->mu);       \
    if (hipSetDevice((tr)->device) != hipSuccess) return fail((tr), LS_ERR_HIP, "hipSetDevice failed")

int ls_add_geometry(ls_tracer *tr, const char *name, int geometry_type, int n_vertices, int n_elements)
{
    LS_ENTER(tr);
    if (!name || n_vertices < 0 || n_elements < 0) return fail(tr, LS_ERR_INVALID_ARGUMENT, "bad argument");
    if (geometry_type != LS_GEOMETRY_TYPE_TRIANGLE && geometry_type != LS_GEOMETRY_TYPE_QUAD)
        return fail(tr, LS_ERR_UNSUPPORTED_TYPE, "only triangle and quad geometries are supported");   // EmbreeTracer.cpp:200-201
    if (geometry_type == LS_GEOMETRY_TYPE_QUAD && n_elements > 0x3FFFFFFF) return fail(tr, LS_ERR_OUT_OF_RANGE, "too many quads");
    if (tr->geoms.count(name)) return fail(tr, LS_ERR_DUPLICATE_GEOMETRY, "geometry key already exists");
    // lowest free id, like rtcAttachGeometry (EmbreeTracer.cpp:205)
    std::vector<int> used;
    for (auto &kv : tr->geoms) used.push_back(kv.second.id);
    std::sort(used.begin(), used.end());
    int id = 0;
    for (int u : used) {
        if (u == id) ++id;
        else if (u > id) break;
    }
    Geometry g;
    g.name = name;
    g.id = id;
    g.n_verts = (uint32_t)n_vertices;
    g.quad = geometry_type == LS_GEOMETRY_TYPE_QUAD;
    g.n_elems = (uint32_t)n_elements;
    g.n_tris = g.quad ? 2u * (uint32_t)n_elements : (uint32_t)n_elements;
    tr->geoms.emplace(name, g);
    tr->geometry_count += 1;
    tr->layout_dirty = true;
    return id;
}

int ls_remove_geometry(ls_tracer *tr, const char *name)
{
    LS_ENTER(tr);
    if (!name) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null name");
    auto it = tr->geoms.find(name);
    if (it == tr->geoms.end()) return -1;  // EmbreeTracer.cpp:224-225
    const int id = it->second.id;
    // frames in flight (on any of the handle's streams) may still read the mesh: order the handle's stream after
    // them, then wait, before the buffers go
    if (flush_pipeline(tr) != LS_OK) return LS_ERR_HIP;
    LS_HIP(hipStreamSynchronize(tr->stream));
    free_geometry(it->second);
    tr->geoms.erase(it);
    tr->geometry_count -= 1;
    tr->layout_dirty = true;
    // EmbreeTracer.cpp:252 commits here: a traceScene that follows traces the remaining geometry (-1 from
    // the commit of a now empty scene is not an error of the removal)
    const int rc = commit_locked(tr);
    if (rc < -1) return rc;
    return id;
}

int ls_update_geometry(ls_tracer *tr, const char *name, const float affine3x4[12], const void *verts,
                       uint32_t vert_stride, const uint32_t *tri_idx)
{
    LS_ENTER(tr);
    if (!verts) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null vertices");
    return update_common(tr, name, affine3x4, verts, vert_stride, tri_idx, hipMemcpyHostToDevice);
}

int ls_update_geometry_components(ls_tracer *tr, const char *name, const float lin[3], const float ang[3],
                                  const void *verts, uint32_t vert_stride, const uint32_t *tri_idx)
{
    LS_ENTER(tr);
    if (!lin || !ang || !verts) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null argument");
    float A[12];
    affine_from_components(lin, ang, A);
    return update_common(tr, name, A, verts, vert_stride, tri_idx, hipMemcpyHostToDevice);
}

void ls_affine_from_components(const float lin[3], const float ang[3], float affine3x4[12])
{
    if (lin && ang && affine3x4) affine_from_components(lin, ang, affine3x4);
}

int ls_update_geometry_device(ls_tracer *tr, const char *name, const float affine3x4[12], const void *d_verts,
                              uint32_t vert_stride, const uint32_t *d_tri_idx)
{
    LS_ENTER(tr);
    if (!d_verts) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null vertices");
    return update_common(tr, name, affine3x4, d_verts, vert_stride, d_tri_idx, hipMemcpyDeviceToDevice);
}

int ls_update_geometry_device_shared(ls_tracer *tr, const char *name, const float affine3x4[12], const void *d_verts,
                                     uint32_t vert_stride, const uint32_t *d_tri_idx)
{
    LS_ENTER(tr);
    if (!d_verts) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null vertices");
    auto it = tr->geoms.find(name ? name : "");
    if (it != tr->geoms.end() && !d_tri_idx && !it->second.has_idx) return fail(tr, LS_ERR_INVALID_ARGUMENT, "no indices yet");
    return update_common(tr, name, affine3x4, d_verts, vert_stride, d_tri_idx, hipMemcpyDeviceToDevice, true);
}

int ls_update_geometry_transform(ls_tracer *tr, const char *name, const float affine3x4[12])
{
    LS_ENTER(tr);
    return update_common(tr, name, affine3x4, nullptr, 0, nullptr, hipMemcpyDeviceToDevice);
}

int ls_commit_scene(ls_tracer *tr)
{
    LS_ENTER(tr);
    return commit_locked(tr);
}

int ls_trace_scene(ls_tracer *tr, uint32_t frame_index, ls_frame *out)
{
    LS_ENTER(tr);
    return trace_locked(tr, frame_index, out, true);
}

int ls_trace_scene_async(ls_tracer *tr, uint32_t frame_index, ls_frame *out)
{
    LS_ENTER(tr);
    return trace_locked(tr, frame_index, out, false);
}

long ls_geometry_count(ls_tracer *tr)
{
    if (!tr) return LS_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(tr->mu);
    return tr->geometry_count;
}

int ls_geometry_id(ls_tracer *tr, const char *name)
{
    LS_ENTER(tr);
    auto it = tr->geoms.find(name ? name : "");
    if (it == tr->geoms.end()) return fail(tr, LS_ERR_UNKNOWN_GEOMETRY, "geometry key does not exist");
    return it->second.id;
}

long ls_vertex_count(ls_tracer *tr, const char *name)
{
    LS_ENTER(tr);
    auto it = tr->geoms.find(name ? name : "");
    if (it == tr->geoms.end()) return fail(tr, LS_ERR_UNKNOWN_GEOMETRY, "geometry key does not exist");
    return (long)it->second.n_verts;
}

long ls_element_count(ls_tracer *tr, const char *name)
{
    LS_ENTER(tr);
    auto it = tr->geoms.find(name ? name : "");
    if (it == tr->geoms.end()) return fail(tr, LS_ERR_UNKNOWN_GEOMETRY, "geometry key does not exist");
    return (long)it->second.n_elems;
}

uint32_t ls_total_rays(ls_tracer *tr) { return tr ? shard_rays(tr) : 0u; }
uint32_t ls_total_channels(ls_tracer *tr) { return tr ? tr->V : 0u; }

const char *ls_last_error(ls_tracer *tr) { return tr ? tr->err.c_str() : "null tracer"; }

int ls_tracer_set_shard(ls_tracer *tr, uint32_t first_az, uint32_t n_az)
{
    LS_ENTER(tr);
    if (n_az == 0 || first_az >= tr->H || n_az > tr->H - first_az) return fail(tr, LS_ERR_OUT_OF_RANGE, "shard outside [0, H)");
    const int rc = flush_pipeline(tr);   // frames in flight keep their shard; the keys are re-armed behind them
    if (rc) return rc;
    tr->az0 = first_az;
    tr->naz = n_az;
    tr->traced = false;
    tr->keys_armed = false;
    tr->keys_b_armed = false;
    tr->keys_c_armed = false;
    return LS_OK;
}

int ls_cloud_to_world(ls_tracer *tr, const float *affine3x4, const float *R, const void *d_points32_in,
                      const uint32_t *d_n_points, void *d_points32_out, const uint32_t *d_out_base,
                      uint32_t *d_out_total, uint32_t out_capacity)
{
    LS_ENTER(tr);
    if (!R || !d_points32_in || !d_n_points || !d_points32_out)
        return fail(tr, LS_ERR_INVALID_ARGUMENT, "R, the input cloud, its count and the output cloud are required");
    static const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    ls::Affine m;
    std::memcpy(m.a, affine3x4 ? affine3x4 : kIdentity, sizeof(m.a));
    std::memcpy(m.rinv, R, sizeof(m.rinv));
    std::memcpy(m.t, tr->t, sizeof(m.t));
    // a traced cloud never holds more points than the sensor has rays; a merged input may: bound by the capacity
    const uint32_t max_points = std::max(out_capacity, tr->V * tr->H);
    ls::launch_cloud_to_world(tr->stream, m, d_points32_in, d_n_points, d_points32_out, d_out_base, d_out_total, out_capacity,
                              max_points);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

int ls_tracer_set_stream(ls_tracer *tr, void *hip_stream)
{
    LS_ENTER(tr);
    const int rc = flush_pipeline(tr);
    if (rc) return rc;
    LS_HIP(hipStreamSynchronize(tr->stream));
    tr->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : tr->own_stream;
    return LS_OK;
}

int ls_tracer_synchronize(ls_tracer *tr)
{
    LS_ENTER(tr);
    const int rc = flush_pipeline(tr);
    if (rc) return rc;
    LS_HIP(hipStreamSynchronize(tr->stream));
    return check_device_status(tr);
}

int ls_expand_points(void *dst_points32, const void *compact16, uint32_t n_points)
{
    if ((!dst_points32 || !compact16) && n_points) return LS_ERR_INVALID_ARGUMENT;
    // XYZIRBytes.cpp:24-40: x@0 y@4 z@8 0@12 intensity@16 ring@20 0@24..31; intensity is the constant 64.0 (EmbreeTracer.cpp:343)
    constexpr size_t kPer = 16384;   // points per work item: 256 KB read, 512 KB written
    const size_t n = n_points, items = (n + kPer - 1) / kPer;
    const float intensity = 64.0f;
    uint32_t ibits;
    std::memcpy(&ibits, &intensity, 4);
    const std::function<void(size_t)> work = [&](size_t i) {
        const uint32_t *src = static_cast<const uint32_t *>(compact16) + 4 * i * kPer;
        uint32_t *dst = static_cast<uint32_t *>(dst_points32) + 8 * i * kPer;
        const size_t cnt = std::min(kPer, n - i * kPer);
        for (size_t k = 0; k < cnt; ++k) {
            dst[8 * k + 0] = src[4 * k + 0];
            dst[8 * k + 1] = src[4 * k + 1];
            dst[8 * k + 2] = src[4 * k + 2];
            dst[8 * k + 3] = 0u;
            dst[8 * k + 4] = ibits;
            dst[8 * k + 5] = src[4 * k + 3];
            dst[8 * k + 6] = 0u;
            dst[8 * k + 7] = 0u;
        }
    };
    HostPool::get().run(items, work);
    return LS_OK;
}

int ls_parallel_copy(void *dst, const void *src, uint64_t bytes)
{
    if ((!dst || !src) && bytes) return LS_ERR_INVALID_ARGUMENT;
    parallel_copy(dst, src, (size_t)bytes);
    return LS_OK;
}

long ls_get_info(ls_tracer *tr, int what)
{
    LS_ENTER(tr);
    switch (what) {
    case LS_INFO_CONCURRENT_STREAMS: return tr->concurrent_streams;
    case LS_INFO_PIPELINE_MODE: return tr->opt_pipeline;
    case LS_INFO_DEVICE_STATUS: {
        const int rc = flush_pipeline(tr);
        if (rc) return rc;
        LS_HIP(hipStreamSynchronize(tr->stream));
        return (long)__atomic_exchange_n(tr->h_status, 0u, __ATOMIC_ACQ_REL);
    }
    case LS_INFO_HOST_THREADS: return HostPool::get().threads();
    case LS_INFO_AZIMUTH_COUNT: return (long)tr->H;
    case LS_INFO_LAST_COMMIT_REFIT: return (!tr->bvh_inst && tr->last_commit_refit) ? 1 : 0;
    case LS_INFO_BVH_INSTANCED: return tr->bvh_inst ? (tr->last_commit_built ? 2 : 1) : 0;
    default: return fail(tr, LS_ERR_INVALID_ARGUMENT, "unknown info key");
    }
}

int ls_tracer_flush(ls_tracer *tr)
{
    LS_ENTER(tr);
    return flush_pipeline(tr);
}

int ls_tracer_set_output_buffers(ls_tracer *tr, void *d_points32, void *d_hits, uint32_t *d_n_points, uint32_t capacity)
{
    LS_ENTER(tr);
    if (!d_points32) {
        tr->ext_points = tr->ext_hits = nullptr;
        tr->ext_n_points = nullptr;
        tr->ext_capacity = 0;
        return LS_OK;
    }
    if (!d_hits || !d_n_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "all three output buffers are required");
    tr->ext_points = d_points32;
    tr->ext_hits = d_hits;
    tr->ext_n_points = d_n_points;
    tr->ext_capacity = capacity;
    return LS_OK;
}

int ls_tracer_set_option(ls_tracer *tr, int option, int value)
{
    LS_ENTER(tr);
    switch (option) {
    case LS_OPT_LEAF_SIZE:
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(tr, LS_ERR_INVALID_ARGUMENT, "leaf size must be 1, 2, 4 or 8");
        tr->leaf_size = (uint32_t)value;
        return LS_OK;
    case LS_OPT_TIMING:
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "timing level must be 0, 1 or 2");
        tr->opt_timing = value;
        tr->trec_open = false;
        return LS_OK;
    case LS_OPT_COUNT_VISITS: tr->opt_count = value != 0; return LS_OK;
    case LS_OPT_HOST_OUTPUT:
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_HOST_OUTPUT: 0, 1 or 2");
        tr->opt_host_output = value;
        return LS_OK;
    case LS_OPT_READBACK_HITS:
        if (value < 0 || value > 1) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_READBACK_HITS: 0 or 1");
        tr->opt_readback_hits = value;
        return LS_OK;
    case LS_OPT_DEBUG_FAULT: tr->opt_debug_fault = value != 0; return LS_OK;
    case LS_OPT_BVH_REFIT: tr->opt_bvh_refit = value != 0; return LS_OK;
    case LS_OPT_BVH_INSTANCED: tr->opt_bvh_instanced = value != 0; tr->committed = false; return LS_OK;   // takes effect at the next commit
    case LS_OPT_BLOCK_CULL:
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_BLOCK_CULL: 0 off, 1 on, 2 auto");
        tr->opt_block_cull = value;
        return LS_OK;
    case LS_OPT_PIPELINE: {
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_PIPELINE: 0 off, 1 two frames on one stream, 2 three streams");
        if (value == 2) {
            // three-stream mode needs three streams whose kernels really overlap; with fewer the frames would
            // serialise silently, so the handle takes mode 1 (two frames on one stream) and says so
            int rc = flush_pipeline(tr);
            if (rc) return rc;
            LS_HIP(hipStreamSynchronize(tr->stream));
            if ((rc = ensure_slot_streams(tr))) return rc;
            if (tr->concurrent_streams < 3) {
                tr->err = "LS_OPT_PIPELINE = 2: only " + std::to_string(tr->concurrent_streams) +
                          " mutually concurrent streams found on this device; using mode 1 (two frames in flight on one stream)";
                value = 1;
            }
        }
        if (value == tr->opt_pipeline) return LS_OK;
        const int rc = flush_pipeline(tr);
        if (rc) return rc;
        LS_HIP(hipStreamSynchronize(tr->stream));
        LS_HIP(hipMemset(tr->d_big_count, 0, 4 * ls::kCounterSlotWords * sizeof(uint32_t)));   // the modes rotate the queue counters differently
        tr->pipe_seq = 0;
        tr->ms_seq = 0;
        tr->opt_pipeline = value;
        return LS_OK;
    }
    case LS_OPT_ENGINE:
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "engine must be 0 (auto), 1 (BVH) or 2 (projection)");
        if (value == 2 && !tr->projection_ok) return fail(tr, LS_ERR_INVALID_ARGUMENT, "projection engine needs channel angles within [-90, 90] degrees");
        tr->engine = value;
        tr->traced = false;
        return LS_OK;
    default: return fail(tr, LS_ERR_INVALID_ARGUMENT, "unknown option");
    }
}

int ls_get_timings(ls_tracer *tr, float ms[LS_T_COUNT])
{
    LS_ENTER(tr);
    if (!ms) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    LS_HIP(hipStreamSynchronize(tr->stream));
    static const int first[LS_T_COUNT] = {0, 1, 2, 3, 4, 5, 7, 8, 9};
    double sum[LS_T_COUNT] = {};
    uint32_t cnt[LS_T_COUNT] = {};
    for (size_t k = 0; k < tr->trec_used; ++k) {
        const ls_tracer::TimingRecord &r = tr->trec[k];
        for (int i = 0; i < LS_T_COUNT; ++i) {
            const int a = first[i], b = a + 1;
            if (!r.set[a] || !r.set[b]) continue;
            float v = 0.0f;
            if (hipEventElapsedTime(&v, r.ev[a], r.ev[b]) == hipSuccess) { sum[i] += v; ++cnt[i]; }
        }
    }
    for (int i = 0; i < LS_T_COUNT; ++i) ms[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0.0f;
    const int n = (int)tr->trec_used;
    tr->trec_used = 0;
    tr->trec_open = false;
    return n;
}

int ls_get_visit_counts(ls_tracer *tr, uint64_t counts[4])
{
    LS_ENTER(tr);
    if (!counts) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    LS_HIP(hipStreamSynchronize(tr->stream));
    LS_HIP(hipMemcpy(counts, tr->d_visits, 32, hipMemcpyDeviceToHost));
    return LS_OK;
}

int ls_expand_gathered_hits_on(ls_tracer *tr, void *hip_stream, const void *d_gathered, uint32_t world, uint32_t capacity,
                               void *d_points32, void *d_hits, uint32_t *d_n_points)
{
    LS_ENTER(tr);
    if (!d_gathered || !d_points32 || !d_hits || !d_n_points || !world || !capacity)
        return fail(tr, LS_ERR_INVALID_ARGUMENT, "null argument");
    ls::launch_expand_slots(hip_stream ? static_cast<hipStream_t>(hip_stream) : tr->stream, tables(tr),
                            static_cast<const uint32_t *>(d_gathered), world, capacity, 16u + 4u * capacity,
                            static_cast<uint8_t *>(d_points32), d_hits, d_n_points);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

int ls_expand_gathered_hits(ls_tracer *tr, const void *d_gathered, uint32_t world, uint32_t capacity, void *d_points32,
                            void *d_hits, uint32_t *d_n_points)
{
    return ls_expand_gathered_hits_on(tr, nullptr, d_gathered, world, capacity, d_points32, d_hits, d_n_points);
}

int ls_generate_rays(ls_tracer *tr, float *dx, float *dy, float *dz)
{
    LS_ENTER(tr);
    if (!dx || !dy || !dz) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    ls::launch_raygen(tr->stream, tables(tr), dx, dy, dz);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

int ls_debug_dense_hits(ls_tracer *tr, float *t, uint32_t *gid)
{
    LS_ENTER(tr);
    if (!t || !gid) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    const uint32_t n = shard_rays(tr);
    if (!tr->traced) {
        for (uint32_t q = 0; q < n; ++q) { t[q] = -1.0f; gid[q] = ls::kInvalid; }
        return LS_OK;
    }
    {
        const int rc = flush_pipeline(tr);
        if (rc) return rc;
    }
    if (tr->traced_projection) {
        // the projection engine keeps no dense arrays: rebuild them from the frame's hit records
        ls::GeomTable gt;
        gt.n = (uint32_t)tr->slot_geom_ids.size();
        gt.tri_first = tr->geom_table.p;
        gt.geom_ids = tr->geom_table.p + gt.n + 1;
        gt.prim_shift = gt.geom_ids + gt.n;
        ls::launch_dense_from_hits(tr->stream, tables(tr), tr->last_d_hits, tr->last_d_n, gt, tr->hit_t.p, tr->hit_gid.p);
        LS_HIP(hipGetLastError());
    }
    LS_HIP(hipStreamSynchronize(tr->stream));
    LS_HIP(hipMemcpy(t, tr->hit_t.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    LS_HIP(hipMemcpy(gid, tr->hit_gid.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return LS_OK;
}

int ls_debug_trace_bruteforce(ls_tracer *tr, float *t, uint32_t *gid)
{
    LS_ENTER(tr);
    if (!t || !gid) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    if (!tr->committed) return fail(tr, LS_ERR_NOT_COMMITTED, "scene not committed");
    if (!tr->scene_materialized) { const int rc = materialize_scene(tr, false); if (rc) return rc; }
    const uint32_t n = shard_rays(tr);
    float *dt = nullptr;
    uint32_t *dg = nullptr;
    LS_HIP(hipMalloc(reinterpret_cast<void **>(&dt), (size_t)n * 4));
    LS_HIP(hipMalloc(reinterpret_cast<void **>(&dg), (size_t)n * 4));
    ls::launch_bruteforce(tr->stream, tables(tr), tr->verts.p, tr->tris.p, tr->n_tris, dt, dg);
    hipError_t e = hipStreamSynchronize(tr->stream);
    if (e == hipSuccess) e = hipMemcpy(t, dt, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(gid, dg, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dt);
    (void)hipFree(dg);
    if (e != hipSuccess) return fail(tr, LS_ERR_HIP, hipGetErrorString(e));
    return LS_OK;
}

int ls_debug_scene_size(ls_tracer *tr, uint32_t *n_verts, uint32_t *n_tris, uint32_t *n_node_slots, uint32_t *leaf_size)
{
    LS_ENTER(tr);
    if (!tr->committed) return fail(tr, LS_ERR_NOT_COMMITTED, "scene not committed");
    if (n_verts) *n_verts = tr->n_verts;
    if (n_tris) *n_tris = tr->n_tris;
    if (n_node_slots) *n_node_slots = tr->n_slots;
    if (leaf_size) *leaf_size = tr->committed_leaf_size;
    return LS_OK;
}

int ls_debug_download_scene(ls_tracer *tr, float *verts_xyz, uint32_t *tri_idx)
{
    LS_ENTER(tr);
    if (!tr->committed) return fail(tr, LS_ERR_NOT_COMMITTED, "scene not committed");
    if (!tr->scene_materialized) { const int rc = materialize_scene(tr, false); if (rc) return rc; }
    LS_HIP(hipStreamSynchronize(tr->stream));
    if (verts_xyz) LS_HIP(hipMemcpy(verts_xyz, tr->verts.p, (size_t)tr->n_verts * 12, hipMemcpyDeviceToHost));
    if (tri_idx) LS_HIP(hipMemcpy(tri_idx, tr->tris.p, (size_t)tr->n_tris * 12, hipMemcpyDeviceToHost));
    return LS_OK;
}

int ls_debug_download_bvh(ls_tracer *tr, void *nodes, void *tri_records)
{
    LS_ENTER(tr);
    if (!tr->committed || !tr->bvh_built) return fail(tr, LS_ERR_NOT_COMMITTED, "no BVH: commit with LS_OPT_ENGINE = 1");
    if (tr->bvh_inst) return fail(tr, LS_ERR_NOT_COMMITTED, "the debug view shows the classic hierarchy: commit with LS_OPT_BVH_INSTANCED = 0");
    LS_HIP(hipStreamSynchronize(tr->stream));
    if (nodes) LS_HIP(hipMemcpy(nodes, tr->nodes.p, (size_t)tr->n_slots * sizeof(ls::FatNode), hipMemcpyDeviceToHost));
    if (tri_records) LS_HIP(hipMemcpy(tri_records, tr->records.p, (size_t)tr->n_tris * sizeof(ls::TriRecord), hipMemcpyDeviceToHost));
    return LS_OK;
}

}  // extern "C"
