// ls_trace.cpp -- traceScene (ITracer.hpp:94; EmbreeTracer.cpp:297-367; OptixTracer.cpp:277-358): output buffers, frames in
// flight (rider mode, three slot streams), the per-frame launch sequence of both engines, stage timings.
#include "ls_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>

namespace lsi {

ls::SensorTables tables(const ls_tracer *tr)
{
    ls::SensorTables tb;
    std::memset(static_cast<void *>(&tb), 0, sizeof(tb));   // (padding too: the frame graph compares argument bytes)
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

ls::GeomTable geom_table(const ls_tracer *tr)
{
    ls::GeomTable gt;
    std::memset(static_cast<void *>(&gt), 0, sizeof(gt));
    gt.n = (uint32_t)tr->slot_geom_ids.size();
    gt.tri_first = tr->geom_table.p;
    gt.geom_ids = tr->geom_table.p + gt.n + 1;
    gt.prim_shift = gt.geom_ids + gt.n;
    return gt;
}

// the shard's azimuth sector [lo, hi] in degrees, padded by the angular margin and 1.5 columns per side; false when the
// handle traces the full turn or a shard of 179 degrees or more (no sector test then)
bool shard_sector(const ls_tracer *tr, double &lo, double &hi)
{
    lo = hi = 0.0;
    if (!(tr->naz < tr->H) || tr->h_step == 0.0f) return false;
    const double step = tr->h_step, pad = kProjectMarginDeg + 1.5 * std::fabs(step);
    lo = (double)tr->h_begin + step * (double)tr->az0;
    hi = (double)tr->h_begin + step * (double)(tr->az0 + tr->naz - 1u);
    if (lo > hi) std::swap(lo, hi);
    lo -= pad;
    hi += pad;
    return hi - lo < 179.0;
}

ls::ProjectParams project_params(const ls_tracer *tr)
{
    ls::ProjectParams pp;
    std::memset(static_cast<void *>(&pp), 0, sizeof(pp));
    pp.tb = tables(tr);
    pp.chan_tan_up = tr->d_tables + 2 * (size_t)tr->V + 2 * (size_t)tr->H;
    pp.chan_tan_dn = pp.chan_tan_up + tr->V;
    pp.chan_perm = reinterpret_cast<const uint32_t *>(pp.chan_tan_dn + tr->V);
    pp.chan_rank = reinterpret_cast<const uint32_t *>(tr->d_tables + 5 * (size_t)tr->V + 4 * (size_t)tr->H);
    pp.chan_lut = reinterpret_cast<const uint16_t *>(tr->d_tables + 6 * (size_t)tr->V + 4 * (size_t)tr->H);
    pp.lut_t0 = tr->lut_t0;
    pp.lut_scale = tr->lut_scale;
    pp.lut_ok = tr->lut_ok ? 1 : 0;
    pp.begin_deg = tr->h_begin;
    pp.step_deg = tr->h_step;  // LidarDevice.cpp:611
    pp.inv_step_deg = pp.step_deg != 0.0f ? 1.0f / pp.step_deg : 0.0f;
    pp.inv_period = std::fabs(pp.step_deg) / 360.0f;
    pp.margin_deg = kProjectMarginDeg;
    // azimuth sector of the shard, padded by the angular margin and 1.5 columns per side, as two boundary
    // directions in counter-clockwise order; used to reject triangles early when it spans less than 180 degrees
    pp.sector_on = 0;
    pp.sec_a[0] = pp.sec_a[1] = pp.sec_b[0] = pp.sec_b[1] = 0.0f;
    double lo, hi;
    if (shard_sector(tr, lo, hi)) {
        pp.sector_on = 1;
        pp.sec_a[0] = (float)std::cos(lo * M_PI / 180.0); pp.sec_a[1] = (float)std::sin(lo * M_PI / 180.0);
        pp.sec_b[0] = (float)std::cos(hi * M_PI / 180.0); pp.sec_b[1] = (float)std::sin(hi * M_PI / 180.0);
    }
    static const uint32_t big_cells = (uint32_t)tune_int("LS_PROJECT_BIG_CELLS", 128);
    static const int debug = tune_int("LS_PROJECT_DEBUG", 0);
    pp.big_cells = big_cells;
    pp.debug = debug;
    pp.spread = 1;   // trace_locked clears it for frames that overlap on the three slot streams
    pp.xcd_remap = 0;
    // an azimuth shard's culled launch deals the survivors to its waves (ls_project.hip, DEALT)
    static const int deal = tune_int("LS_PROJECT_CULL_DEAL", -1);
    pp.cull_deal = deal >= 0 ? deal : pp.sector_on;
    static const int cols_lds = tune_int("LS_PROJECT_COLS_LDS", -1);
    // (a build option, -DLS_EXP_COLS_LDS, and off: measured on an eighth of a turn -- at SYN-1M 0.1 - 0.4 us off a frame, at SYN-10M 2 us ON the rank with the
    // most triangles in its sector: the 4 KB cost two resident workgroups per CU, and its grid is thousands of workgroups
    // that come and go.  Kept behind the experiment knob.)
    pp.cols_lds = cols_lds > 0 && tr->naz <= ls::kColsLdsMax && tr->V <= 2048u ? 1 : 0;
    return pp;
}

namespace {

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
    // Experiment (LS_EXPERIMENTAL builds, LS_CU_MASK_MODE): the three frames in flight on disjoint thirds of the CUs
    // (hipExtStreamCreateWithCUMask), 1 = every third CU, 2 = contiguous thirds -- side by side instead of interleaving
    // phases on every CU.  Candidate c takes third c % 3; the three streams chosen below must hold three different thirds.
    const int mask_mode = tune_int("LS_CU_MASK_MODE", 0);
    int third_of[kCandidates];
    {
        int n_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, tr->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        for (int c = 0; c < kCandidates; ++c) {
            third_of[c] = mask_mode ? c % 3 : -1 - c;
            if (!mask_mode) { LS_HIP(hipStreamCreateWithFlags(&cand[c], hipStreamNonBlocking)); continue; }
            std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
            for (int i = 0; i < n_cu; ++i) {
                const bool mine = mask_mode == 1 ? i % 3 == c % 3 : (i * 3) / n_cu == c % 3;
                if (mine) mask[(size_t)i / 32] |= 1u << (i % 32);
            }
            LS_HIP(hipExtStreamCreateWithCUMask(&cand[c], (uint32_t)mask.size(), mask.data()));
        }
    }
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
        for (int k = 0; k < n && ok; ++k) ok = third_of[chosen[k]] != third_of[c];
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

}  // namespace

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

namespace {

// pinned host buffers of the synchronous ls_trace_scene (written by the pack kernel itself or by D2H copies)
int ensure_host_buffers(ls_tracer *tr, size_t records)
{
    if (records <= tr->h_cap) return LS_OK;
    if (tr->h_points) LS_HIP(hipHostFree(tr->h_points));
    if (tr->h_hits) LS_HIP(hipHostFree(tr->h_hits));
    tr->h_points = nullptr;
    tr->h_hits = nullptr;
    tr->h_cap = 0;
    // explicitly coherent (fine-grained) and mapped: ls_trace_scene_begin / _expand read these while the stream is still
    // running, ordered only by the progress words that follow the pack launches -- with HIP_HOST_COHERENT=0, or a runtime
    // whose default is non-coherent host memory, a plain hipHostMalloc would not promise that
    LS_HIP(hipHostMalloc(reinterpret_cast<void **>(&tr->h_points), records * 32, hipHostMallocMapped | hipHostMallocCoherent));
    LS_HIP(hipHostMalloc(reinterpret_cast<void **>(&tr->h_hits), records * 16, hipHostMallocMapped | hipHostMallocCoherent));
    tr->h_cap = records;
    return LS_OK;
}

}  // namespace

// the sticky device status word, read after a host wait: a frame whose chained prefix gave up is lost
int check_device_status(ls_tracer *tr)
{
    const uint32_t st = __atomic_exchange_n(tr->h_status, 0u, __ATOMIC_ACQ_REL);
    if (!st) return LS_OK;
    tr->err = "device status " + std::to_string(st) + ": the chained prefix of a pipelined finish + pack pass gave up waiting; "
              "that frame's cloud is incomplete";
    return LS_ERR_HIP;
}

namespace {

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
    if (use_projection(tr)) {   // the fused finish + pack launch's status words: one region per key set
        const size_t cap1 = tr->pack_status_ms.cap;
        if ((rc = ensure(tr, tr->pack_status_ms, 3 * ((nr + 255) / 256 + 8)))) return rc;
        if (tr->pack_status_ms.cap != cap1) tr->keys_armed = false;   // (fresh memory: initialised with the keys)
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

}  // namespace

int flush_pipeline(ls_tracer *tr)
{
    if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open (ls_frame_graph_begin without ls_frame_graph_end)");
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
void mark(ls_tracer *tr, int i, hipEvent_t *ride)
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


}  // namespace lsi

namespace ls {
LaunchSink *&thread_sink()
{
    static thread_local LaunchSink *sink = nullptr;
    return sink;
}
}  // namespace ls

namespace lsi {

// ---------------------------------------------------------------------------------------------------------------
// LS_OPT_FRAME_GRAPH: a frame of the three-stream rotation as ONE graph launch (ls_launch.h).  Nearest reference shape:
// the one-trace-per-frame loop of MeshProjector.cpp:446-464; the reference has no counterpart (its OptiX path pays a
// cudaMalloc, two uploads and a device-wide wait per frame, OptixTracer.cpp:277-358).
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kFrameRetry = 1000;   // trace_once -> trace_locked: the frame graph was discarded, issue the frame again

void frame_graph_free(FrameGraph &fg)
{
    if (fg.exec) (void)hipGraphExecDestroy(fg.exec);
    if (fg.graph) (void)hipGraphDestroy(fg.graph);
    fg.exec = nullptr;
    fg.graph = nullptr;
    fg.recs.clear();
    fg.nodes.clear();
    fg.sig = 0;
}

// what decides the SEQUENCE of launches of a frame (their arguments may change from frame to frame: those are patched)
uint64_t frame_signature(const ls_tracer *tr, const std::vector<ls::GeomSource> &srcs, uint32_t slot)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(tr->fg_bracket ? tr->fg_tag | 1ull : 0ull);
    mix(slot);
    mix(tr->V);
    mix(shard_rays(tr) ? 1u : 0u);
    mix(srcs.size());
    for (const ls::GeomSource &g : srcs) mix(((uint64_t)(g.ntris ? 1u : 0u) << 1) | (g.boxes ? 1u : 0u));
    return h ? h : 1ull;
}

// begin building the frame of `slot` on its stream: describe against the cached graph, or capture a new one
int frame_graph_open(ls_tracer *tr, uint32_t slot, hipStream_t s, uint64_t sig)
{
    FrameGraph &fg = tr->fgraph[slot];
    if (fg.exec && fg.sig != sig) {
        // another launch sequence: the cached graph goes -- once its last launch on this stream has finished (rare: a changed
        // geometry set, culling switched, a caller's bracket that comes or goes)
        LS_HIP(hipStreamSynchronize(s));
        frame_graph_free(fg);
    }
    ls::LaunchSink &sk = tr->fg_sink;
    sk.n = 0;
    sk.stream = s;
    tr->fg_slot = slot;
    if (fg.exec) {
        sk.mode = ls::LaunchSink::kDescribe;
    } else {
        LS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        sk.mode = ls::LaunchSink::kCapture;
        fg.sig = sig;
    }
    tr->fg_open = true;
    ls::thread_sink() = &sk;
    return LS_OK;
}

// the kernel nodes of `graph` in an order that respects its edges (our launches sit on one stream: each depends on the
// one before, so any topological order lists them as they were issued, whatever a captured library call added around them)
int kernel_nodes_in_order(ls_tracer *tr, hipGraph_t graph, std::vector<hipGraphNode_t> &out)
{
    size_t n = 0, ne = 0;
    LS_HIP(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) LS_HIP(hipGraphGetNodes(graph, nodes.data(), &n));
    LS_HIP(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    if (ne) LS_HIP(hipGraphGetEdges(graph, from.data(), to.data(), &ne));
    std::vector<uint32_t> indeg(n, 0);
    auto index_of = [&](hipGraphNode_t x) { for (size_t i = 0; i < n; ++i) if (nodes[i] == x) return i; return n; };
    std::vector<size_t> fi(ne), ti(ne);
    for (size_t e = 0; e < ne; ++e) {
        fi[e] = index_of(from[e]);
        ti[e] = index_of(to[e]);
        if (fi[e] == n || ti[e] == n) return fail(tr, LS_ERR_HIP, "frame graph: an edge names an unknown node");
        ++indeg[ti[e]];
    }
    std::vector<size_t> ready, order;
    for (size_t i = 0; i < n; ++i) if (!indeg[i]) ready.push_back(i);
    while (!ready.empty()) {
        const size_t i = ready.back();
        ready.pop_back();
        order.push_back(i);
        for (size_t e = 0; e < ne; ++e) if (fi[e] == i && --indeg[ti[e]] == 0) ready.push_back(ti[e]);
    }
    if (order.size() != n) return fail(tr, LS_ERR_HIP, "frame graph: the captured graph has a cycle");
    out.clear();
    for (size_t i : order) {
        hipGraphNodeType ty;
        LS_HIP(hipGraphNodeGetType(nodes[i], &ty));
        if (ty == hipGraphNodeTypeKernel) out.push_back(nodes[i]);
    }
    return LS_OK;
}

// the frame graph is given up for this frame (and, when `broken`, for good): nothing was launched, the rotation steps back
int frame_graph_discard(ls_tracer *tr, FrameGraph &fg, bool broken)
{
    if (fg.exec) (void)hipStreamSynchronize(tr->fg_sink.stream);   // (an earlier launch of it may still be running)
    frame_graph_free(fg);
    if (broken) tr->fg_broken = true;
    if (tr->ms_seq) --tr->ms_seq;   // (trace_once counted the frame when its launches were recorded)
    return kFrameRetry;
}

// close the frame being built and launch it: LS_OK, kFrameRetry (nothing launched, issue it again) or an error
int frame_graph_close(ls_tracer *tr)
{
    FrameGraph &fg = tr->fgraph[tr->fg_slot];
    ls::LaunchSink &sk = tr->fg_sink;
    hipStream_t s = sk.stream;
    const int mode = sk.mode;
    sk.mode = ls::LaunchSink::kOff;
    tr->fg_open = false;
    ls::thread_sink() = nullptr;
    if (mode == ls::LaunchSink::kCapture) {
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (e != hipSuccess || !graph) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            tr->err = std::string("frame graph: the capture failed (") + hipGetErrorString(e) + "); plain launches from now on";
            return frame_graph_discard(tr, fg, true);
        }
        fg.graph = graph;
        if (hipGraphInstantiate(&fg.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            fg.exec = nullptr;
            tr->err = "frame graph: hipGraphInstantiate failed; plain launches from now on";
            return frame_graph_discard(tr, fg, true);
        }
        std::vector<hipGraphNode_t> kn;
        if (kernel_nodes_in_order(tr, graph, kn) != LS_OK) return frame_graph_discard(tr, fg, true);
        fg.nodes.clear();
        size_t at = 0;
        for (size_t i = 0; i < sk.n; ++i) {
            hipGraphNode_t found = nullptr;
            for (; at < kn.size() && !found; ++at) {
                hipKernelNodeParams p;
                if (hipGraphKernelNodeGetParams(kn[at], &p) == hipSuccess && p.func == sk.recs[i].func) found = kn[at];
            }
            if (!found) {
                tr->err = "frame graph: a launch has no kernel node in the captured graph; plain launches from now on";
                return frame_graph_discard(tr, fg, true);
            }
            fg.nodes.push_back(found);
        }
        fg.recs.assign(sk.recs.begin(), sk.recs.begin() + (ptrdiff_t)sk.n);
        ++tr->fg_captures;
    } else {
        bool same_sequence = sk.n == fg.recs.size();
        for (size_t i = 0; same_sequence && i < sk.n; ++i) same_sequence = sk.recs[i].func == fg.recs[i].func;
        if (!same_sequence) return frame_graph_discard(tr, fg, false);   // (the signature missed something: capture anew)
        tr->fg_last_patched = 0;
        bool idle = false;
        for (size_t i = 0; i < sk.n; ++i) {
            ls::LaunchRecord &now = sk.recs[i];
            if (ls::same_launch(now, fg.recs[i])) continue;
            // A node's arguments are about to change while this exec's previous launch -- three frames back on this very
            // stream -- may still be queued or running: nothing bounds how far the host runs ahead of the device, and whether
            // a launched graph keeps a snapshot of its kernel arguments or reads the exec's (ROCm keeps a graph's kernel
            // arguments in a device-side pool that SetParams rewrites in place) is the runtime's business.  So the stream is
            // idle before the first patch: one query when it already is -- the steady state of a GPU-bound stream of frames,
            // the host at most three frames ahead -- a wait otherwise.  Frames whose arguments did not change pay nothing.
            if (!idle) {
                if (hipStreamQuery(s) != hipSuccess) {
                    (void)hipGetLastError();
                    const hipError_t e = hipStreamSynchronize(s);
                    if (e != hipSuccess) {   // (nothing of this frame was launched: the rotation steps back like on any other discard)
                        (void)frame_graph_discard(tr, fg, true);
                        tr->err = std::string("frame graph: waiting for the graph's previous launch: ") + hipGetErrorString(e);
                        return LS_ERR_HIP;
                    }
                    ++tr->fg_patch_waits;
                }
                idle = true;
            }
            tr->fg_last_patched |= 1u << (i < 31 ? i : 31);
            void *argv[ls::kMaxLaunchArgs];
            ls::argument_pointers(now, argv);
            hipKernelNodeParams p;
            std::memset(&p, 0, sizeof(p));
            p.func = const_cast<void *>(now.func);
            p.gridDim = now.grid;
            p.blockDim = now.block;
            p.sharedMemBytes = now.shmem;
            p.kernelParams = argv;
            p.extra = nullptr;
            if (hipGraphExecKernelNodeSetParams(fg.exec, fg.nodes[i], &p) != hipSuccess) {
                (void)hipGetLastError();
                return frame_graph_discard(tr, fg, false);
            }
            std::swap(fg.recs[i], now);   // (the sink's record is rewritten by the next frame anyway)
            ++tr->fg_patches;
        }
    }
    LS_HIP(hipGraphLaunch(fg.exec, s));
    ++tr->fg_replays;
    return LS_OK;
}

}  // namespace

void frame_graph_destroy(ls_tracer *tr)
{
    for (FrameGraph &fg : tr->fgraph) frame_graph_free(fg);
}

namespace {
int trace_once(ls_tracer *tr, uint32_t frame, ls_frame *out, bool readback)
{
    if (!out) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null frame");
    std::memset(out, 0, sizeof(*out));
    out->frame = frame;
    out->n_rays = shard_rays(tr);
    tr->traced = false;
    tr->last_slot = 0xFFFFFFFFu;
    tr->last_stream = tr->stream;
    if (!tr->committed || tr->n_tris == 0) return -1;  // OptixTracer.cpp:280-288: cleared cloud, -1
    int rc;
    if ((rc = ensure_outputs(tr))) return rc;
    if (tr->ext_points && tr->ext_capacity < shard_rays(tr))
        return fail(tr, LS_ERR_OUT_OF_RANGE, "external output buffers smaller than the shard's ray count");
    if (tr->ext_hits_only && tr->opt_emit_points)
        return fail(tr, LS_ERR_INVALID_ARGUMENT, "hit buffers alone are installed (ls_tracer_set_hit_buffers) and LS_OPT_EMIT_POINTS is 1");

    hipStream_t s = tr->stream;   // three-stream mode switches to the frame's own stream below
    bool progress = false;
    const ls::SensorTables tb = tables(tr);
    uint8_t *d_points = tr->ext_points ? static_cast<uint8_t *>(tr->ext_points) : tr->points.p;
    void *d_hits = tr->ext_points ? tr->ext_hits : static_cast<void *>(tr->hits.p);
    uint32_t *d_n = tr->ext_points ? tr->ext_n_points : tr->d_n_points;
    // synchronous call with the library's own outputs: the pack kernel writes the pinned host buffers itself
    const bool hv = readback && tr->opt_host_output && !tr->ext_points;
    if (hv && (rc = ensure_host_buffers(tr, shard_rays(tr)))) return rc;
    // (3: no point records at all -- LS_OPT_EMIT_POINTS = 0, asynchronous frames only)
    if (readback && !tr->opt_emit_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_EMIT_POINTS = 0: the synchronous call delivers points; switch it on first");
    const uint32_t compact = hv && tr->opt_host_output == 2 ? 1u : (!tr->opt_emit_points ? 3u : 0u);
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
        // ls_trace_scene_begin: the finish and pack passes report their progress to the host (one frame in flight only)
        ls::ProgressArgs pg{nullptr, 0u, 0u, nullptr};
        if (tr->progress_req && hv && compact && !pipelined && !multi && !tr->opt_count && !tr->opt_timing) {
            if (++tr->progress_epoch == 0u) tr->progress_epoch = 1u;
            // the pack pass's two launches split where the POINTS halved last frame (the upper rings mostly see sky);
            // a frame without a hint, or with another raster, splits the ray blocks in the middle
            if (tr->pack_split == 0u || tr->pack_split >= n_blocks || tr->pack_split_blocks != n_blocks) tr->pack_split = n_blocks / 2u;
            tr->pack_split_blocks = n_blocks;
            pg = {tr->h_progress, tr->progress_epoch, tr->pack_split, nullptr};
            progress = true;
        }
        if (!pipelined && !multi && (rc = flush_pipeline(tr))) return rc;
        {   // spread runs balance the cells per wave (shorter kernel: 23.6 -> 21.2 us alone) but touch more cache lines,
            // which costs more than it gains once three frames overlap (16.9 -> 17.2 us per frame): LS_PROJECT_SPREAD overrides
            static const int spread_env = tune_int("LS_PROJECT_SPREAD", -1);
            pp.spread = spread_env >= 0 ? spread_env : (multi ? 0 : 1);
            pp.xcd_remap = multi ? 0u : 1u;   // (the same trade: good for a frame alone, bad for three at once; ls_project.hip)
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
            if (tr->pack_status_ms.p) {
                // no status word carries a tag, every slot's tag word starts at 1 (the regions follow the shard's block count)
                LS_HIP(hipMemsetAsync(tr->pack_status_ms.p, 0, tr->pack_status_ms.cap * 8, s));
                for (uint32_t k = 0; k < 3u; ++k)
                    LS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tr->pack_status_ms.p + (size_t)k * (n_blocks + 8u) + n_blocks), 1, 1, s));
            }
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
            tr->last_slot = slot;
            tr->last_stream = s;
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
            std::memset(static_cast<void *>(&src), 0, sizeof(src));   // (padding too: the frame graph compares argument bytes)
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
            const bool culled = cull_enabled(tr, ge) && ge.d_boxes && ge.d_corners && !ge.order_stale && !ge.bounds_stale &&
                                ls::project_tris_per_wave(ge.n_tris) == 64u;
            src.perm = culled ? ge.d_perm : nullptr;
            src.boxes = culled ? ge.d_boxes : nullptr;
            src.corners = culled ? ge.d_corners : nullptr;
            if (culled) src.idx = ge.d_idx_sorted;
            any_culled = any_culled || culled;
            srcs.push_back(src);
        }
        // survivor list of this frame's k_cull: one of three (as many frames as can be in flight)
        uint32_t *cull_list = nullptr;
        if (any_culled) {
            const uint32_t entries = ls::project_cull_entries(srcs.data(), (uint32_t)srcs.size(), pp.sector_on != 0);
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
        const ls::GeomTable gt = geom_table(tr);
        // how full the survivor segments of a recent culled frame were (k_pack / k_finish_pack leave 1 + the fullest one in
        // pinned host memory, no wait): sizes this frame's k_project<CULLED> grid.  A scene that changes under it costs a frame
        // or three of waves walking their segment in rounds, never a wrong cloud.  The geometry set it speaks for is the
        // handle's: a new layout forgets it.
        uint32_t *hint_word = tr->h_status + 1;
        // With hysteresis: under a pose that changes every frame the raw number wanders, every change of it that
        // crosses a quantum of the grid is a changed node parameter of a captured frame graph, and a patch may have to wait for
        // the graph's previous launch.  The number in use grows at once (with an eighth on top, so that a scene that keeps growing
        // does not patch every frame) and shrinks only when the fresh one has fallen below half of it.
        uint32_t survivors_hint = 0u;
        if (any_culled) {
            const uint32_t fresh = __atomic_load_n(hint_word, __ATOMIC_RELAXED);
            if (fresh && (fresh > tr->cull_hint_in_use || fresh < tr->cull_hint_in_use / 2u)) tr->cull_hint_in_use = fresh + fresh / 8u;
            if (!fresh) tr->cull_hint_in_use = 0u;   // (a new layout forgot it)
            survivors_hint = tr->cull_hint_in_use;
        }
        if (pipelined) {
            // one launch: this frame's k_project workgroups + the previous frame's finish + pack workgroups
            ls::launch_project(s, pp, srcs.data(), (uint32_t)srcs.size(), keys, bigq, tr->big_capacity, big_count, nullptr,
                               tr->pipe_pending ? &tr->pipe_fa : nullptr, cull_list, nullptr, nullptr, survivors_hint);
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
            fa.cull_hint = nullptr;   // (rider mode re-arms the counters of the frame after the next: nothing to read there)
            tr->pipe_pending = true;
            ++tr->pipe_seq;
            if (readback && (rc = flush_pipeline(tr))) return rc;
        } else {
            // LS_OPT_FRAME_GRAPH (three-stream mode): the launches below are captured into this slot's graph, or only
            // described and compared with it; either way the frame then goes out as one graph launch
            if (multi && tr->opt_frame_graph && !tr->fg_broken && (rc = frame_graph_open(tr, slot, s, frame_signature(tr, srcs, slot)))) return rc;
            // one launch per 16 geometries (the descriptors travel as kernel arguments)
            if (ride) mark(tr, 8, &ev_k1);
            { static const int no_hits = tune_int("LS_PACK_NO_HITS", 0); if (no_hits && !readback) d_hits = nullptr; }   // (experiment: what the 16-byte hit records cost)
            ls::launch_project(s, pp, srcs.data(), (uint32_t)srcs.size(), keys, bigq, tr->big_capacity, big_count, stats, nullptr, cull_list,
                               ev_k0, ev_k1, survivors_hint);
            if (!ride) mark(tr, 8);
            // Three-stream mode, a shard of at most kFuseBlocks ray blocks: finish + pack as ONE launch (k_finish_pack: the
            // chained prefix of rider mode; a few hundred workgroups, all resident at once, publish within a microsecond of
            // each other).  Each of the two launches it replaces sat at the ~5 us floor of a dependent launch for half a
            // megabyte of keys; at the full raster's 2 048 blocks the look-back costs more than the second read of the keys
            // (16.3 - 16.6 us per frame against 15.4) and the two launches stay.
            static const uint32_t fuse_blocks = (uint32_t)tune_int("LS_FUSE_FINISH_PACK_BLOCKS", 512);
            // Plain launches only: there the frame is bound by the host's enqueues and a launch less is what pays (an eighth-
            // of-a-turn shard at SYN-1M: 10.9 - 15.9 -> 8.2 - 11.9 us per frame); inside a captured frame graph a node costs
            // the host nothing and the look-back's polls cost more than the second read of the keys (8.3 -> 9.1 us, 10.8 -> 12.0
            // at SYN-10M).
            static const int fuse_in_graph = tune_int("LS_FUSE_IN_GRAPH", 0);   // (experiment)
            const bool graphed = multi && tr->opt_frame_graph && !tr->fg_broken && !fuse_in_graph;
            static const int fuse_single = tune_int("LS_FUSE_FINISH_PACK_SINGLE", 0);   // (experiment: also with one frame in flight)
            if ((multi || fuse_single) && !graphed && !progress && !stats && n_blocks <= fuse_blocks && tr->pack_status_ms.p) {
                ls::FinishPackArgs fa;
                std::memset(static_cast<void *>(&fa), 0, sizeof(fa));   // (padding too: the frame graph compares argument bytes)
                fa.best = keys;
                fa.big = bigq;
                fa.big_capacity = tr->big_capacity;
                fa.big_count = big_count;
                fa.rearm_big_count = big_count;   // the slot's own: re-armed by the last workgroup, behind every reader
                fa.status = tr->pack_status_ms.p + (size_t)slot * (n_blocks + 8u);
                fa.epoch_word = reinterpret_cast<uint32_t *>(tr->pack_status_ms.p + (size_t)slot * (n_blocks + 8u) + n_blocks);
                fa.spin_limit = 1u << 18;
                fa.device_status = tr->h_status;
                fa.gt = gt;
                fa.points32 = d_points;
                fa.hits = d_hits;
                fa.n_points = d_n;
                fa.n_blocks = n_blocks;
                fa.compact = compact;
                fa.cull_hint = any_culled ? hint_word : nullptr;
                ls::launch_finish_pack(s, pp, fa, nullptr);
                mark(tr, 9);
                mark(tr, 10);
            } else {
                // three-stream mode: the two passes as fewer, fatter waves (the device is short of wave slots there, not alone:
                // 15.2 -> 13.4 us per frame with three in flight, 24.3 -> 25.4+ with one -- ls_project.hip, k_project_finish_wide)
                static const int rpl_env = tune_int("LS_PROJECT_RAYS_PER_LANE", 0);   // (experiment: 1 switches it off)
                const uint32_t rpl = multi && !progress && !stats ? (rpl_env ? (uint32_t)rpl_env : ls::project_rays_per_lane(n_blocks)) : 1u;
                ls::launch_project_finish(s, pp, keys, bigq, tr->big_capacity, big_count, counts, stats, rpl);
                mark(tr, 9);
                // (a frame that reports its progress sends 8-byte (ray, t) records: ls_trace_scene_expand rebuilds the points)
                pg.cull_hint = any_culled ? hint_word : nullptr;
                ls::launch_pack_keys(s, tb, keys, tr->hit_t.p, tr->hit_gid.p, counts, next_counts, big_count, gt, d_points, d_hits, d_n,
                                     progress ? 2u : compact, &pg, rpl);
                mark(tr, 10);
            }
            if (multi) {
                // no event per frame: a flush (or a mesh copy) records one per stream and orders the handle's stream after it
                tr->slot_pending[slot] = true;
                ++tr->ms_seq;
                // (a caller's bracket keeps the graph open for its own work: ls_frame_graph_end closes it)
                if (tr->fg_open && (!tr->fg_bracket || readback) && (rc = frame_graph_close(tr))) return rc;
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
        // the ray queues' heads are zero: the last BVH frame's k_rowcount left them so (a frame that never got that far, or
        // the first one, zeroes them here)
        if (!tr->queue_heads_armed) LS_HIP(hipMemsetAsync(tr->d_queue_heads, 0, ls::kQueues * 16 * sizeof(uint32_t), s));
        tr->queue_heads_armed = false;
        ls::RayQueues rq;
        rq.heads = tr->d_queue_heads;
        rq.chan_mul = tr->chan_mul;
        rq.refill_min = tr->refill_min;
        // a wave runs its leaf tests when 16 of its lanes stand at a leaf (or none has a node to go to): the ~200 instructions
        // of a leaf test are then issued every few trips for many lanes instead of on every trip for a handful
        // (4 / 8 / 16 / 24 / 32 / 48 lanes: 143.2 / 140.7 / 139.1 / 141.3 / 149.6 / 154.8 us per frame against 153.4 with none)
        { static const int lw = tune_int("LS_TRACE_LEAF_WAIT", 16); rq.leaf_wait = (uint32_t)std::max(0, std::min(64, lw)); }
        rq.chan_order = reinterpret_cast<const uint32_t *>(tr->d_tables + 4 * (size_t)tr->V + 2 * (size_t)tr->H);   // chan_perm (fill_tables)
        { static const bool no_order = tune_int("LS_TRACE_NO_ORDER", 0) != 0; if (no_order) rq.chan_order = nullptr; }
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
            // the four-wide twins of hierarchies that have stood for kWidenAfterFrames frames are made now (ls_commit.cpp: small ones
            // were made at their commit); the wide walk needs every geometry's
            bool all_wide = tr->wide_valid;
            if (tr->wide_valid) {
                constexpr uint32_t kWidenAfterFrames = 2u;
                for (uint32_t i = 0; i < batch.n; ++i) {
                    ls_tracer::InstSlot &sl = tr->inst_layout[i];
                    if (sl.wide_made || sl.n_leaves < 2u) continue;
                    if (++sl.wide_age >= kWidenAfterFrames) {
                        ls::launch_widen(s, tr->nodes.p + sl.node_first, sl.n_leaves, tr->wide_nodes.p + sl.node_first);
                        sl.wide_made = true;
                    } else {
                        all_wide = false;
                    }
                }
            }
            tr->wide_in_use = all_wide;
            ls::launch_trace_instanced(s, tr->trace_blocks, tb, rq, batch, tr->nodes.p, all_wide ? tr->wide_nodes.p : nullptr, tr->records.p, tr->inst_leaf_size,
                                       (batch.n == 1u && tr->treelet_valid) ? tr->treelet.p : nullptr, tr->hit_t.p, tr->hit_gid.p, tr->spill.p, tr->opt_count ? tr->d_visits : nullptr);
        } else {
            ls::launch_trace(s, tr->trace_blocks, tb, rq, tr->nodes.p, tr->records.p, tr->n_leaves, tr->committed_leaf_size,
                             tr->n_tris, tr->hit_t.p, tr->hit_gid.p, tr->spill.p, tr->opt_count ? tr->d_visits : nullptr);
        }
        mark(tr, 8);
        ls::launch_rowcount(s, tr->hit_gid.p, shard_rays(tr), tr->row_counts.p, tr->d_queue_heads);
        tr->queue_heads_armed = shard_rays(tr) != 0u;
        tr->keys_armed = false;  // the counter array was just used with the BVH layout
        mark(tr, 9);
        const ls::GeomTable gt = geom_table(tr);
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

    if (hv && progress) {
        // ls_trace_scene_begin polls the device's progress words instead of waiting for the stream
        tr->progress_active = true;
        tr->begin_blocks = (shard_rays(tr) + 255u) / 256u;
        out->compact16 = nullptr;   // ((ray, t) records, for ls_trace_scene_expand only)
        return LS_OK;
    }
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

int trace_locked(ls_tracer *tr, uint32_t frame, ls_frame *out, bool readback)
{
    int rc = trace_once(tr, frame, out, readback);
    // a frame graph that had to be given up launched nothing: the frame is issued again (captured anew, or as plain launches)
    if (rc == kFrameRetry) rc = trace_once(tr, frame, out, readback);
    if (rc == kFrameRetry) rc = fail(tr, LS_ERR_HIP, "frame graph: discarded twice in a row");
    if (rc < LS_OK && rc != -1 && tr->fg_open) {   // an error with a capture open: end it, launch nothing
        ls::LaunchSink &sk = tr->fg_sink;
        if (sk.mode == ls::LaunchSink::kCapture) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(sk.stream, &g);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
        }
        sk.mode = ls::LaunchSink::kOff;
        tr->fg_open = false;
        ls::thread_sink() = nullptr;
    }
    return rc;
}

}  // namespace lsi

using namespace lsi;

extern "C" {

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

namespace {
// spin on a progress word of the frame in flight (the device releases it with system scope); a frame takes tens of
// microseconds, so this is a busy wait -- with a ceiling of seconds behind which the stream is waited for instead
bool wait_epoch(const uint32_t *word, uint32_t epoch)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == epoch) return true;
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return false;
    }
}
}  // namespace

int ls_trace_scene_begin(ls_tracer *tr, uint32_t frame_index, uint32_t *n_points)
{
    LS_ENTER(tr);
    if (!n_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null output");
    *n_points = 0;
    tr->begin_open = false;
    tr->progress_active = false;
    if (tr->ext_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "ls_trace_scene_begin delivers to host memory: reset ls_tracer_set_output_buffers first");
    // compact points in pinned host memory, whatever LS_OPT_HOST_OUTPUT says for ls_trace_scene; no hit records
    const int host_output = tr->opt_host_output, readback_hits = tr->opt_readback_hits;
    tr->opt_host_output = 2;
    tr->opt_readback_hits = 0;
    tr->progress_req = true;
    ls_frame f;
    const int rc = trace_locked(tr, frame_index, &f, true);
    tr->progress_req = false;
    tr->opt_host_output = host_output;
    tr->opt_readback_hits = readback_hits;
    if (rc < 0) { tr->progress_active = false; return rc; }   // (-1: empty scene, zero points)
    if (tr->progress_active) {
        if (!wait_epoch(&tr->h_progress->total_epoch, tr->progress_epoch)) {
            // five seconds without the word: a slow but healthy frame (first-launch code load, a debugger, a loaded box) or a
            // lost one.  The stream is waited for; after that every word the frame writes is final: only a frame whose
            // words are STILL missing is an error
            LS_HIP(hipStreamSynchronize(tr->stream));
            int st;
            if ((st = check_device_status(tr))) { tr->progress_active = false; return st; }
            if (__atomic_load_n(&tr->h_progress->total_epoch, __ATOMIC_ACQUIRE) != tr->progress_epoch) {
                tr->progress_active = false;
                return fail(tr, LS_ERR_HIP, "the frame's hit count never reached the host");
            }
        }
        tr->begin_points = __atomic_load_n(&tr->h_progress->total, __ATOMIC_RELAXED);
        tr->begin_first = tr->pack_split ? std::min(__atomic_load_n(&tr->h_progress->n_first, __ATOMIC_RELAXED), tr->begin_points) : 0u;
        tr->pack_split = __atomic_load_n(&tr->h_progress->even_split, __ATOMIC_RELAXED);   // the next frame's split
    } else {
        tr->begin_points = f.n_points;   // (BVH engine, frames in flight, external buffers ...: the frame is complete already)
    }
    tr->begin_open = true;
    *n_points = tr->begin_points;
    return LS_OK;
}

int ls_trace_scene_expand(ls_tracer *tr, void *dst_points32)
{
    LS_ENTER(tr);
    if (!tr->begin_open) return fail(tr, LS_ERR_NOT_COMMITTED, "no frame begun (ls_trace_scene_begin)");
    tr->begin_open = false;
    const uint32_t n = tr->begin_points;
    if (n && !dst_points32) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null destination");
    uint8_t *dst = static_cast<uint8_t *>(dst_points32);
    if (!tr->progress_active) return n ? ls_expand_points(dst, tr->h_points, n) : LS_OK;
    tr->progress_active = false;
    const ls::HostProgress *hp = tr->h_progress;
    const uint32_t epoch = tr->progress_epoch;
    // One job for the worker threads, started NOW -- their wake-up runs under the pack pass --: the items of the first
    // half of the ray blocks wait for "the first half has arrived", the others for "all have" (n_first came with the hit
    // count); the first half is expanded while the second half is still crossing PCIe
    const uint32_t n_first = tr->begin_first;
    const size_t items_first = (n_first + kExpandItem - 1) / kExpandItem, items_rest = ((size_t)(n - n_first) + kExpandItem - 1) / kExpandItem;
    std::atomic<bool> ok{true};
    const uint8_t *src = tr->h_points;
    const uint32_t V = tr->V, H = tr->H;
    const float *sin_theta = tr->host_tables.data(), *cos_theta = sin_theta + V, *cs_phi = sin_theta + 2 * (size_t)V + 2 * (size_t)H + 3 * (size_t)V;
    const std::function<void(size_t)> work = [&](size_t i) {
        const bool first = i < items_first;
        if (!wait_epoch(first ? &hp->half_epoch : &hp->all_epoch, epoch)) { ok.store(false); return; }
        const size_t at = first ? i * kExpandItem : (size_t)n_first + (i - items_first) * kExpandItem;
        const size_t end = first ? (size_t)n_first : (size_t)n;
        expand_hits_range(dst + 32 * at, src + 8 * at, std::min(kExpandItem, end - at), sin_theta, cos_theta, cs_phi, V, H);
    };
    pool_run(items_first + items_rest, work);
    if (!n && !wait_epoch(&hp->all_epoch, epoch)) ok.store(false);   // (nothing to expand: still the frame's end)
    if (!ok.load()) {
        // (as in ls_trace_scene_begin: wait for the stream, then the words are final -- expand whatever a worker gave up on)
        LS_HIP(hipStreamSynchronize(tr->stream));
        int st;
        if ((st = check_device_status(tr))) return st;
        if (__atomic_load_n(&hp->all_epoch, __ATOMIC_ACQUIRE) != epoch) return fail(tr, LS_ERR_HIP, "the frame's progress words never reached the host");
        if (n) expand_hits_range(dst, src, n, sin_theta, cos_theta, cs_phi, V, H);
    }
    return check_device_status(tr);
}

int ls_frame_graph_begin(ls_tracer *tr, uint64_t tag)
{
    LS_ENTER(tr);
    if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open already");
    tr->fg_bracket = true;
    tr->fg_tag = tag << 1;
    return LS_OK;
}

int ls_frame_graph_stream(ls_tracer *tr, void **hip_stream, uint32_t *slot, int *mode)
{
    LS_ENTER(tr);
    // the frame issued last; when nothing was traced (an empty scene) the stream and slot the NEXT frame of the rotation takes
    hipStream_t s = tr->last_stream ? tr->last_stream : tr->stream;
    uint32_t sl = tr->last_slot;
    if (!tr->traced && tr->opt_pipeline == 2 && use_projection(tr) && tr->slot_stream[0] && !tr->opt_count && !tr->opt_timing) {
        sl = tr->ms_seq % 3u;
        s = tr->slot_stream[sl];
    }
    if (hip_stream) *hip_stream = s;
    if (slot) *slot = sl;
    if (mode) *mode = !tr->fg_open ? LS_FRAME_EAGER : (tr->fg_sink.mode == ls::LaunchSink::kCapture ? LS_FRAME_CAPTURING : LS_FRAME_REPLAYING);
    return LS_OK;
}

int ls_frame_graph_end(ls_tracer *tr)
{
    LS_ENTER(tr);
    tr->fg_bracket = false;
    if (!tr->fg_open) return LS_OK;   // the frame went out as plain launches (or nothing was traced)
    const int rc = frame_graph_close(tr);
    return rc == kFrameRetry ? 1 : rc;
}

int ls_frame_graph_reset(ls_tracer *tr)
{
    LS_ENTER(tr);
    if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open");
    // (a graph that is still executing keeps what it needs until it has finished: the runtime defers the release)
    const int rc = flush_pipeline(tr);
    if (rc) return rc;
    LS_HIP(hipStreamSynchronize(tr->stream));
    frame_graph_destroy(tr);
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

int ls_tracer_flush(ls_tracer *tr)
{
    LS_ENTER(tr);
    return flush_pipeline(tr);
}

int ls_tracer_wait_event(ls_tracer *tr, void *hip_event)
{
    LS_ENTER(tr);
    if (!hip_event) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null event");
    LS_HIP(hipStreamWaitEvent(tr->stream, static_cast<hipEvent_t>(hip_event), 0));
    ++tr->main_epoch;   // every slot stream orders its next frame after the handle's stream
    return LS_OK;
}

int ls_tracer_next_frame_waits(ls_tracer *tr, void *hip_event)
{
    LS_ENTER(tr);
    if (!hip_event) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null event");
    // three-stream mode with its streams in place: the next frame runs on slot ms_seq % 3 -- one wait there
    if (tr->opt_pipeline == 2 && use_projection(tr) && tr->slot_stream[0] && !tr->opt_count && !tr->opt_timing) {
        LS_HIP(hipStreamWaitEvent(tr->slot_stream[tr->ms_seq % 3u], static_cast<hipEvent_t>(hip_event), 0));
        return LS_OK;
    }
    LS_HIP(hipStreamWaitEvent(tr->stream, static_cast<hipEvent_t>(hip_event), 0));
    ++tr->main_epoch;
    return LS_OK;
}

int ls_tracer_order_after_last_frame(ls_tracer *tr, void *hip_stream)
{
    LS_ENTER(tr);
    if (!hip_stream) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null stream");
    hipStream_t waiter = static_cast<hipStream_t>(hip_stream);
    hipStream_t src = tr->stream;
    hipEvent_t ev = nullptr;
    if (tr->opt_pipeline == 2 && tr->ms_seq && tr->slot_pending[(tr->ms_seq - 1u) % 3u]) {
        // three-stream mode: the frame issued last runs on its slot's stream; the other frames in flight are not waited for
        const uint32_t slot = (tr->ms_seq - 1u) % 3u;
        src = tr->slot_stream[slot];
        ev = tr->ev_done[slot];
    } else {
        // one stream: rider mode still owes the last frame its finish + pack; then the handle's stream holds the frame
        const int rc = flush_pipeline(tr);
        if (rc) return rc;
        if (!tr->ev_frame) LS_HIP(hipEventCreateWithFlags(&tr->ev_frame, hipEventDisableTiming | hipEventDisableSystemFence));
        ev = tr->ev_frame;
    }
    if (src == waiter) return LS_OK;
    LS_HIP(hipEventRecord(ev, src));
    LS_HIP(hipStreamWaitEvent(waiter, ev, 0));
    return LS_OK;
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

}  // extern "C"
