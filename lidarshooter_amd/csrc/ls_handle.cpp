// ls_handle.cpp -- the tracer handle: create / destroy (EmbreeTracer::create, EmbreeTracer.cpp:10-70), the sensor's
// ray-direction tables (LidarDevice.cpp:294-342), options, facts, shard / stream / output-buffer plumbing and the small
// stand-alone calls (ray generation, cloud to world, expansion of gathered hit slots).  No compute happens on the host:
// if there is no usable HIP device the create call fails (LS_ERR_NO_DEVICE); there is no CPU fallback in this library.
#include "ls_internal.h"

#include <algorithm>
#include <cmath>

namespace lsi {

namespace {

// LidarDevice.cpp:306-316 on the host: the V+H distinct angles of a revolution go through libm
// (sinf/cosf, exactly like the reference's CPU path); the kernels only multiply table entries.
void fill_tables(ls_tracer *tr, std::vector<float> &tab)
{
    const uint32_t V = tr->V, H = tr->H;
    tab.resize(2 * (size_t)V + 2 * (size_t)H + 3 * (size_t)V + 2 * (size_t)H + (size_t)V + ls::kCullLutBuckets / 2);
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
        static const double emargin = tune_int("LS_ELEV_MARGIN_MICRODEG", (int)std::lround(kProjectElevMarginDeg * 1e6)) * 1e-6;
        const double hi = chi + emargin, lo = chi - emargin;
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
    // k_cull's look-up table over tan(elevation): bucket b starts at t0 + b / scale; lut[b] = first position whose
    // tan_up is at or above the bucket's start.  Usable when every two consecutive buckets hold at most two channels
    // (the kernel starts one bucket early and steps at most twice); the last bucket reaches to +infinity.
    uint16_t *lut = reinterpret_cast<uint16_t *>(rank + V);
    tr->lut_ok = false;
    tr->lut_t0 = tr->lut_scale = 0.0f;
    const uint32_t nb = ls::kCullLutBuckets;
    const float first = up[0], last = up[V - 1];
    if (V <= 65535u && std::isfinite(first) && std::isfinite(last) && last > first) {
        const double t0 = first, width = ((double)last - (double)first) * (1.0 + 1e-6) / (double)(nb - 1);   // `last` falls into bucket nb - 2 or nb - 1
        const float scale = (float)(1.0 / width);
        std::vector<uint32_t> start(nb + 1);
        uint32_t i = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            // the bucket start as the kernel's arithmetic sees it: x = (v - t0) * scale lands in bucket b for v >= s_b
            const double s_b = t0 + (double)b * width;
            while (i < V && (double)up[i] < s_b) ++i;
            start[b] = i;
            lut[b] = (uint16_t)i;
        }
        start[nb] = V;
        bool ok = true;
        for (uint32_t b = 0; b < nb && ok; ++b) ok = start[std::min(b + 2u, nb)] - start[b] <= 2u;
        tr->lut_ok = ok;
        tr->lut_t0 = first;
        tr->lut_scale = scale;
    }
}

}  // namespace

}  // namespace lsi

using namespace lsi;

extern "C" {

int ls_abi_version(void) { return LS_ABI_VERSION; }

// (ls_source_hash.inc is written by the Makefile from the sources this binary is made of; the marker lets a script read the
// hash out of the file without loading it)
static const char kSourceHash[] = "LS_SOURCE_HASH="
#include "ls_source_hash.inc"
    ;
const char *ls_source_hash(void) { return kSourceHash + 15; }

// the sensor's tables on the device and on the host, from tr's sensor fields (create, ls_tracer_set_sensor*)
static int upload_tables(ls_tracer *tr)
{
    std::vector<float> tab;
    fill_tables(tr, tab);
    float *d = nullptr;
    LS_HIP(hipMalloc(reinterpret_cast<void **>(&d), tab.size() * 4));
    if (hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(tr, LS_ERR_HIP, "uploading the sensor tables failed");
    }
    if (tr->d_tables) (void)hipFree(tr->d_tables);
    tr->d_tables = d;
    tr->host_tables = tab;
    return LS_OK;
}

static void check_projection_ok(ls_tracer *tr)
{
    tr->projection_ok = true;
    for (float chi : tr->vertical)
        if (!(chi >= -90.0f && chi <= 90.0f)) tr->projection_ok = false;  // elevation == channel angle only there
    if (!std::isfinite(tr->h_begin) || !std::isfinite(tr->h_end)) tr->projection_ok = false;
    if (tr->V > 32767u) tr->projection_ok = false;  // channel range is packed into 16+15 bits
}

// the sensor fields of `tr` from a descriptor / from given tables (validated by the caller)
static void take_sensor_desc(ls_tracer *tr, const ls_sensor_desc *sd)
{
    tr->V = sd->n_vertical;
    tr->H = sd->h_count;
    tr->vertical.assign(sd->vertical_deg, sd->vertical_deg + sd->n_vertical);
    tr->given_tables.clear();
    tr->h_begin = sd->h_begin;
    tr->h_end = sd->h_end;
    tr->h_step = (sd->h_end - sd->h_begin) / static_cast<float>(sd->h_count - 1u);  // LidarDevice.cpp:611
    std::memcpy(tr->rinv, sd->Rinv, sizeof(tr->rinv));
    std::memcpy(tr->t, sd->t, sizeof(tr->t));
}

static void take_sensor_tables(ls_tracer *tr, const ls_sensor_tables *st)
{
    const uint32_t V = tr->V = st->n_vertical, H = tr->H = st->h_count;
    // (the elevations the tables stand for -- check_tables -- not the caller's description of them; float keeps them to 4e-6 degrees)
    tr->vertical.resize(V);
    for (uint32_t v = 0; v < V; ++v) tr->vertical[v] = static_cast<float>(std::atan2((double)st->cos_theta[v], (double)st->sin_theta[v]) * 180.0 / M_PI);
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
}

// The angles the caller's factor tables really stand for, in double: a channel's elevation above the horizon is
// atan2(cos theta, sin theta), a column's azimuth atan2(sin phi, cos phi).  The footprint bounds of the projection engine are
// built from THESE (with the elevation slack at 2e-4 degrees an elevation_deg that is off by more than that
// would silently drop hits at ring boundaries -- a wrong cloud with no error); what the caller says in elevation_deg /
// h_begin_deg / h_step_deg is a cross-check, and a description that is grossly off its own tables is refused.
constexpr double kGivenElevationToleranceDeg = 0.01;   // caller's elevation_deg against the derived one
constexpr double kGivenAzimuthToleranceDeg = 0.002;    // h_begin_deg + h * h_step_deg against the derived azimuth (the bounds carry 0.005 degrees + 1/16 column)

static const char *check_tables(const ls_sensor_tables *st)
{
    if (!(st && st->sin_theta && st->cos_theta && st->elevation_deg && st->sin_phi && st->cos_phi && st->n_vertical != 0 && st->h_count >= 2))
        return "incomplete sensor tables";
    for (uint32_t v = 0; v < st->n_vertical; ++v) {
        const double s = st->sin_theta[v], c = st->cos_theta[v];
        if (!std::isfinite(s) || !std::isfinite(c) || std::fabs(std::hypot(s, c) - 1.0) > 1e-3) return "sin_theta / cos_theta are not the sine and cosine of one angle";
        const double derived = std::atan2(c, s) * 180.0 / M_PI;
        if (!(std::fabs(derived - (double)st->elevation_deg[v]) <= kGivenElevationToleranceDeg))
            return "elevation_deg does not describe sin_theta / cos_theta (more than 0.01 degrees off atan2(cos_theta, sin_theta))";
    }
    for (uint32_t h = 0; h < st->h_count; ++h) {
        const double s = st->sin_phi[h], c = st->cos_phi[h];
        if (!std::isfinite(s) || !std::isfinite(c) || std::fabs(std::hypot(s, c) - 1.0) > 1e-3) return "sin_phi / cos_phi are not the sine and cosine of one angle";
        const double said = (double)st->h_begin_deg + (double)st->h_step_deg * (double)h;
        if (!std::isfinite(said)) return "h_begin_deg / h_step_deg are not finite";
        const double off = std::remainder(std::atan2(s, c) * 180.0 / M_PI - said, 360.0);
        if (!(std::fabs(off) <= kGivenAzimuthToleranceDeg)) return "h_begin_deg / h_step_deg do not describe sin_phi / cos_phi (more than 0.002 degrees off)";
    }
    return nullptr;
}

// A handle takes another sensor -- raster, channel tables, pose -- and keeps its geometries (ITracer::setSensorConfig,
// ITracer.cpp:48: EmbreeTracer::traceScene reads _config every frame, EmbreeTracer.cpp:299-307, so a swapped or
// re-initialised LidarDevice takes effect at the next trace).  Everything in flight completes first; the shard goes back
// to the full turn; a committed scene is committed again for the new sensor (the classic BVH lives in the sensor frame).
// the sensor fields take_sensor_* overwrite, kept until the new tables are on the device
struct SensorFields {
    uint32_t V, H;
    std::vector<float> vertical, given_tables;
    float h_begin, h_end, h_step, rinv[9], t[3];
    explicit SensorFields(const ls_tracer *tr)
        : V(tr->V), H(tr->H), vertical(tr->vertical), given_tables(tr->given_tables), h_begin(tr->h_begin), h_end(tr->h_end), h_step(tr->h_step)
    {
        std::memcpy(rinv, tr->rinv, sizeof(rinv));
        std::memcpy(t, tr->t, sizeof(t));
    }
    void restore(ls_tracer *tr)
    {
        tr->V = V; tr->H = H;
        tr->vertical.swap(vertical);
        tr->given_tables.swap(given_tables);
        tr->h_begin = h_begin; tr->h_end = h_end; tr->h_step = h_step;
        std::memcpy(tr->rinv, rinv, sizeof(rinv));
        std::memcpy(tr->t, t, sizeof(t));
    }
};

// `before`: the handle's sensor as it was; the new one has been taken over already.  The tables go up FIRST: a failed
// upload (hipMalloc, the copy) puts the old sensor back -- its tables are still on the device, its shard and frame graphs
// untouched -- instead of leaving the new V and H over the old, smaller table buffer.
static int sensor_changed(ls_tracer *tr, SensorFields &before)
{
    int rc = upload_tables(tr);
    if (rc) { before.restore(tr); return rc; }
    frame_graph_destroy(tr);
    tr->az0 = 0;
    tr->naz = tr->H;
    check_projection_ok(tr);
    if (tr->engine == 2 && !tr->projection_ok) tr->engine = 0;   // (the projection engine was asked for and no longer applies)
    tr->keys_armed = tr->keys_b_armed = tr->keys_c_armed = false;
    tr->traced = false;
    tr->pack_split = 0;
    tr->begin_open = tr->progress_active = false;
    tr->layout_dirty = true;          // every sensor-frame product (materialised scene, sensor-centred Morton order) is stale
    tr->scene_materialized = false;
    tr->bvh_order_valid = false;
    tr->classic_nodes_valid = false;
    if (tr->committed) {
        tr->committed = false;
        rc = commit_locked(tr);
        if (rc < -1) return rc;
    }
    return LS_OK;
}

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
    if (upload_tables(tr) != LS_OK) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_maxabs), 4) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_visits), 32) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_n_points), 4) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_queue_heads), ls::kQueues * 16 * sizeof(uint32_t)) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void **>(&tr->d_big_count), 4 * ls::kCounterSlotWords * sizeof(uint32_t)) != hipSuccess) return bail(LS_ERR_HIP);
    check_projection_ok(tr);
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
        // round 6: a wave refills only when ALL its lanes are idle (64; 56 until now): frame 138.5 -> 127.8 us with the four-wide
        // walk at SYN-1M, 344.6 -> 339.1 at configs[4]'s size (tools/sweep_wide.sh, sweep_wide2.sh; 24 / 40 / 48: 141 / 138 / 134)
        tr->chan_mul = 1u;
        tr->refill_min = 64u;
        { const uint32_t v = (uint32_t)tune_int("LS_TRACE_CHAN_MUL", 1); if (v && gcd(v, tr->V) == 1u) tr->chan_mul = v; }
        { const int v = tune_int("LS_TRACE_REFILL_MIN", 64); if (v >= 1 && v <= 64) tr->refill_min = (uint32_t)v; }
    }
    // (coherent and mapped, explicitly: the host polls these words while kernels are still running)
    constexpr unsigned kHostWords = hipHostMallocMapped | hipHostMallocCoherent;
    if (hipHostMalloc(reinterpret_cast<void **>(&tr->h_n_points), 16, kHostWords) != hipSuccess) return bail(LS_ERR_HIP);
    if (hipHostMalloc(reinterpret_cast<void **>(&tr->h_status), 16, kHostWords) != hipSuccess) return bail(LS_ERR_HIP);
    tr->h_status[0] = tr->h_status[1] = tr->h_status[2] = tr->h_status[3] = 0u;   // [0] the sticky status, [1] the survivor hint (ls_trace.cpp)
    if (hipHostMalloc(reinterpret_cast<void **>(&tr->h_progress), sizeof(ls::HostProgress), kHostWords) != hipSuccess) return bail(LS_ERR_HIP);
    std::memset(tr->h_progress, 0, sizeof(ls::HostProgress));

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
    take_sensor_desc(tr, sd);
    return create_device_state(tr, hip_device, out);
}

int ls_tracer_set_sensor(ls_tracer *tr, const ls_sensor_desc *sd)
{
    LS_ENTER(tr);
    if (!sd || !sd->vertical_deg || sd->n_vertical == 0 || sd->h_count < 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "incomplete sensor descriptor");
    if ((unsigned long long)sd->n_vertical * sd->h_count > 0x7FFFFFFFull) return fail(tr, LS_ERR_OUT_OF_RANGE, "ray indices are 32-bit");
    if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open");
    // (another sensor puts the handle back on the full turn: whoever installed output buffers sized them for the old raster,
    // or for a shard of it -- a group -- and has to take them back first)
    if (tr->ext_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "external output buffers are installed (a group holds this tracer?): reset them before changing the sensor");
    const int rc = flush_pipeline(tr);
    if (rc) return rc;
    LS_HIP(hipStreamSynchronize(tr->stream));
    SensorFields before(tr);
    take_sensor_desc(tr, sd);
    return sensor_changed(tr, before);
}

int ls_tracer_set_sensor_tables(ls_tracer *tr, const ls_sensor_tables *st)
{
    LS_ENTER(tr);
    if (const char *why = check_tables(st)) return fail(tr, LS_ERR_INVALID_ARGUMENT, why);
    if ((unsigned long long)st->n_vertical * st->h_count > 0x7FFFFFFFull) return fail(tr, LS_ERR_OUT_OF_RANGE, "ray indices are 32-bit");
    if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open");
    // (another sensor puts the handle back on the full turn: whoever installed output buffers sized them for the old raster,
    // or for a shard of it -- a group -- and has to take them back first)
    if (tr->ext_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "external output buffers are installed (a group holds this tracer?): reset them before changing the sensor");
    const int rc = flush_pipeline(tr);
    if (rc) return rc;
    LS_HIP(hipStreamSynchronize(tr->stream));
    SensorFields before(tr);
    take_sensor_tables(tr, st);
    return sensor_changed(tr, before);
}

int ls_tracer_create_tables(const ls_sensor_tables *st, int hip_device, ls_tracer **out)
{
    if (!out) return LS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (check_tables(st)) return LS_ERR_INVALID_ARGUMENT;
    if ((unsigned long long)st->n_vertical * st->h_count > 0x7FFFFFFFull) return LS_ERR_OUT_OF_RANGE;
    ls_tracer *tr = new ls_tracer();
    take_sensor_tables(tr, st);
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
    frame_graph_destroy(tr);
    release(tr->best_keys_b); release(tr->big_queue_b); release(tr->points_b); release(tr->hits_b); release(tr->pack_status); release(tr->pack_status_ms);
    release(tr->best_keys_c); release(tr->big_queue_c); release(tr->points_c); release(tr->hits_c);
    if (tr->d_n_points_b) (void)hipFree(tr->d_n_points_b);
    if (tr->d_n_points_c) (void)hipFree(tr->d_n_points_c);
    if (tr->ev_main) (void)hipEventDestroy(tr->ev_main);
    if (tr->ev_frame) (void)hipEventDestroy(tr->ev_frame);
    for (int i = 0; i < 3; ++i) {
        if (tr->ev_done[i]) (void)hipEventDestroy(tr->ev_done[i]);
        bool dup = false;
        for (int k = 0; k < i; ++k) dup = dup || tr->slot_stream[k] == tr->slot_stream[i];
        if (tr->slot_stream[i] && !dup) (void)hipStreamDestroy(tr->slot_stream[i]);
    }
    release(tr->verts); release(tr->tris); release(tr->keys_a); release(tr->keys_b); release(tr->vals_a);
    release(tr->vals_b); release(tr->geom_table); release(tr->sort_temp); release(tr->records);
    release(tr->inst_verts); release(tr->treelet); release(tr->wide_nodes);
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
    if (tr->h_progress) (void)hipHostFree(tr->h_progress);
    for (auto &r : tr->trec)
        for (auto &e : r.ev)
            if (e) (void)hipEventDestroy(e);
    if (tr->own_stream) (void)hipStreamDestroy(tr->own_stream);
    delete tr;
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
    __atomic_store_n(tr->h_status + 1, 0u, __ATOMIC_RELAXED);   // (the survivor hint spoke for the old shard; a frame still running may write it once more: a hint, not a promise)
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
    case LS_INFO_HOST_THREADS: return host_pool_threads();
    case LS_INFO_AZIMUTH_COUNT: return (long)tr->H;
    case LS_INFO_LAST_COMMIT_REFIT: return (!tr->bvh_inst && tr->last_commit_refit) ? 1 : 0;
    case LS_INFO_BVH_INSTANCED: return tr->bvh_inst ? (tr->last_commit_built ? 2 : 1) : 0;
    case LS_INFO_NEXT_SLOT: return tr->opt_pipeline == 2 ? (long)(tr->ms_seq % 3u) : 0;
    case LS_INFO_FRAME_GRAPH_STATE: return tr->opt_frame_graph ? (tr->fg_broken ? 2 : 1) : 0;
    case LS_INFO_FRAME_GRAPH_CAPTURES: return (long)tr->fg_captures;
    case LS_INFO_FRAME_GRAPH_REPLAYS: return (long)tr->fg_replays;
    case LS_INFO_FRAME_GRAPH_PATCHES: return (long)tr->fg_patches;
    case LS_INFO_FRAME_GRAPH_LAST_PATCHED: return (long)tr->fg_last_patched;
    case LS_INFO_EMIT_POINTS: return (long)tr->opt_emit_points;
    case LS_INFO_FRAME_GRAPH_PATCH_WAITS: return (long)tr->fg_patch_waits;
    case LS_INFO_BVH_WIDE: return (tr->bvh_inst && tr->wide_valid && tr->wide_in_use) ? 1 : 0;
    default: return fail(tr, LS_ERR_INVALID_ARGUMENT, "unknown info key");
    }
}

int ls_tracer_set_output_buffers(ls_tracer *tr, void *d_points32, void *d_hits, uint32_t *d_n_points, uint32_t capacity)
{
    LS_ENTER(tr);
    tr->ext_hits_only = false;
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

int ls_tracer_set_hit_buffers(ls_tracer *tr, void *d_hits, uint32_t *d_n_points, uint32_t capacity)
{
    LS_ENTER(tr);
    if (!d_hits) {
        tr->ext_points = tr->ext_hits = nullptr;
        tr->ext_n_points = nullptr;
        tr->ext_capacity = 0;
        tr->ext_hits_only = false;
        return LS_OK;
    }
    if (!d_n_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "the count word is required");
    if (tr->opt_emit_points) return fail(tr, LS_ERR_INVALID_ARGUMENT, "hit buffers alone need LS_OPT_EMIT_POINTS = 0 first");
    tr->ext_points = d_hits;   // ("external buffers are installed"; never written: no pass emits points with LS_OPT_EMIT_POINTS = 0)
    tr->ext_hits = d_hits;
    tr->ext_n_points = d_n_points;
    tr->ext_capacity = capacity;
    tr->ext_hits_only = true;
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
    case LS_OPT_UPLOAD_MODE:
        if (value < 0 || value > 2) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_UPLOAD_MODE: 0, 1 or 2");
        tr->opt_upload_mode = value;
        return LS_OK;
    case LS_OPT_DEBUG_FAULT: tr->opt_debug_fault = value != 0; return LS_OK;
    case LS_OPT_BVH_REFIT: tr->opt_bvh_refit = value != 0; return LS_OK;
    case LS_OPT_BVH_INSTANCED: tr->opt_bvh_instanced = value != 0; tr->committed = false; return LS_OK;   // takes effect at the next commit
    case LS_OPT_BVH_WIDE: tr->opt_bvh_wide = value != 0; tr->committed = false; return LS_OK;             // the same
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
    case LS_OPT_EMIT_POINTS:
        if (value < 0 || value > 1) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_EMIT_POINTS: 0 or 1");
        if (value && tr->ext_hits_only) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_EMIT_POINTS = 1 while hit buffers alone are installed (ls_tracer_set_hit_buffers): there is no point buffer");
        tr->opt_emit_points = value;
        return LS_OK;
    case LS_OPT_FRAME_GRAPH: {
        if (value < 0 || value > 1) return fail(tr, LS_ERR_INVALID_ARGUMENT, "LS_OPT_FRAME_GRAPH: 0 or 1");
        if (tr->fg_open) return fail(tr, LS_ERR_INVALID_ARGUMENT, "a frame graph is open");
        if (value == tr->opt_frame_graph) return LS_OK;
        const int rc = flush_pipeline(tr);
        if (rc) return rc;
        LS_HIP(hipStreamSynchronize(tr->stream));
        frame_graph_destroy(tr);
        tr->fg_broken = false;
        tr->opt_frame_graph = value;
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

int ls_expand_gathered_hits_sized(ls_tracer *tr, void *hip_stream, const void *d_gathered, uint32_t world, uint32_t gathered_capacity,
                                  void *d_points32, void *d_hits, uint32_t *d_n_points, void *host_stat64, uint32_t epoch)
{
    LS_ENTER(tr);
    if (!d_gathered || !d_points32 || !d_hits || !d_n_points || !world || !gathered_capacity)
        return fail(tr, LS_ERR_INVALID_ARGUMENT, "null argument");
    static_assert(sizeof(ls::GatherStat) == 64, "one line of pinned host memory");
    ls::launch_expand_slots(hip_stream ? static_cast<hipStream_t>(hip_stream) : tr->stream, tables(tr),
                            static_cast<const uint32_t *>(d_gathered), world, gathered_capacity, 16u + 4u * gathered_capacity,
                            static_cast<uint8_t *>(d_points32), d_hits, d_n_points, static_cast<ls::GatherStat *>(host_stat64), epoch);
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

int ls_generate_rays_aos(ls_tracer *tr, void *d_rays, void *d_hits)
{
    LS_ENTER(tr);
    if (!d_rays && !d_hits) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null outputs");
    ls::launch_raygen_aos(tr->stream, tables(tr), d_rays, d_hits);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

}  // extern "C"
