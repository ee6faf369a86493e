// ls_registry.cpp -- ITracer's geometry bookkeeping behind the C ABI: addGeometry / removeGeometry / updateGeometry and the
// per-name getters (EmbreeTracer.cpp:82-113, :115-288, :369-439; OptixTracer.cpp:63-261), and how mesh data reaches HBM.
#include "ls_internal.h"
#include "ls_host_pool.h"

#include <algorithm>
#include <cmath>
#include <functional>

namespace lsi {

namespace {

// host copy / DMA granularity of a staged upload (LS_OPT_UPLOAD_MODE = 0; tools/upload_sweep.py sweeps them)
const size_t kCopyChunk = (size_t)std::max(16, tune_int("LS_COPY_CHUNK_KB", 512)) << 10;
const size_t kDmaRun = (size_t)std::max(1, tune_int("LS_DMA_RUN", 4));

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

}  // namespace

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

namespace {

// Host memory -> device, without waiting for the device: the copy pool moves the caller's (pageable) bytes into
// a pinned staging buffer chunk by chunk, and the calling thread enqueues each chunk's DMA on the handle's
// stream as soon as the chunk is staged, so copying and DMA overlap.  When the call returns the caller's
// memory is free again (MeshProjector.cpp:448-461 reuses it); the staging buffer is protected by `ev`.
int stage_upload(ls_tracer *tr, void *&stage, size_t &stage_cap, hipEvent_t &ev, void *d_dst, const void *src, size_t bytes)
{
    if (!bytes) return LS_OK;
    if (tr->opt_upload_mode == 1) {
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
    if (tr->opt_upload_mode == 2) {   // ablation: one thread, one DMA
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

// the triangles' indices just changed (an upload, a hand-over, a quad conversion -- all enqueued on the handle's stream):
// their maximum is reduced behind them; commit_locked looks at it
int check_indices_later(ls_tracer *tr, Geometry &g)
{
    if (!g.d_idx_max) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_idx_max), 4));
    ls::launch_index_max(tr->stream, g.idx(), g.n_tris * 3u, g.d_idx_max);
    LS_HIP(hipGetLastError());
    g.idx_unchecked = true;
    g.idx_bad = false;
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
            g.has_idx = true; g.idx_dirty = true; tr->tris_rebased = false; g.order_stale = true; g.blas_dirty = g.blas_topo_dirty = true;
        } else if (idx) { g.shared_idx = idx; g.has_idx = true; g.idx_dirty = true; tr->tris_rebased = false; g.order_stale = true; g.blas_dirty = g.blas_topo_dirty = true; }
        if (idx) return check_indices_later(tr, g);   // (the caller's buffer may hold anything: every hand-over is looked at)
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
        g.idx_dirty = true; tr->tris_rebased = false;
        return check_indices_later(tr, g);
    }
    return LS_OK;
}

}  // namespace

void free_geometry(Geometry &g)
{
    if (g.d_raw) (void)hipFree(g.d_raw);
    if (g.d_idx) (void)hipFree(g.d_idx);
    if (g.d_quad_idx) (void)hipFree(g.d_quad_idx);
    g.d_quad_idx = nullptr;
    if (g.d_idx_max) (void)hipFree(g.d_idx_max);
    g.d_idx_max = nullptr;
    if (g.h_stage_v) (void)hipHostFree(g.h_stage_v);
    if (g.h_stage_i) (void)hipHostFree(g.h_stage_i);
    if (g.d_perm) (void)hipFree(g.d_perm);
    if (g.d_idx_sorted) (void)hipFree(g.d_idx_sorted);
    if (g.d_boxes) (void)hipFree(g.d_boxes);
    if (g.d_corners) (void)hipFree(g.d_corners);
    g.d_perm = g.d_idx_sorted = nullptr;
    g.d_boxes = g.d_corners = nullptr;
    if (g.ev_stage_v) (void)hipEventDestroy(g.ev_stage_v);
    if (g.ev_stage_i) (void)hipEventDestroy(g.ev_stage_i);
    g.d_raw = nullptr;
    g.d_idx = nullptr;
    g.h_stage_v = g.h_stage_i = nullptr;
    g.stage_v_cap = g.stage_i_cap = 0;
    g.ev_stage_v = g.ev_stage_i = nullptr;
}

}  // namespace lsi

using namespace lsi;

extern "C" {

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

// EmbreeTracer::getGeometryType (EmbreeTracer.cpp:103-113)
int ls_geometry_type(ls_tracer *tr, const char *name)
{
    LS_ENTER(tr);
    auto it = tr->geoms.find(name ? name : "");
    if (it == tr->geoms.end()) return fail(tr, LS_ERR_UNKNOWN_GEOMETRY, "geometry key does not exist");
    return it->second.quad ? LS_GEOMETRY_TYPE_QUAD : LS_GEOMETRY_TYPE_TRIANGLE;
}

}  // extern "C"
