// ls_debug.cpp -- include/lidarshooter_hip_debug.h: views into the handle for tests/ and bench.py (dense per-ray results,
// the exhaustive device checker, the committed scene and BVH arrays).  Not part of the drop-in surface.
#include "../../include/lidarshooter_hip_debug.h"
#include "ls_internal.h"

using namespace lsi;

extern "C" {

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
        ls::launch_dense_from_hits(tr->stream, tables(tr), tr->last_d_hits, tr->last_d_n, geom_table(tr), tr->hit_t.p, tr->hit_gid.p);
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

int ls_debug_sort_pairs(ls_tracer *tr, uint32_t *keys, uint32_t *vals, uint32_t n)
{
    LS_ENTER(tr);
    if ((!keys || !vals) && n) return fail(tr, LS_ERR_INVALID_ARGUMENT, "null argument");
    if (!n) return LS_OK;
    uint32_t *d = nullptr;
    void *temp = nullptr;
    const size_t tb = ls::sort_temp_bytes(n);
    LS_HIP(hipMalloc(reinterpret_cast<void **>(&d), (size_t)n * 16));
    hipError_t e = hipMalloc(&temp, tb);
    if (e == hipSuccess) e = hipMemcpy(d, keys, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, vals, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        e = ls::launch_sort(tr->stream, temp, tb, d, d + 2 * (size_t)n, d + n, d + 3 * (size_t)n, n) ? hipStreamSynchronize(tr->stream)
                                                                                                       : hipErrorInvalidValue;
    }
    if (e == hipSuccess) e = hipMemcpy(keys, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(vals, d + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)hipFree(temp);
    if (e != hipSuccess) return fail(tr, LS_ERR_HIP, hipGetErrorString(e));
    return LS_OK;
}

}  // extern "C"

int ls_debug_expand_hits(void *dst_points32, const void *hits8, uint32_t n, const float *sin_theta, const float *cos_theta,
                         const float *cs_phi, uint32_t V, uint32_t H)
{
    if (!n) return LS_OK;
    if (!dst_points32 || !hits8 || !sin_theta || !cos_theta || !cs_phi || !V || !H) return LS_ERR_INVALID_ARGUMENT;
    lsi::expand_hits_range(static_cast<uint8_t *>(dst_points32), static_cast<const uint8_t *>(hits8), n, sin_theta, cos_theta, cs_phi, V, H);
    return LS_OK;
}

