// ls_commit.cpp -- commitScene (ITracer.hpp:87; EmbreeTracer.cpp:290-295 rtcCommitScene; OptixTracer.cpp:263-275, :517-571):
// the geometry layout, the projection engine's group-culling data, and the BVH engine's build / refit / instanced
// hierarchies.
#include "ls_internal.h"

#include <algorithm>
#include <cmath>

namespace lsi {

// (re)build the committed scene arrays (transformed vertices, rebased indices) on the device
// keep_indices: the rebased indices of the last call are still right (same layout, no index upload since: a refit)
int materialize_scene(ls_tracer *tr, bool with_maxabs, bool keep_indices)
{
    int rc;
    if ((rc = ensure(tr, tr->verts, (size_t)tr->n_verts * 3))) return rc;
    const uint32_t *tris_before = tr->tris.p;
    if ((rc = ensure(tr, tr->tris, (size_t)tr->n_tris * 3))) return rc;
    const bool rebase = !(keep_indices && tr->tris_rebased && tr->tris.p == tris_before);
    hipStream_t s = tr->stream;
    if (with_maxabs) LS_HIP(hipMemsetAsync(tr->d_maxabs, 0, 4, s));
    for (const auto &le : tr->layout) {
        auto it = tr->geoms.find(le.name);
        if (it == tr->geoms.end()) return fail(tr, LS_ERR_NOT_COMMITTED, "geometry removed since the last commit");
        Geometry &ge = it->second;
        ls::launch_transform(s, ge.raw(), ge.stride, ge.n_verts, ge.affine, tr->rinv, tr->t,
                             tr->verts.p + 3 * (size_t)le.vfirst, tr->d_maxabs);
        if (rebase) ls::launch_rebase(s, ge.idx(), ge.n_tris * 3, le.vfirst, tr->tris.p + 3 * (size_t)le.tfirst);
    }
    LS_HIP(hipGetLastError());
    tr->tris_rebased = true;
    tr->scene_materialized = true;
    return LS_OK;
}

// When group culling pays: measured, not asserted (round 6; tools/cull_crossover.sh, profiles/r06_cull_crossover.txt, EXPERIMENTS.md
// E8.4 -- one MI355X, SYN-128 over the BASELINE grid formula at other sizes, three frames in flight through the C++ loop, us per frame
// culling off / on):
//   full raster     1.0 M 13.3 / 16.7    1.5 M 17.3 / 18.2    2.0 M 21.9 / 19.2    3.0 M 29.7 / 22.1    5.0 M 45.0 / 27.2    10 M 81.4 / 38.8
//   the same with 32 rings: 1 M 10.2 / 11.9, 2 M 17.0 / 11.8, 5 M 35.0 / 12.9;  64 rings: 1 M 11.3 / 15.3, 2 M 17.9 / 16.4, 5 M 38.4 / 16.8
//   an eighth of a turn (what a rank of eight traces; frames as graphs, ranks 0 / 4):
//                   1.0 M 8.9, 9.2 / 9.3, 9.0    1.5 M 9.8, 9.5 / 9.7, 9.9    2.0 M 11.6, 10.5 / 9.2, 9.4    3.0 M 15.4, 14.8 / 9.0, 9.3
// Without culling a frame grows by 8.6 us per million triangles, with it by 2.5 (the survivors) on top of a fixed 3.4 us at 1 M -- the
// cull pass and the dependent launch behind it: the lines cross between 1.5 M and 2 M whatever the ring spacing (32 rings: near
// 1.2 M; 64 and 128: 1.6 - 1.8 M -- what moves the crossing is the fixed cost, which the rings do not change).  Under a narrow
// shard both ways are bound by the host's 9 us per frame up to 1.5 M (a wash), culling wins from 2 M.
// auto = geometries of 1.75 M triangles or more (round 5: 2 M, two data points); of 512 k or more under a shard narrower than half
// a turn (unchanged: no worse anywhere, and a rank's GPU then reads its sector instead of the whole mesh).
bool cull_enabled(const ls_tracer *tr, const Geometry &g)
{
    if (tr->opt_block_cull != 2) return tr->opt_block_cull != 0;
    if (g.n_tris >= 1750000u) return true;
    double lo, hi;
    return g.n_tris >= 524288u && shard_sector(tr, lo, hi);
}

// Group-culling data of one geometry, brought up to date (stream-ordered on the handle's stream).
int prepare_blocks(ls_tracer *tr, Geometry &g)
{
    const bool want = cull_enabled(tr, g) && g.has_verts && g.has_idx && ls::project_tris_per_wave(g.n_tris) == 64u;
    if (!want) return LS_OK;
    if (!g.order_stale && !g.bounds_stale) return LS_OK;
    // frames in flight on the slot streams read d_idx_sorted / d_perm / d_boxes / d_corners
    int rc;
    if ((rc = flush_pipeline(tr))) return rc;
    ++tr->main_epoch;
    const uint32_t nt = g.n_tris;
    if (!g.d_perm) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_perm), (size_t)nt * 4));
    if (!g.d_idx_sorted) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_idx_sorted), (size_t)nt * 12));
    if (!g.d_boxes) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_boxes), ls::project_box_entries(nt) * 32));   // group bounds, then block bounds
    if (!g.d_corners) LS_HIP(hipMalloc(reinterpret_cast<void **>(&g.d_corners), (size_t)nt * 48));
    if (g.order_stale) {
        if ((rc = ensure(tr, tr->keys_a, nt))) return rc;
        if ((rc = ensure(tr, tr->keys_b, nt))) return rc;
        if ((rc = ensure(tr, tr->vals_a, nt))) return rc;
        if ((rc = ensure(tr, tr->sort_temp, ls::sort_temp_bytes(nt)))) return rc;
        if (!tr->d_aabb6) LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_aabb6), 32));
        tr->bvh_order_valid = false;   // keys_a / keys_b / vals_a are the scratch of this pass
        ++tr->key_scratch_epoch;
        if (!ls::launch_mesh_order(tr->stream, static_cast<const uint8_t *>(g.raw()), g.stride, g.n_verts, g.idx(), nt, tr->d_aabb6,
                                   tr->keys_a.p, tr->keys_b.p, tr->vals_a.p, tr->sort_temp.p, tr->sort_temp.cap, g.d_perm, g.d_idx_sorted))
            return fail(tr, LS_ERR_OUT_OF_RANGE, "the sort scratch is smaller than the mesh order needs");
        g.order_stale = false;
        g.bounds_stale = true;
    }
    if (g.bounds_stale) {
        ls::launch_group_bounds(tr->stream, static_cast<const uint8_t *>(g.raw()), g.stride, g.d_idx_sorted, nt, g.d_boxes);
        ls::launch_corners(tr->stream, static_cast<const uint8_t *>(g.raw()), g.stride, g.d_idx_sorted, g.d_perm, nt, g.d_corners);
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
    constexpr uint32_t kWidenAtOnceLeaves = 65536u;
    const uint32_t g = tr->leaf_size;
    const bool fresh = relayout || !tr->inst_valid || tr->inst_leaf_size != g || tr->inst_layout.size() != order.size() ||
                       (tr->opt_bvh_wide != 0) != tr->wide_valid;   // (the option changed: the twins are made with the hierarchies)
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
        if (tr->opt_bvh_wide && (rc = ensure(tr, tr->wide_nodes, (size_t)nodes + 1))) return rc;
        if ((rc = ensure(tr, tr->range_boxes, (size_t)range + 2))) return rc;
        tr->n_leaves = nodes;
        tr->inst_leaf_size = g;
    }
    bool any = false;
    for (const Geometry *ge : order) any = any || fresh || ge->blas_dirty;
    tr->last_commit_built = any;
    if (any) {
        if ((rc = ensure(tr, tr->inst_verts, (size_t)tr->n_verts * 3))) return rc;
        if ((rc = ensure(tr, tr->keys_a, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->keys_b, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->vals_a, tr->n_tris))) return rc;
        if ((rc = ensure(tr, tr->vals_b, tr->n_tris))) return rc;
        uint32_t biggest = 0;
        for (const Geometry *ge : order) biggest = std::max(biggest, ge->n_tris);
        if ((rc = ensure(tr, tr->sort_temp, ls::sort_temp_bytes(biggest)))) return rc;
        if (!tr->d_inst_maxabs) {
            LS_HIP(hipMalloc(reinterpret_cast<void **>(&tr->d_inst_maxabs), ls::kGeomsPerLaunch * 4));
            LS_HIP(hipMemsetAsync(tr->d_inst_maxabs, 0, ls::kGeomsPerLaunch * 4, tr->stream));
        }
        tr->bvh_order_valid = false;   // the key arrays hold per-geometry slices now
        tr->classic_nodes_valid = false;
        if (fresh) ++tr->key_scratch_epoch;   // the slices moved
        hipStream_t s = tr->stream;
        static const float kIdA[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, kIdR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, kZero[3] = {0, 0, 0};
        // (d_inst_maxabs is zero here: zeroed when it was made and behind every build's read-back, off the next build's path)
        for (size_t i = 0; i < order.size(); ++i) {
            Geometry &ge = *order[i];
            if (!fresh && !ge.blas_dirty) continue;
            const ls_tracer::LayoutEntry &le = tr->layout[i];
            const ls_tracer::InstSlot &sl = tr->inst_layout[i];
            float *verts = tr->inst_verts.p + 3 * (size_t)le.vfirst;
            const uint32_t *tris = ge.idx();   // mesh-local indices are what a per-geometry hierarchy wants: no rebased copy
            uint32_t *ka = tr->keys_a.p + le.tfirst, *kb = tr->keys_b.p + le.tfirst, *vb = tr->vals_b.p + le.tfirst;
            // identity transform: the packed copy holds the vertices as uploaded (1 * x + 0 * y + 0 * z + 0 is x)
            ls::launch_transform(s, ge.raw(), ge.stride, ge.n_verts, kIdA, kIdR, kZero, verts, tr->d_inst_maxabs + i);
            // vertices alone changed and the geometry's sorted keys are still in place: a refit (same order, same
            // topology, every box recomputed) -- what the classic path does with LS_OPT_BVH_REFIT
            const bool refit = tr->opt_bvh_refit && !ge.blas_topo_dirty && ge.blas_sorted_epoch == tr->key_scratch_epoch;
            if (!refit) {
                // the keys, and with them the sort's first tile histograms; the values are the triangles' positions
                ls::launch_morton(s, verts, tris, ge.n_tris, tr->d_inst_maxabs + i, ka, nullptr, ls::sort_first_counts(tr->sort_temp.p, ge.n_tris));
                if (!ls::launch_sort(s, tr->sort_temp.p, tr->sort_temp.cap, ka, kb, nullptr, vb, ge.n_tris, true))
                    return fail(tr, LS_ERR_OUT_OF_RANGE, "the sort scratch is smaller than a geometry's hierarchy build needs");
                ge.blas_sorted_epoch = tr->key_scratch_epoch;
            }
            float4 *rb = tr->range_boxes.p + sl.range_first;
            ls::launch_leaves_tree(s, verts, tris, vb, ge.n_tris, g, tr->records.p + sl.rec_first, sl.rt, rb, true);
            // (a refit finds the topology of these very keys in the nodes: fresh layouts and classic builds in between clear the condition)
            if (refit) ls::launch_refit_nodes(s, sl.n_leaves, sl.rt, rb, tr->nodes.p + sl.node_first);
            else ls::launch_hierarchy(s, kb, sl.n_leaves, g, sl.rt, rb, tr->nodes.p + sl.node_first);
            // The four-wide twins the trace walks (LS_OPT_BVH_WIDE) follow every build and every refit of this geometry -- at
            // once for a small hierarchy (a moving instance's refit: a few microseconds), LAZILY for a big one: k_widen costs 34 us
            // per million nodes, the wide walk wins ~13 us per frame there, and a hierarchy that is rebuilt or refitted every frame
            // would pay the first without ever collecting the second.  The trace makes them once a hierarchy has survived
            // kWidenAfterFrames frames (ls_trace.cpp) and walks the binary nodes until then.
            ls_tracer::InstSlot &slw = tr->inst_layout[i];
            slw.wide_made = false;
            slw.wide_age = 0;
            if (tr->opt_bvh_wide && sl.n_leaves <= kWidenAtOnceLeaves) {
                ls::launch_widen(s, tr->nodes.p + sl.node_first, sl.n_leaves, tr->wide_nodes.p + sl.node_first);
                slw.wide_made = true;
            }
        }
        tr->wide_valid = tr->opt_bvh_wide != 0;
        // a scene of one geometry: the top of its hierarchy for the trace grid's LDS (static with the hierarchy)
        tr->treelet_valid = false;
        static const bool no_treelet = tune_int("LS_TRACE_NO_TREELET", 0) != 0;
        if (order.size() == 1 && tr->inst_layout[0].n_leaves > 1u && !no_treelet && !tr->opt_bvh_wide) {   // (the binary walk's: the four-wide walk does not stage it)
            if ((rc = ensure(tr, tr->treelet, (size_t)ls::kTreeletNodes))) return rc;
            ls::launch_treelet(s, tr->nodes.p + tr->inst_layout[0].node_first, tr->inst_layout[0].n_leaves, tr->treelet.p);
            tr->treelet_valid = true;
        }
        LS_HIP(hipGetLastError());
        // the extent of every rebuilt mesh (the widening of its boxes at trace time is scaled by it)
        uint32_t bits[ls::kGeomsPerLaunch];
        LS_HIP(hipMemcpyAsync(bits, tr->d_inst_maxabs, sizeof(bits), hipMemcpyDeviceToHost, s));
        LS_HIP(hipStreamSynchronize(s));
        LS_HIP(hipMemsetAsync(tr->d_inst_maxabs, 0, ls::kGeomsPerLaunch * 4, s));   // for the next build
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

// Index uploads since the last commit: their maxima come back (one wait for all of them), and a geometry whose triangles
// name a vertex it does not have fails the commit -- LS_ERR_OUT_OF_RANGE, nothing is built or traced over it, the handle
// stays usable, new indices (or ls_remove_geometry) clear the condition.  Reference: shared buffers, no check
// (EmbreeTracer.cpp:140-176); there a bad index is a wild host read, here it would be a device memory fault.
static int check_geometry_indices(ls_tracer *tr)
{
    std::vector<Geometry *> pending;
    for (auto &kv : tr->geoms)
        if (kv.second.idx_unchecked && kv.second.d_idx_max) pending.push_back(&kv.second);
    if (!pending.empty()) {
        std::vector<uint32_t> maxima(pending.size(), 0u);
        for (size_t i = 0; i < pending.size(); ++i)
            LS_HIP(hipMemcpyAsync(&maxima[i], pending[i]->d_idx_max, 4, hipMemcpyDeviceToHost, tr->stream));
        LS_HIP(hipStreamSynchronize(tr->stream));
        for (size_t i = 0; i < pending.size(); ++i) {
            Geometry &g = *pending[i];
            g.idx_unchecked = false;
            g.idx_bad = g.n_tris > 0 && maxima[i] >= g.n_verts;
            g.idx_bad_value = maxima[i];
        }
    }
    for (auto &kv : tr->geoms) {
        const Geometry &g = kv.second;
        if (g.idx_bad && g.has_idx)
            return fail(tr, LS_ERR_OUT_OF_RANGE, ("geometry '" + g.name + "': vertex index " + std::to_string(g.idx_bad_value) + " in its triangles, " +
                                                      std::to_string(g.n_verts) + " vertices registered -- refused (it would fault the device)").c_str());
    }
    return LS_OK;
}

int commit_locked(ls_tracer *tr)
{
    tr->committed = false;
    tr->traced = false;
    tr->bvh_built = false;
    tr->scene_materialized = false;
    {
        const int rc = check_geometry_indices(tr);
        if (rc) return rc;
    }
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
        __atomic_store_n(tr->h_status + 1, 0u, __ATOMIC_RELAXED);   // (the survivor hint spoke for the old set of geometries)
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
        if ((rc = materialize_scene(tr, !refit, refit))) return rc;   // (a refit: same layout, same indices as the last build's; nobody reads the extent: k_morton is not run)
        mark(tr, 1);
        if (!refit) ls::launch_morton(s, tr->verts.p, tr->tris.p, nt, tr->d_maxabs, tr->keys_a.p, nullptr, ls::sort_first_counts(tr->sort_temp.p, nt));
        mark(tr, 2);
        if (!refit && !ls::launch_sort(s, tr->sort_temp.p, tr->sort_temp.cap, tr->keys_a.p, tr->keys_b.p, nullptr, tr->vals_b.p, nt, true))
            return fail(tr, LS_ERR_OUT_OF_RANGE, "the sort scratch is smaller than the hierarchy build needs");
        mark(tr, 3);
        tr->bvh_order_valid = true;
        tr->bvh_order_tris = nt;
        tr->last_commit_refit = refit;
        ls::launch_leaves_tree(s, tr->verts.p, tr->tris.p, tr->vals_b.p, nt, g, tr->records.p, tr->rt, tr->range_boxes.p, false);
        mark(tr, 4);
        mark(tr, 5);
        if (refit && tr->classic_nodes_valid) ls::launch_refit_nodes(s, L, tr->rt, tr->range_boxes.p, tr->nodes.p);
        else ls::launch_hierarchy(s, tr->keys_b.p, L, g, tr->rt, tr->range_boxes.p, tr->nodes.p);
        tr->classic_nodes_valid = true;
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

}  // namespace lsi

using namespace lsi;

extern "C" {

int ls_commit_scene(ls_tracer *tr)
{
    LS_ENTER(tr);
    return commit_locked(tr);
}

}  // extern "C"
