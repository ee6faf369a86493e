/*
 * lidarshooter_hip_debug.h -- test and measurement hooks of liblidarshooter_hip.so: views into a tracer handle that
 * tests/ and bench.py use to compare the device's results with the CPU oracle.  Not part of the drop-in surface of
 * include/lidarshooter_hip.h (no ITracer virtual maps to any of these) and not needed by the adapter.
 */
#ifndef LIDARSHOOTER_HIP_DEBUG_H
#define LIDARSHOOTER_HIP_DEBUG_H

#include "lidarshooter_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fault injection (values live in the option / flag spaces of the public headers, their names do not) --------- */
/* ls_tracer_set_option: 1 makes the next pipelined frame publish a wrong epoch, so that the chained prefix gives up and
 * the device status word is raised (one frame). */
#define LS_OPT_DEBUG_FAULT 9
/* ls_group_create_opts (include/lidarshooter_group.h): the gathered arrangement answers are treated as if a peer had
 * answered "neither" -- a one-rank test takes the disagreement path.  (With two real ranks tests/shim/ makes a peer that
 * really refuses.) */
#define LS_GROUP_FLAG_DEBUG_PEER_REFUSES 0x100u

/* Dense per-ray results of the last trace, host buffers of n_rays entries (shard-local order
 * q = v*n_az + (h-first_az)): t (< 0 = miss) and global triangle id (0xFFFFFFFF = miss). */
int ls_debug_dense_hits(ls_tracer *tr, float *t, uint32_t *gid);

/* Exhaustive closest hit on the device (every ray against every triangle, same triangle test):
 * the full-size checker for the BVH path. */
int ls_debug_trace_bruteforce(ls_tracer *tr, float *t, uint32_t *gid);

/* Transformed (sensor-frame) vertices and rebased indices of the committed scene. */
int ls_debug_scene_size(ls_tracer *tr, uint32_t *n_verts, uint32_t *n_tris, uint32_t *n_node_slots,
                        uint32_t *leaf_size);
int ls_debug_download_scene(ls_tracer *tr, float *verts_xyz, uint32_t *tri_idx);

/* BVH arrays: n_node_slots BVH2 nodes (64 B each) and n_tris triangle records (48 B each):
 *   node i: float4 q[4] = (L.lo.xyz, bits(left ref)), (L.hi.xyz, bits(right ref)), (R.lo.xyz, 0), (R.hi.xyz, 0)
 *     child ref: bit 31 set = leaf k (records [k*leaf_size, k*leaf_size+leaf_size) clipped to n_tris),
 *     else index of another node; node 0 is the root (a one-leaf scene has no node at all)
 *   triangle record: float v0[3]; uint32 gid; float e1[3]; float NgC; float e2[3]; uint32 pad */
int ls_debug_download_bvh(ls_tracer *tr, void *nodes, void *tri_records);

/* The build path's radix sort on its own (ls_sort.hip): sorts n (key, value) pairs by their 30-bit keys in place (host
 * arrays; stable: equal keys keep their input order). */
int ls_debug_sort_pairs(ls_tracer *tr, uint32_t *keys, uint32_t *vals, uint32_t n);

/* The host half of ls_trace_scene_expand on its own (no device, no handle): n 8-byte (ray, t) records in ascending ray
 * order -> n 32-byte points, from factor tables sin_theta[V], cos_theta[V] and interleaved (cos_phi, sin_phi)[H]. */
int ls_debug_expand_hits(void *dst_points32, const void *hits8, uint32_t n, const float *sin_theta, const float *cos_theta,
                         const float *cs_phi, uint32_t V, uint32_t H);

#ifdef __cplusplus
}
#endif
#endif /* LIDARSHOOTER_HIP_DEBUG_H */
