/*
 * lidarshooter_hip.h -- C ABI of the MI355X (gfx950) LiDAR ray-casting backend.
 *
 * Drop-in boundary: the reference's tracer plugin surface `lidarshooter::ITracer`
 * (ros_ws/src/lidarshooter/src/ITracer.hpp:29-152) as implemented by EmbreeTracer (CPU) and
 * OptixTracer (CUDA).  The reference has no FFI for this path -- its backends are compiled in --
 * so the entry points below are exactly what a `HipTracer : public ITracer` adapter binds, one
 * call per virtual (INTEGRATION.md shows that adapter).  Plain pointers and sizes only; no
 * exceptions cross this boundary; every mutator returns 0 / a non-negative id on success and a
 * negative ls_status on failure, and ls_last_error() holds the message.
 *
 * Threading (ITracer callers: ROS spinner thread for update/commit/trace, Qt thread for
 * add/remove, unsynchronised -- mainwindow.cpp:153,320 vs MeshProjector.cpp:446-464): every call
 * takes the handle's internal mutex.  Buffers returned by ls_trace_scene stay valid until the
 * next call on the same handle.
 *
 * All paths in comments are relative to /root/reference/ros_ws/src/lidarshooter/src/.
 */
#ifndef LIDARSHOOTER_HIP_H
#define LIDARSHOOTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): + ls_tracer_set_hit_buffers, ls_group_frame_status, LS_INFO_EMIT_POINTS, LS_INFO_FRAME_GRAPH_PATCH_WAITS,
 *    ls_source_hash, LS_OPT_BVH_WIDE / LS_INFO_BVH_WIDE, LS_GROUP_INFO_ARRANGEMENT_*; ls_tracer_set_sensor* is refused while external output buffers are
 *    installed; ls_commit_scene can return LS_ERR_OUT_OF_RANGE; the test hooks left this header for lidarshooter_hip_debug.h.
 *    The bindings (capi.py, groupapi.py, integration/HipTracer.hpp) check it against ls_abi_version() at load. */
#define LS_ABI_VERSION 4

typedef struct ls_tracer ls_tracer;

typedef enum ls_status {
    LS_OK = 0,
    LS_ERR_INVALID_ARGUMENT = -2,
    LS_ERR_UNKNOWN_GEOMETRY = -3, /* name not registered (EmbreeTracer.cpp:224-225 returns -1 for remove) */
    LS_ERR_DUPLICATE_GEOMETRY = -4,
    LS_ERR_UNSUPPORTED_TYPE = -5, /* neither triangles nor quads (EmbreeTracer.cpp:200-201 returns 0/false there) */
    LS_ERR_HIP = -6,              /* a HIP runtime call failed; message in ls_last_error */
    LS_ERR_NO_DEVICE = -7,        /* no usable gfx950 device: the product never falls back to the CPU */
    LS_ERR_NOT_COMMITTED = -8,
    LS_ERR_OUT_OF_RANGE = -9
    /* -1 is reserved for the reference's own "empty scene" / "unknown name" return value */
} ls_status;

/* Geometry element types; values follow Embree's RTCGeometryType so the adapter can pass the
 * enum it receives through ITracer::addGeometry (ITracer.hpp:50) unchanged. */
#define LS_GEOMETRY_TYPE_TRIANGLE 0 /* RTC_GEOMETRY_TYPE_TRIANGLE */
#define LS_GEOMETRY_TYPE_QUAD 1     /* RTC_GEOMETRY_TYPE_QUAD (EmbreeTracer.cpp:179-198): 4 indices per element; traced as
                                     * Embree does, as the triangle pair (v0,v1,v3), (v2,v3,v1); ls_hit.prim = quad index */

/* Sensor description = what LidarDevice holds after loadConfiguration()
 * (LidarDevice.cpp:482-633, :758-822).  The adapter fills it from its LidarDevice; the repo's own
 * host mirror (lidarshooter_amd/host/LidarDevice.hpp) fills it from the same JSON files. */
typedef struct ls_sensor_desc {
    const float *vertical_deg; /* channels.vertical[]: degrees above the horizon, channel order   */
    uint32_t n_vertical;       /* V                                                                */
    float h_begin;             /* channels.horizontal.range.begin (degrees)                        */
    float h_end;               /* channels.horizontal.range.end                                    */
    uint32_t h_count;          /* channels.horizontal.count = H; step = (end-begin)/(H-1) (:611)   */
    float Rinv[9];             /* sensorToBase.Rinv, row-major (:813)                              */
    float t[3];                /* (baseToOrigin.tx, baseToOrigin.ty, sensorToBase.tz) (:383-388)   */
} ls_sensor_desc;

/* The same sensor as ray-direction factor tables, for a caller that holds a live LidarDevice but not its
 * private fields (the ROS-typed adapter integration/HipTracer.hpp recovers these through the public
 * API only: LidarDevice::nextRay1 and originToSensor, LidarDevice.hpp:180,252).  Ray (v, h) has the
 * direction (sin_theta[v]*cos_phi[h], sin_theta[v]*sin_phi[h], cos_theta[v]) -- the products the
 * reference forms (LidarDevice.cpp:310-316); the kernels multiply exactly these entries.
 * The footprint bounds of the projection engine need the ANGLES too.  Elevations: the library derives every
 * channel's from the tables themselves, atan2(cos_theta, sin_theta) in double (the bounds' elevation slack is
 * 2e-4 degrees: no description could be trusted to that); elevation_deg is a cross-check -- tables whose
 * elevation_deg is more than 0.01 degrees off that are refused (LS_ERR_INVALID_ARGUMENT), never traced wrongly.
 * Azimuths: h_begin_deg + h * h_step_deg feeds bounds with 0.005 degrees + 1/16 column of slack and must be within
 * 0.002 degrees of atan2(sin_phi, cos_phi) for every column, or the tables are refused as well. */
typedef struct ls_sensor_tables {
    const float *sin_theta;     /* [n_vertical]                                              */
    const float *cos_theta;     /* [n_vertical]                                              */
    const float *elevation_deg; /* [n_vertical] degrees above the horizon (90 - theta)       */
    uint32_t n_vertical;
    const float *sin_phi;       /* [h_count]                                                 */
    const float *cos_phi;       /* [h_count]                                                 */
    uint32_t h_count;
    float h_begin_deg;          /* azimuth of column 0                                       */
    float h_step_deg;           /* azimuth of column h = h_begin_deg + h_step_deg * h        */
    float Rinv[9];              /* as in ls_sensor_desc                                      */
    float t[3];
} ls_sensor_tables;

/* One hit, 16 bytes.  Points and hit records are emitted in ray-index order (r = v*H + h,
 * LidarDevice.cpp:824-845), one record per point. */
typedef struct ls_hit {
    uint32_t ray;   /* global ray index r; channel (ring) = r / H, azimuth column = r % H */
    uint32_t geom;  /* geomID as returned by ls_add_geometry                               */
    uint32_t prim;  /* triangle index within that geometry (Embree primID)                 */
    float t;        /* hit distance along the unit ray direction                           */
} ls_hit;

/* Result of one traceScene().  points32 is the PointCloud2 `data` payload: n_points records of
 * 32 bytes, x f32@0 y f32@4 z f32@8 0@12 intensity f32@16 ring i32@20 0@24..31
 * (XYZIRBytes.cpp:24-40); the adapter sets cloud.width = n_points (EmbreeTracer.cpp:364). */
typedef struct ls_frame {
    const uint8_t *points32; /* pinned host memory, 32*n_points bytes              */
    const ls_hit *hits;      /* pinned host memory, n_points records               */
    uint32_t n_points;
    uint32_t n_rays;         /* rays traced by this handle's shard (V * shard width) */
    uint32_t frame;          /* echo of the frame index (PointCloud2 header.seq, LidarDevice.cpp:108) */
    const void *d_points32;  /* the same data in device memory (for RCCL gathers)  */
    const void *d_hits;
    const uint32_t *d_n_points; /* device word holding n_points                    */
    const void *compact16;   /* LS_OPT_HOST_OUTPUT = 2: n_points records of 16 bytes (x, y, z f32, ring i32) in pinned
                              * host memory instead of points32 (which is NULL then); ls_expand_points rebuilds the
                              * 32-byte records -- half the bytes cross PCIe                                        */
} ls_frame;

/* ---- lifetime: EmbreeTracer::create / ~EmbreeTracer (EmbreeTracer.cpp:10-70),
 *      OptixTracer::create (OptixTracer.hpp:91) --------------------------------------------- */
int ls_tracer_create(const ls_sensor_desc *sensor, int hip_device, ls_tracer **out);
int ls_tracer_create_tables(const ls_sensor_tables *sensor, int hip_device, ls_tracer **out);
/* ITracer::setSensorConfig (ITracer.hpp:129, ITracer.cpp:48) / a LidarDevice initialised again (LidarDevice.hpp:116-117):
 * EmbreeTracer::traceScene reads its LidarDevice every frame (EmbreeTracer.cpp:299-307), so another sensor -- raster,
 * channel table, pose -- takes effect at the next trace.  These calls give the handle another sensor and keep its
 * geometries: frames in flight complete first, the shard goes back to the full turn, a committed scene is committed
 * again for the new sensor.  Forms as the two create calls. */
int ls_tracer_set_sensor(ls_tracer *tr, const ls_sensor_desc *sensor);
int ls_tracer_set_sensor_tables(ls_tracer *tr, const ls_sensor_tables *sensor);
void ls_tracer_destroy(ls_tracer *tr);

/* ---- ITracer::addGeometry (ITracer.hpp:50; EmbreeTracer.cpp:115-218; OptixTracer.cpp:63-133).
 * Returns the geomID (>= 0, lowest free id like rtcAttachGeometry) or a negative ls_status. */
int ls_add_geometry(ls_tracer *tr, const char *name, int geometry_type, int n_vertices, int n_elements);

/* ---- ITracer::removeGeometry (ITracer.hpp:59; EmbreeTracer.cpp:220-260).
 * Returns the removed geomID, or -1 if the name is unknown (Embree behaviour). */
int ls_remove_geometry(ls_tracer *tr, const char *name);

/* ---- ITracer::updateGeometry(name, Eigen::Affine3f, mesh) (ITracer.hpp:69; EmbreeTracer.cpp:262-274;
 * MeshTransformer.cpp:142-205).  affine3x4: row-major [linear | translation].  verts: n_vertices
 * records of vert_stride bytes whose first 12 bytes are x,y,z float32 (pcl cloud.data with
 * point_step, MeshTransformer.cpp:176-181).  tri_idx: 3*n_elements vertex indices (polygons[i].vertices,
 * MeshTransformer.cpp:512-518; 4*n_elements for a quad geometry, :521-552); NULL keeps the indices of the previous update.  Host pointers; the
 * caller may reuse them as soon as the call returns (MeshProjector.cpp:448-461 does).  By default the copy
 * goes straight from the caller's pageable memory at PCIe rate and the call returns when the memory has been
 * read; with LS_OPT_UPLOAD_MODE = 0 the library's worker threads stage it through pinned memory instead and the
 * call never waits for the device. */
int ls_update_geometry(ls_tracer *tr, const char *name, const float affine3x4[12], const void *verts,
                       uint32_t vert_stride, const uint32_t *tri_idx);

/* ---- ITracer::updateGeometry(name, translation, rotation, mesh) (ITracer.hpp:80;
 * EmbreeTracer.cpp:276-288): T = Translation(lin)*Rz(ang.z)*Ry(ang.y)*Rx(ang.x)
 * (MeshTransformer.cpp:467-477). */
int ls_update_geometry_components(ls_tracer *tr, const char *name, const float lin[3], const float ang[3],
                                  const void *verts, uint32_t vert_stride, const uint32_t *tri_idx);

/* Same as ls_update_geometry with verts / tri_idx already resident in HBM on the tracer's device
 * (no PCIe in the frame loop; used by bench.py).  The copies are stream-ordered. */
int ls_update_geometry_device(ls_tracer *tr, const char *name, const float affine3x4[12], const void *d_verts,
                              uint32_t vert_stride, const uint32_t *d_tri_idx);

/* Zero-copy variant: the library keeps the two device pointers and READS THE CALLER'S BUFFERS IN PLACE
 * during the following commitScene / traceScene calls (the vertex transform is fused into the trace
 * kernel, so nothing needs to be staged).  The caller must leave the buffers unchanged and alive until
 * that traceScene has completed on the handle's stream, or until the next update of this geometry.
 * d_tri_idx = NULL keeps the previously shared / copied indices. */
int ls_update_geometry_device_shared(ls_tracer *tr, const char *name, const float affine3x4[12], const void *d_verts,
                                     uint32_t vert_stride, const uint32_t *d_tri_idx);

/* Only the rigid transform of an already uploaded mesh changes (AffineMesh pose integration,
 * AffineMesh.cpp:108-128): no vertex traffic at all. */
int ls_update_geometry_transform(ls_tracer *tr, const char *name, const float affine3x4[12]);

/* T = Translation(lin) * Rz(ang.z) * Ry(ang.y) * Rx(ang.x) as the row-major 3x4 matrix the calls above take
 * (MeshTransformer.cpp:467-477, Eigen's AngleAxis arithmetic) -- lets the adapter turn the (translation,
 * rotation) overload of ITracer::updateGeometry into ls_update_geometry_transform when the mesh is unchanged. */
void ls_affine_from_components(const float lin[3], const float ang[3], float affine3x4[12]);

/* ---- ITracer::commitScene (ITracer.hpp:87; EmbreeTracer.cpp:290-295 rtcCommitScene;
 * OptixTracer.cpp:263-275, :517-571).  Fixes the geometry layout (global triangle ids in (geomID,
 * primID) order); with the BVH engine it also transforms every geometry into the sensor frame and
 * builds the BVH on the device (the projection engine has nothing to build).  Returns 0, or -1 on an
 * empty scene (OptixTracer.cpp:266-267).  A geometry whose index buffer changed since the last commit is
 * checked here (one max-reduction over the indices on the device, read back): LS_ERR_OUT_OF_RANGE when a
 * triangle names a vertex the geometry does not have -- the message (ls_last_error) names geometry and
 * index; nothing is built, ls_trace_scene is refused until a commit succeeds, and new indices for that
 * geometry (or ls_remove_geometry) clear the condition.  Non-finite or huge
 * vertex COORDINATES are not an error: such triangles are never hit (as in the reference's ray test). */
int ls_commit_scene(ls_tracer *tr);

/* ---- ITracer::traceScene (ITracer.hpp:94; EmbreeTracer.cpp:297-367; OptixTracer.cpp:277-358).
 * Ray generation + closest hit + packing.  Returns 0, or -1 on an empty scene with a zero-point
 * frame (OptixTracer.cpp:280-288). */
int ls_trace_scene(ls_tracer *tr, uint32_t frame_index, ls_frame *out);

/* As ls_trace_scene but leaves the results on the device (out->points32 / hits are NULL,
 * n_points is not read back): nothing in it blocks the host.  The results are ordered on the handle's
 * stream when the call returns -- unless LS_OPT_PIPELINE keeps several frames in flight (see there). */
int ls_trace_scene_async(ls_tracer *tr, uint32_t frame_index, ls_frame *out);

/* traceScene for a caller whose cloud lives in ordinary host memory (PointCloud2::data, EmbreeTracer.cpp:297-367), in two
 * steps that overlap the device, the PCIe link and the host:
 *   ls_trace_scene_begin   launches the frame and returns as soon as the HIT COUNT is known -- the pack pass is still
 *                          running -- so that the caller can size its cloud (data.resize(32 * n_points)) meanwhile;
 *   ls_trace_scene_expand  writes the n_points 32-byte records (XYZIRBytes.cpp:24-40) to dst_points32 with the library's
 *                          worker threads: the first half of the cloud while the second half is still crossing PCIe as
 *                          8-byte (ray, t) records, from which the host rebuilds xyz = t * direction with the factor
 *                          tables and the operation order of the device (the same bits).  Returns when the cloud is
 *                          complete.
 * Same results as ls_trace_scene + ls_expand_points; -1 / zero points on an empty scene.  One frame in flight. */
int ls_trace_scene_begin(ls_tracer *tr, uint32_t frame_index, uint32_t *n_points);
int ls_trace_scene_expand(ls_tracer *tr, void *dst_points32);

/* ---- ITracer::getGeometryCount (ITracer.hpp:101) */
long ls_geometry_count(ls_tracer *tr);

/* Per-name getters of EmbreeTracer (EmbreeTracer.cpp:82-113, :369-439); LS_ERR_UNKNOWN_GEOMETRY if the name is not
 * registered (the reference returns -1 from getGeometryId and throws TraceException codes 1 / 4 / 8 from
 * getVertexCount / getElementCount / getGeometryType: the adapter maps the status to exactly those). */
int ls_geometry_id(ls_tracer *tr, const char *name);
long ls_vertex_count(ls_tracer *tr, const char *name);
long ls_element_count(ls_tracer *tr, const char *name);
/* EmbreeTracer::getGeometryType (EmbreeTracer.cpp:103-113; test/EmbreeTracer_test.cpp:116-120): the LS_GEOMETRY_TYPE_*
 * the geometry was added with (= the RTCGeometryType value). */
int ls_geometry_type(ls_tracer *tr, const char *name);

/* LidarDevice::getTotalRays / getTotalChannels (LidarDevice.cpp:411-419) for this handle. */
uint32_t ls_total_rays(ls_tracer *tr);
uint32_t ls_total_channels(ls_tracer *tr);

const char *ls_last_error(ls_tracer *tr);
int ls_abi_version(void);
/* Sixteen hex digits: SHA-256 over the library's sources (every .hip / .h / .cpp file of csrc: file name then contents, in sorted order)
 * as they were when THIS binary was built.  bench.py prints it next to the hash of the sources it finds, so that a stale
 * .so cannot speak for newer sources (the library is built in-tree and travels to the GPU box as a binary). */
const char *ls_source_hash(void);

/* ---- multi-GPU: restrict this handle to the azimuth columns [first_az, first_az + n_az) of every
 * channel.  Ray indices in ls_hit stay global.  Default: the full revolution. */
int ls_tracer_set_shard(ls_tracer *tr, uint32_t first_az, uint32_t n_az);

/* Multi-GPU collection (the all-gatherv of hit records).  Each rank points its outputs
 * (ls_tracer_set_output_buffers) into a slot of 64 + 16*capacity bytes: [n_points u32 | pad to 64 B |
 * capacity ls_hit records] -- the 32-byte points need not travel, they are a function of (ray, t).
 * After one all-gather of the slots (RCCL, in rank order), this call compacts the `world` slots at
 * d_gathered into one contiguous cloud on the device: d_hits (all records, ascending azimuth
 * sector), d_points32 (rebuilt points), *d_n_points.  Stream-ordered on the handle's stream. */
int ls_expand_gathered_hits(ls_tracer *tr, const void *d_gathered, uint32_t world, uint32_t capacity, void *d_points32,
                            void *d_hits, uint32_t *d_n_points);
/* The same on a stream of the caller's (a hipStream_t): the collective's stream, so that the compaction follows the
 * gather without holding up the tracer's stream (include/lidarshooter_group.h does this). */
int ls_expand_gathered_hits_on(ls_tracer *tr, void *hip_stream, const void *d_gathered, uint32_t world, uint32_t capacity,
                               void *d_points32, void *d_hits, uint32_t *d_n_points);
/* The sized gather's form (include/lidarshooter_group.h, LS_GROUP_FLAG_SIZED_GATHER): only the FRONT of every slot travelled --
 * the header with the rank's true count and the first gathered_capacity records; the `world` pieces lie 64 + 16 *
 * gathered_capacity bytes apart.  The cloud is rebuilt from the records that are there, and 64 bytes of pinned host memory
 * (host_stat64; may be NULL) receive { u32 largest true count of any rank, u32 truncated (some rank had more hits than
 * travelled), u32 epoch (released last) }. */
int ls_expand_gathered_hits_sized(ls_tracer *tr, void *hip_stream, const void *d_gathered, uint32_t world, uint32_t gathered_capacity,
                                  void *d_points32, void *d_hits, uint32_t *d_n_points, void *host_stat64, uint32_t epoch);

/* The step after the tracer: the sensor-frame cloud into the world frame, on the
 * device.  Replaces CloudTransformer::applyInverseTransform (CloudTransformer.cpp:283-318) +
 * LidarDevice::originToSensorInverse (LidarDevice.cpp:393-401): p_world = R * (T * p) + t for the
 * x,y,z of every 32-byte point; T = affine3x4 (row-major, NULL = identity), R = the sensor's
 * sensorToBase rotation (row-major; the handle only holds its inverse), t = the handle's translation.
 * d_points32_in holds *d_n_points records (as ls_trace_scene_async leaves them); they are written to
 * d_points32_out starting at record *d_out_base (NULL = 0) -- in == out with base 0 transforms in
 * place -- and *d_out_total (NULL = not wanted) receives *d_out_base + *d_n_points, so the clouds of
 * several sensors merge into one buffer by chaining total -> base.  Records that would not fit
 * out_capacity are dropped.  Stream-ordered on the handle's stream. */
int ls_cloud_to_world(ls_tracer *tr, const float *affine3x4, const float *R, const void *d_points32_in,
                      const uint32_t *d_n_points, void *d_points32_out, const uint32_t *d_out_base,
                      uint32_t *d_out_total, uint32_t out_capacity);

/* Run all device work of this handle on `hip_stream` (a hipStream_t; NULL = the handle's own
 * stream).  Lets the caller order its collectives after ls_trace_scene_async. */
int ls_tracer_set_stream(ls_tracer *tr, void *hip_stream);
int ls_tracer_synchronize(ls_tracer *tr);
/* LS_OPT_PIPELINE: order the handle's stream after every frame still in flight (no host wait). */
int ls_tracer_flush(ls_tracer *tr);
/* Finer than a flush, for a consumer that works on frames one by one while later frames are already in flight (the
 * collective stream of include/lidarshooter_group.h):
 *   ls_tracer_order_after_last_frame  orders hip_stream (a hipStream_t) after the frame issued last -- its points, hit
 *       records and count are complete for work enqueued on hip_stream afterwards -- and after nothing else: with
 *       three frames in flight the two other frames keep running.
 *   ls_tracer_wait_event  makes everything the handle does from now on (its next frames included, whatever stream
 *       carries them) start after hip_event (a hipEvent_t) has completed -- e.g. the event behind the last reader of an
 *       output buffer that the next frame writes again.
 *   ls_tracer_next_frame_waits  the narrow form of the same: only the frame issued NEXT starts after hip_event (with three
 *       frames in flight that is one wait on that frame's stream instead of three runtime calls; the frames after it are
 *       not ordered behind the event unless they are behind that frame anyway). */
int ls_tracer_order_after_last_frame(ls_tracer *tr, void *hip_stream);
int ls_tracer_wait_event(ls_tracer *tr, void *hip_event);
int ls_tracer_next_frame_waits(ls_tracer *tr, void *hip_event);

/* Write packed points / hit records into caller-owned device buffers (capacity in records,
 * >= ls_total_rays of the shard) instead of the handle's own; NULL restores the default. */
int ls_tracer_set_output_buffers(ls_tracer *tr, void *d_points32, void *d_hits, uint32_t *d_n_points,
                                 uint32_t capacity);
/* The same for hit records alone (LS_OPT_EMIT_POINTS = 0: a sharded group rebuilds the points of the whole frame from the
 * gathered records, include/lidarshooter_group.h): no point buffer exists, so while these buffers are installed
 * LS_OPT_EMIT_POINTS = 1 is refused, and a trace with it set fails with LS_ERR_INVALID_ARGUMENT instead of writing 32 bytes
 * per hit through a pointer that was never meant for them.  NULL d_hits restores the default. */
int ls_tracer_set_hit_buffers(ls_tracer *tr, void *d_hits, uint32_t *d_n_points, uint32_t capacity);

/* ---- options */
#define LS_OPT_LEAF_SIZE 1      /* triangles per BVH leaf (1,2,4,8), default 1; takes effect at next commit  */
#define LS_OPT_TIMING 2         /* 1: bracket every stage with hipEvents, 2: only the trace kernel        */
#define LS_OPT_COUNT_VISITS 3   /* 1: trace kernel also counts node fetches / triangle tests              */
#define LS_OPT_PIPELINE 6       /* frames in flight (projection engine, asynchronous API).  0 (default): one.
                                 * 1: two, on the handle's stream -- the finish + pack workgroups of a frame ride in the
                                 *    launch of the next frame's k_project; a frame's outputs are ordered on the
                                 *    handle's stream after the NEXT ls_trace_scene_async.
                                 * 2: three -- whole frames rotate over three streams of the library; outputs are ordered
                                 *    on the handle's stream by ls_tracer_flush (no event or wait per frame).
                                 * In both modes ls_tracer_flush / ls_tracer_synchronize complete everything, the
                                 * library's own output buffers rotate with the frames (a caller that sets output
                                 * buffers rotates them itself), and meshes handed over with
                                 * ls_update_geometry_device_shared must stay unchanged while frames are in flight.  */
#define LS_OPT_HOST_OUTPUT 7    /* synchronous ls_trace_scene: 1 (default) the pack kernel writes points (and hit
                                 *    records) straight into the pinned host buffers that ls_frame returns -- one
                                 *    host wait per frame, no copy engine; 2: the same with 16-byte compact point
                                 *    records (ls_frame.compact16, expanded by ls_expand_points); 0: device buffers,
                                 *    then count + sized D2H copies (two waits).                               */
#define LS_OPT_READBACK_HITS 8  /* synchronous ls_trace_scene: 1 (default) ls_frame.hits is filled; 0: the 16-byte
                                 *    hit records stay on the device (ls_frame.hits = NULL, d_hits valid) -- the ITracer
                                 *    adapter only needs the 32-byte points.                                   */
/*      option 9 is a test hook (lidarshooter_hip_debug.h: LS_OPT_DEBUG_FAULT); not part of this surface             */
#define LS_OPT_BVH_REFIT 11     /* BVH engine: 1 (default) a commit after which only vertices / poses differ refits the
                                 *    hierarchy (OptixTracer.cpp:532-535 OPERATION_UPDATE): no key pass, no sort; 0: always
                                 *    a full build.  Identical results.                                                   */
#define LS_OPT_BVH_INSTANCED 12 /* BVH engine: 1 (default) one hierarchy per geometry, built once in MESH space; every frame
                                 *    carries the rays into each geometry's mesh space and tests a leaf's triangles exactly as
                                 *    the other paths do (corners through the frame's transform, same bits) -- a commit after
                                 *    which only poses differ, the sensor's or a mesh's, builds and refits nothing; new vertices refit that geometry
                                 *    alone (LS_OPT_BVH_REFIT), new indices rebuild it.  Scenes of
                                 *    more than 16 geometries or with a (nearly) singular mesh matrix take the classic path
                                 *    (0: always): one hierarchy in the sensor frame, refitted per LS_OPT_BVH_REFIT.
                                 *    Identical results.  Takes effect at the next commit.                                */
#define LS_OPT_BLOCK_CULL 10    /* projection engine, meshes of 524 288 triangles or more: keep the mesh in Morton order with a
                                 *    bound per 4 triangles and drop, before their indices are read, the groups that no ring
                                 *    of the raster and no column of the shard can meet.  0 off, 1 on, 2 (default) auto: on for
                                 *    geometries of 2 000 000 triangles or more (where the kernel is bandwidth-bound), and
                                 *    from 524 288 triangles when the handle is an azimuth shard narrower than the raster
                                 *    (ls_tracer_set_shard: most groups then lie outside the sector).  Identical results.  */
#define LS_OPT_UPLOAD_MODE 13   /* ls_update_geometry from host memory: 1 (default) one copy straight from the caller's pageable
                                 *    memory at PCIe rate, the call returns when the memory has been read; 0: worker threads
                                 *    stage it through pinned memory chunk by chunk, the call never waits for the device
                                 *    (measured slower: 0.26-0.31 ms against 0.15 ms for 8 MB); 2: one thread, one copy.  */
#define LS_OPT_ENGINE 5         /* closest-hit engine: 0 auto (default), 1 BVH traversal, 2 sensor-space
                                 *    projection (streams triangles over the ray raster); identical results.
                                 *    Takes effect at the next commit.                                      */
#define LS_OPT_BVH_WIDE 16      /* BVH engine, instanced mode: 1 (default) the trace walks FOUR-wide nodes made of the binary hierarchy (a
                                 *    node's slots are its grandchildren: half the trips per ray, four slab tests in flight per trip;
                                 *    + 128 bytes per node) -- made at the commit for hierarchies of up to 65 536 leaves, and for bigger
                                 *    ones by the second frame that finds them unchanged (34 us per million nodes: a hierarchy rebuilt
                                 *    or refitted every frame never pays it and is walked through its binary nodes); 0: the binary
                                 *    nodes always.  Identical results.  Takes effect at the next commit.                            */
#define LS_OPT_EMIT_POINTS 15   /* ls_trace_scene_async: 1 (default) the pack pass writes the 32-byte points and the 16-byte hit
                                 *    records; 0: the hit records only (d_points32 stays untouched) -- for a consumer that
                                 *    rebuilds points from (ray, t) anyway: a sharded group gathers hit records and
                                 *    ls_expand_gathered_hits rebuilds every rank's points from them (8 MB of writes per
                                 *    frame less at 524 288 rays).  The synchronous ls_trace_scene always delivers points.  */
#define LS_OPT_FRAME_GRAPH 14   /* LS_OPT_PIPELINE = 2 (three frames in flight, projection engine): 1 = the launches of a frame
                                 *    are captured ONCE per stream of the rotation into a HIP graph and every later frame
                                 *    on that stream is one hipGraphLaunch; the kernel nodes whose arguments changed since
                                 *    (a pose, an output buffer, the shard) are patched first
                                 *    (hipGraphExecKernelNodeSetParams); a change of the launch sequence itself (another
                                 *    set of geometries, culling switched) captures anew.  0 (default): plain launches.
                                 *    Identical results; host cost per frame 9 -> 6 us for the three launches of a frame,
                                 *    26 -> 6-8 us for a sharded group frame (tools/micro/graph_launch.hip).  A runtime
                                 *    that refuses the capture leaves the handle on plain launches
                                 *    (LS_INFO_FRAME_GRAPH_STATE).                                                        */
int ls_tracer_set_option(ls_tracer *tr, int option, int value);

/* A frame graph that also holds work of the CALLER's (include/lidarshooter_group.h: the frame's collective and the
 * rebuild of the gathered cloud), so that the whole frame is one graph launch.  Needs LS_OPT_PIPELINE = 2 and
 * LS_OPT_FRAME_GRAPH = 1.
 *   ls_frame_graph_begin   the next ls_trace_scene_async leaves its frame graph open.  `tag` names the caller's part: a
 *                          graph captured under another tag is captured anew.
 *   ls_frame_graph_stream  after that ls_trace_scene_async: the stream (a hipStream_t) the frame is on, which of the three
 *                          streams of the rotation it is (0..2: one graph, hence one set of caller buffers, per slot;
 *                          LS_FRAME_NO_SLOT for a frame that ran on the handle's own stream: timing / counting frames) and
 *                          how it is being built: LS_FRAME_EAGER (plain launches: the caller enqueues its own work on the
 *                          stream as usual), LS_FRAME_CAPTURING (the stream is capturing: the caller enqueues its work on
 *                          it ONCE, kernels of this library through this library's entry points, and it becomes part of the
 *                          graph) or LS_FRAME_REPLAYING (the graph exists: the caller enqueues NOTHING except through
 *                          this library's entry points, which only compare their arguments with the captured ones).
 *   ls_frame_graph_end     closes the graph and launches it.  Returns 0, or 1 = nothing was launched, the graph was
 *                          discarded (the launch sequence changed, or the runtime refused the capture): issue the
 *                          frame again from ls_frame_graph_begin (at most once more).
 *   ls_frame_graph_reset   destroys the cached graphs (a caller whose captured work dies -- a communicator -- calls it). */
#define LS_FRAME_NO_SLOT 0xFFFFFFFFu
#define LS_FRAME_EAGER 0
#define LS_FRAME_CAPTURING 1
#define LS_FRAME_REPLAYING 2
int ls_frame_graph_begin(ls_tracer *tr, uint64_t tag);
int ls_frame_graph_stream(ls_tracer *tr, void **hip_stream, uint32_t *slot, int *mode);
int ls_frame_graph_end(ls_tracer *tr);
int ls_frame_graph_reset(ls_tracer *tr);

/* Rebuild n_points 32-byte PointCloud2 records (XYZIRBytes.cpp:24-40; intensity 64.0, EmbreeTracer.cpp:343) at
 * dst_points32 from the compact records of ls_frame.compact16, with the library's worker threads. */
int ls_expand_points(void *dst_points32, const void *compact16, uint32_t n_points);

/* Copy `bytes` from src to dst with the library's worker threads (both host pointers).  The adapter
 * uses it to move a frame's points from the pinned buffer of ls_frame into PointCloud2::data; a single
 * thread moves 8 MB in ~0.7 ms, the PCIe transfer of the same bytes takes 0.15 ms. */
int ls_parallel_copy(void *dst, const void *src, uint64_t bytes);

/* Facts about the handle: returns the value or a negative ls_status. */
#define LS_INFO_CONCURRENT_STREAMS 1 /* LS_OPT_PIPELINE = 2: mutually concurrent streams found by the calibration
                                      *    (3 = full three-stream mode; fewer: the handle runs mode 1 instead)    */
#define LS_INFO_PIPELINE_MODE 2      /* the frames-in-flight mode actually in use (0, 1 or 2)                    */
#define LS_INFO_DEVICE_STATUS 3      /* sticky device status word (0 = ok); read-and-clear, synchronises          */
#define LS_INFO_HOST_THREADS 4       /* worker threads of the host copy pool                                     */
#define LS_INFO_AZIMUTH_COUNT 5      /* H: azimuth columns of the sensor's full raster (whatever the shard)       */
#define LS_INFO_LAST_COMMIT_REFIT 6  /* BVH engine: 1 if the last commitScene refitted instead of rebuilding      */
#define LS_INFO_BVH_INSTANCED 7      /* BVH engine: 0 classic hierarchy; 1 instanced and the last commit built nothing;
                                      *    2 instanced and the last commit (re)built some geometry's hierarchy        */
#define LS_INFO_NEXT_SLOT 8           /* LS_OPT_PIPELINE = 2: which of the three streams (0..2) the NEXT frame runs on   */
#define LS_INFO_FRAME_GRAPH_STATE 9   /* 0 off, 1 on, 2 on but the runtime refused a capture: plain launches since       */
#define LS_INFO_FRAME_GRAPH_CAPTURES 10 /* frame graphs captured so far                                                  */
#define LS_INFO_FRAME_GRAPH_REPLAYS 11  /* frames issued as one graph launch                                              */
#define LS_INFO_FRAME_GRAPH_PATCHES 12  /* kernel nodes patched with new arguments before a replay                        */
#define LS_INFO_FRAME_GRAPH_LAST_PATCHED 13 /* bit i: launch i of the frame replayed last went out with new arguments     */
#define LS_INFO_EMIT_POINTS 14       /* the current LS_OPT_EMIT_POINTS                                            */
#define LS_INFO_BVH_WIDE 16                /* 1: the last trace walked the four-wide nodes (LS_OPT_BVH_WIDE, instanced mode, every geometry's made) */
#define LS_INFO_FRAME_GRAPH_PATCH_WAITS 15 /* patches that first had to wait for the previous launch of their graph (the host ran more than three frames ahead) */
long ls_get_info(ls_tracer *tr, int what);

/* Mean stage durations (milliseconds, hipEvents on the handle's stream) over every frame recorded
 * since the previous call; recording never synchronises, this call does.  Returns the number of
 * frames averaged (>= 0) or a negative ls_status. */
#define LS_T_TRANSFORM 0
#define LS_T_MORTON 1
#define LS_T_SORT 2
#define LS_T_LEAVES 3
#define LS_T_RANGE_TREE 4
#define LS_T_HIERARCHY 5
#define LS_T_TRACE 6      /* BVH: traversal kernel; projection: per-triangle footprint kernel */
#define LS_T_TRACE_AUX 7  /* BVH: row counts; projection: long-row kernel + resolve          */
#define LS_T_PACK 8
#define LS_T_COUNT 9
int ls_get_timings(ls_tracer *tr, float ms[LS_T_COUNT]);

/* Totals of the last trace when LS_OPT_COUNT_VISITS is on: {node fetches, triangle tests,
 * sum over waves of traversal-loop trips (a wave runs as long as its slowest lane), max trips}. */
int ls_get_visit_counts(ls_tracer *tr, uint64_t counts[4]);

/* ---- ray generation on its own: LidarDevice::allRaysGPU (LidarDeviceKernels.cu:25-126).
 * Writes the shard's rays as SoA float arrays of n = ls_total_rays() entries each, in device
 * memory owned by the caller: dir_x, dir_y, dir_z (origins are all zero, LidarDevice.cpp:320). */
int ls_generate_rays(ls_tracer *tr, float *d_dir_x, float *d_dir_y, float *d_dir_z);

/* The same kernel's two outputs in the reference's own layout (LidarDeviceKernels.cu:38-51): n = ls_total_rays() records
 * each, in ray-index order, device memory owned by the caller.
 *   d_rays:  lidarshooter::Ray, 32 bytes (Ray.hpp:16-35): origin xyz f32@0 (0, 0, 0), tmin f32@12, direction xyz f32@16,
 *            tmax f32@28.  The reference's kernel leaves tmin / tmax as it found them; here they are written as its
 *            OptiX programs use them: tmin 0, tmax 1e16 (OptixTracerModules.cu:45-46).
 *   d_hits:  lidarshooter::Hit, 24 bytes (Hit.hpp:16-29): t f32@0 = 1e16 ("no hit yet"), normal xyz f32@4 (left alone by
 *            the reference; written as 0 here), intensity f32@16 = 64.0, ring i32@20 = the channel index.
 * Either pointer may be NULL (that output is skipped). */
int ls_generate_rays_aos(ls_tracer *tr, void *d_rays, void *d_hits);

#ifdef __cplusplus
}
#endif
#endif /* LIDARSHOOTER_HIP_H */
