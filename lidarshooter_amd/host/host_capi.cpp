// host_capi.cpp -- tiny C entry points over the C++ host mirror so that tests/ and bench.py
// (Python) can drive LidarDevice / HipTracer / loadPolygonFileSTL exactly as a C++ user would.
// Exceptions are turned into negative return codes here; nothing else lives in this file.
#include <chrono>
#include <cstring>
#include <memory>
#include <string>

#include "HipTracer.hpp"
#include "Trajectory.hpp"
#include "../../include/lidarshooter_group.h"

using namespace lidarshooter;

struct lsh_device { LidarDevice::Ptr p; };
struct lsh_mesh { PolygonMesh m; };
struct lsh_tracer { HipTracer::Ptr p; };

static thread_local std::string g_err;

extern "C" {

const char* lsh_last_error(void) { return g_err.c_str(); }

lsh_device* lsh_device_create(const char* config_path, const char* sensor_uid)
{
    try {
        auto* d = new lsh_device();
        d->p = LidarDevice::create(config_path ? config_path : "", sensor_uid ? sensor_uid : "");
        return d;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void lsh_device_destroy(lsh_device* d) { delete d; }
unsigned lsh_device_total_rays(lsh_device* d) { return d->p->getTotalRays(); }
unsigned lsh_device_total_channels(lsh_device* d) { return d->p->getTotalChannels(); }
unsigned lsh_device_scan_ray_count(lsh_device* d) { return d->p->getScanRayCount(); }
const char* lsh_device_uid(lsh_device* d) { return d->p->getSensorUid().c_str(); }
void lsh_device_desc(lsh_device* d, ls_sensor_desc* out) { *out = d->p->sensorDesc(); }
void lsh_device_rotation(lsh_device* d, float* R9, float* Rinv9)
{
    std::memcpy(R9, d->p->rotation().data(), 36);
    std::memcpy(Rinv9, d->p->rotationInverse().data(), 36);
}
float lsh_device_step(lsh_device* d) { return d->p->horizontalStep(); }
void lsh_device_ray_direction(lsh_device* d, unsigned v, unsigned h, float* dir3) { d->p->rayDirection(v, h, dir3); }
void lsh_device_origin_to_sensor(lsh_device* d, float* p3, int inverse)
{
    if (inverse) d->p->originToSensorInverse(p3);
    else d->p->originToSensor(p3);
}
// header fields of initMessage: out[0..5] = seq,height,width,point_step,row_step,n_fields; flags: bit0 bigendian, bit1 dense
void lsh_device_init_message(lsh_device* d, int frame, unsigned* out6, unsigned* flags, char* frame_id, unsigned frame_id_cap)
{
    PointCloud2 m;
    d->p->initMessage(m, frame);
    out6[0] = m.header.seq; out6[1] = m.height; out6[2] = m.width; out6[3] = m.point_step; out6[4] = m.row_step;
    out6[5] = static_cast<unsigned>(m.fields.size());
    *flags = (m.is_bigendian ? 1u : 0u) | (m.is_dense ? 2u : 0u);
    if (frame_id && frame_id_cap) {
        std::strncpy(frame_id, m.header.frame_id.c_str(), frame_id_cap - 1);
        frame_id[frame_id_cap - 1] = 0;
    }
}

// field i of the message header initMessage builds (name copied into `name`, at most name_cap - 1 characters)
int lsh_device_message_field(lsh_device* d, unsigned i, char* name, unsigned name_cap, unsigned* offset, unsigned* datatype, unsigned* count)
{
    PointCloud2 m;
    d->p->initMessage(m, 0);
    if (i >= m.fields.size()) return -1;
    if (name && name_cap) {
        std::strncpy(name, m.fields[i].name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    *offset = m.fields[i].offset;
    *datatype = m.fields[i].datatype;
    *count = m.fields[i].count;
    return 0;
}

lsh_mesh* lsh_mesh_load_stl(const char* path)
{
    auto* m = new lsh_mesh();
    if (loadPolygonFileSTL(path, m->m) <= 0) { delete m; g_err = "cannot read STL"; return nullptr; }
    return m;
}
void lsh_mesh_destroy(lsh_mesh* m) { delete m; }
unsigned lsh_mesh_num_points(lsh_mesh* m) { return m->m.cloud.width * m->m.cloud.height; }
unsigned lsh_mesh_num_polygons(lsh_mesh* m) { return static_cast<unsigned>(m->m.polygons.size()); }
unsigned lsh_mesh_point_step(lsh_mesh* m) { return m->m.cloud.point_step; }
const void* lsh_mesh_point_data(lsh_mesh* m) { return m->m.cloud.data.data(); }
void lsh_mesh_copy_polygons(lsh_mesh* m, unsigned* out)
{
    size_t k = 0;
    for (const auto& p : m->m.polygons) for (auto v : p.vertices) out[k++] = v;
}

lsh_tracer* lsh_tracer_create(lsh_device* d, int hip_device)
{
    try {
        auto* t = new lsh_tracer();
        t->p = HipTracer::create(d->p, nullptr, hip_device);
        return t;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void lsh_tracer_destroy(lsh_tracer* t) { delete t; }
int lsh_tracer_add_geometry(lsh_tracer* t, const char* name, int type, int nv, int ne)
{
    return t->p->addGeometry(name, static_cast<RTCGeometryType>(type), nv, ne);
}
int lsh_tracer_remove_geometry(lsh_tracer* t, const char* name) { return t->p->removeGeometry(name); }
int lsh_tracer_update_geometry(lsh_tracer* t, const char* name, const float* affine12, lsh_mesh* m)
{
    try {
        Affine3f A;
        std::memcpy(A.data(), affine12, 48);
        return t->p->updateGeometry(name, A, m->m);
    } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
int lsh_tracer_update_geometry_components(lsh_tracer* t, const char* name, const float* lin3, const float* ang3, lsh_mesh* m)
{
    try {
        return t->p->updateGeometry(name, Vector3f{{lin3[0], lin3[1], lin3[2]}}, Vector3f{{ang3[0], ang3[1], ang3[2]}}, m->m);
    } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
int lsh_tracer_commit_scene(lsh_tracer* t) { return t->p->commitScene(); }
int lsh_tracer_trace_scene(lsh_tracer* t, unsigned frame)
{
    try { return t->p->traceScene(frame); } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
long lsh_tracer_geometry_count(lsh_tracer* t) { return t->p->getGeometryCount(); }
long lsh_tracer_vertex_count(lsh_tracer* t, const char* name)
{
    try { return t->p->getVertexCount(name); } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
long lsh_tracer_element_count(lsh_tracer* t, const char* name)
{
    try { return t->p->getElementCount(name); } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
int lsh_tracer_geometry_id(lsh_tracer* t, const char* name)
{
    try { return t->p->getGeometryId(name); } catch (const std::exception& e) { g_err = e.what(); return -100; }
}
// TraceException -> -1000 - its code (EmbreeTracer.cpp:103-113: 8)
int lsh_tracer_geometry_type(lsh_tracer* t, const char* name)
{
    try { return static_cast<int>(t->p->getGeometryType(name)); }
    catch (const TraceException& e) { g_err = e.what(); return -1000 - static_cast<int>(e.getErrorCode()); }
}
// cloud of the last traceScene: out4 = width,height,point_step,seq; returns pointer to cloud.data
const void* lsh_tracer_cloud(lsh_tracer* t, unsigned* out4)
{
    auto c = t->p->getTraceCloud();
    out4[0] = c->width; out4[1] = c->height; out4[2] = c->point_step; out4[3] = c->header.seq;
    return c->data.data();
}
const void* lsh_tracer_hits(lsh_tracer* t, unsigned* n)
{
    *n = t->p->getHitCount();
    return t->p->getHits();
}
void* lsh_tracer_handle(lsh_tracer* t) { return t->p->handle(); }

// The per-frame call sequence of MeshProjector::traceAffineMesh (MeshProjector.cpp:446-464) on the C ABI, `n_frames`
// times without the interpreter in between (bench.py's timed loop; lsbench.cpp is the same loop as a program):
// for every mesh ls_update_geometry_transform (the mesh is resident and unchanged, its pose is restated; mesh m takes
// affines[(frame % n_affines[m]) * 12 ...] of its own list), ls_commit_scene, the frame's output set (they rotate over
// n_out caller-owned sets), ls_trace_scene_async.  Returns 0 or the first negative status.
int lsh_stream_frames(void* tracer, const char* const* names, const float* const* affines, const unsigned* n_affines, unsigned n_meshes,
                      void* const* out_points, void* const* out_hits, void* const* out_counts, unsigned n_out, unsigned capacity,
                      unsigned first_frame, unsigned n_frames)
{
    ls_tracer* tr = static_cast<ls_tracer*>(tracer);
    ls_frame f;
    for (unsigned k = 0; k < n_frames; ++k) {
        const unsigned frame = first_frame + k;
        for (unsigned m = 0; m < n_meshes; ++m) {
            const int rc = ls_update_geometry_transform(tr, names[m], affines[m] + 12u * (n_affines[m] ? frame % n_affines[m] : 0u));
            if (rc < 0) return rc;
        }
        int rc = ls_commit_scene(tr);
        if (rc < -1) return rc;
        if (n_out) {
            const unsigned o = frame % n_out;
            rc = ls_tracer_set_output_buffers(tr, out_points[o], out_hits[o], static_cast<uint32_t*>(out_counts[o]), capacity);
            if (rc < 0) return rc;
        }
        rc = ls_trace_scene_async(tr, frame, &f);
        if (rc < -1) return rc;
    }
    return 0;
}

// lsh_stream_frames with a clock around each of its calls (tools/shard_cost.py: where a frame's host time goes):
// ns[0] pose updates, ns[1] commit, ns[2] trace -- summed over the frames.
int lsh_stream_frames_timed(void* tracer, const char* const* names, const float* const* affines, const unsigned* n_affines, unsigned n_meshes,
                            unsigned first_frame, unsigned n_frames, double* ns3)
{
    ls_tracer* tr = static_cast<ls_tracer*>(tracer);
    ls_frame f;
    auto now = [] { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    ns3[0] = ns3[1] = ns3[2] = 0.0;
    for (unsigned k = 0; k < n_frames; ++k) {
        const unsigned frame = first_frame + k;
        const double t0 = now();
        for (unsigned m = 0; m < n_meshes; ++m) {
            const int rc = ls_update_geometry_transform(tr, names[m], affines[m] + 12u * (n_affines[m] ? frame % n_affines[m] : 0u));
            if (rc < 0) return rc;
        }
        const double t1 = now();
        int rc = ls_commit_scene(tr);
        if (rc < -1) return rc;
        const double t2 = now();
        rc = ls_trace_scene_async(tr, frame, &f);
        if (rc < -1) return rc;
        const double t3 = now();
        ns3[0] += t1 - t0; ns3[1] += t2 - t1; ns3[2] += t3 - t2;
    }
    return 0;
}

// The same loop over the GPUs of a node (include/lidarshooter_group.h): every rank restates its meshes' poses, commits and
// hands the frame to its group -- azimuth shard + one all-gather of hit-record slots, or whole frames interleaved.
int lsh_group_stream_frames(void* group, void* tracer, const char* const* names, const float* const* affines, const unsigned* n_affines,
                            unsigned n_meshes, unsigned first_frame, unsigned n_frames)
{
    ls_tracer* tr = static_cast<ls_tracer*>(tracer);
    ls_group* g = static_cast<ls_group*>(group);
    for (unsigned k = 0; k < n_frames; ++k) {
        const unsigned frame = first_frame + k;
        for (unsigned m = 0; m < n_meshes; ++m) {
            const int rc = ls_update_geometry_transform(tr, names[m], affines[m] + 12u * (n_affines[m] ? frame % n_affines[m] : 0u));
            if (rc < 0) return rc;
        }
        int rc = ls_commit_scene(tr);
        if (rc < -1) return rc;
        rc = ls_group_trace(g, frame);
        if (rc < -1) return rc;
    }
    return 0;
}

// Trajectory player: writes up to `cap` poses (6 floats each: linear xyz, angular xyz); returns the count
int lsh_trajectory_play(const char* path, float period, float* out6, int cap)
{
    try {
        const auto poses = Trajectory::load(path).play(period);
        const int n = static_cast<int>(poses.size());
        for (int i = 0; i < n && i < cap; ++i) {
            for (int k = 0; k < 3; ++k) { out6[6 * i + k] = poses[i].linear[k]; out6[6 * i + 3 + k] = poses[i].angular[k]; }
        }
        return n;
    } catch (const std::exception& e) { g_err = e.what(); return -100; }
}

}  // extern "C"
