// adapter_capi.cpp -- C entry points that drive the ROS-typed adapter (integration/HipTracer.hpp) the way
// lidarshooter's MeshProjector does (MeshProjector.cpp:322-340, :446-464), compiled against integration/stubs.
// tests/test_adapter.py and bench.py's drop-in legs load the resulting libls_adapter_test.so with ctypes.
// Exceptions become negative return codes (-100) with the message in lsa_last_error().
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "HipTracer.hpp"

extern "C" {
struct lsh_mesh;
lsh_mesh* lsh_mesh_load_stl(const char* path);
void lsh_mesh_destroy(lsh_mesh* m);
unsigned lsh_mesh_num_points(lsh_mesh* m);
unsigned lsh_mesh_num_polygons(lsh_mesh* m);
unsigned lsh_mesh_point_step(lsh_mesh* m);
const void* lsh_mesh_point_data(lsh_mesh* m);
void lsh_mesh_copy_polygons(lsh_mesh* m, unsigned* out);
}

using namespace lidarshooter;

namespace {

struct TrackedMesh {   // what MeshProjector keeps per mesh: the AffineMesh's mesh + displacement
    pcl::PolygonMesh::Ptr mesh;
    Eigen::Vector3f linear, angular;
};

thread_local std::string g_error;

}  // namespace

struct lsa_ctx {
    LidarDevice::Ptr device;
    HipTracer::Ptr tracer;
    ITracer::Ptr itracer;   // every per-frame call goes through the base-class pointer, like MeshProjector::_tracer
    std::map<std::string, TrackedMesh> meshes;
    std::uint32_t frameIndex = 0;
};

// liblidarshooter_host.so (the host mirror) also defines classes named lidarshooter::LidarDevice etc.: this
// library is built with -fvisibility=hidden so the two sets never meet; only the lsa_* entry points are exported
#pragma GCC visibility push(default)
extern "C" {
const char* lsa_last_error(void);
}
#pragma GCC visibility pop

// (-200: the adapter threw BadGeometryException -- the library refused a mesh)
#define LSA_TRY(expr)                                                          \
    try { return (expr); } catch (const BadGeometryException& e) { g_error = e.what(); return -200; } catch (const std::exception& e) { g_error = e.what(); return -100; }

#pragma GCC visibility push(default)
extern "C" {

const char* lsa_last_error(void) { return g_error.c_str(); }

lsa_ctx* lsa_create(const char* sensor_config, int hip_device)
{
    try {
        auto* c = new lsa_ctx();
        c->device = LidarDevice::create(sensor_config);
        c->tracer = HipTracer::create(c->device, nullptr, nullptr, hip_device);
        c->itracer = c->tracer->getPtr();
        return c;
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

void lsa_destroy(lsa_ctx* c) { delete c; }

// the adapter's sensor probe on its own (pure host code: runs without a GPU).  tables4 = sin_theta[V], cos_theta[V],
// sin_phi[H], cos_phi[H] concatenated; elevation[V]; misc = begin, step; returns 0 or -100
int lsa_probe_sensor(const char* sensor_config, unsigned* vh2, float* tables4, unsigned tables_cap, float* elevation,
                     float* misc2, float* rinv9, float* t3)
{
    try {
        auto device = LidarDevice::create(sensor_config);
        HipTracer::SensorProbe probe(*device);
        const ls_sensor_tables t = probe.tables();
        vh2[0] = t.n_vertical;
        vh2[1] = t.h_count;
        if (tables_cap < 2 * (t.n_vertical + t.h_count)) { g_error = "tables_cap too small"; return -100; }
        std::memcpy(tables4, t.sin_theta, 4 * t.n_vertical);
        std::memcpy(tables4 + t.n_vertical, t.cos_theta, 4 * t.n_vertical);
        std::memcpy(tables4 + 2 * t.n_vertical, t.sin_phi, 4 * t.h_count);
        std::memcpy(tables4 + 2 * t.n_vertical + t.h_count, t.cos_phi, 4 * t.h_count);
        std::memcpy(elevation, t.elevation_deg, 4 * t.n_vertical);
        misc2[0] = t.h_begin_deg;
        misc2[1] = t.h_step_deg;
        std::memcpy(rinv9, t.Rinv, 36);
        std::memcpy(t3, t.t, 12);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -100;
    }
}

// a mesh as pcl::io::loadPolygonFileSTL leaves it: pcl::PointXYZ records (16 bytes: x, y, z, pad), triangles
int lsa_mesh_from_arrays_ex(lsa_ctx* c, const char* name, const float* xyz, unsigned n_vertices, const unsigned* elements,
                            unsigned n_elements, unsigned point_step, unsigned vertices_per_element);

int lsa_mesh_from_arrays(lsa_ctx* c, const char* name, const float* xyz, unsigned n_vertices, const unsigned* triangles,
                         unsigned n_triangles, unsigned point_step)
{
    return lsa_mesh_from_arrays_ex(c, name, xyz, n_vertices, triangles, n_triangles, point_step, 3);
}

// vertices_per_element 4: a quad mesh (pcl::Vertices with four indices each)
int lsa_mesh_from_arrays_ex(lsa_ctx* c, const char* name, const float* xyz, unsigned n_vertices, const unsigned* triangles,
                            unsigned n_triangles, unsigned point_step, unsigned vertices_per_element)
{
    if (point_step < 12) return -2;
    TrackedMesh tm;
    tm.mesh = pcl::PolygonMesh::Ptr(new pcl::PolygonMesh());
    auto& cloud = tm.mesh->cloud;
    cloud.width = n_vertices;
    cloud.height = 1;
    cloud.point_step = point_step;
    cloud.row_step = point_step * n_vertices;
    cloud.data.assign(static_cast<std::size_t>(n_vertices) * point_step, 0);
    for (unsigned i = 0; i < n_vertices; ++i) std::memcpy(cloud.data.data() + static_cast<std::size_t>(i) * point_step, xyz + 3 * i, 12);
    tm.mesh->polygons.resize(n_triangles);
    for (unsigned i = 0; i < n_triangles; ++i) {
        tm.mesh->polygons[i].vertices.resize(vertices_per_element);
        for (unsigned k = 0; k < vertices_per_element; ++k)
            tm.mesh->polygons[i].vertices[k] = static_cast<pcl::index_t>(triangles[vertices_per_element * i + k]);
    }
    c->meshes[name] = tm;
    return 0;
}

int lsa_mesh_from_stl(lsa_ctx* c, const char* name, const char* path)
{
    lsh_mesh* m = lsh_mesh_load_stl(path);
    if (!m) { g_error = "cannot read STL"; return -100; }
    const unsigned nv = lsh_mesh_num_points(m), nt = lsh_mesh_num_polygons(m), step = lsh_mesh_point_step(m);
    std::vector<float> xyz(3 * static_cast<std::size_t>(nv));
    const auto* src = static_cast<const unsigned char*>(lsh_mesh_point_data(m));
    for (unsigned i = 0; i < nv; ++i) std::memcpy(&xyz[3 * i], src + static_cast<std::size_t>(i) * step, 12);
    std::vector<unsigned> tri(3 * static_cast<std::size_t>(nt));
    lsh_mesh_copy_polygons(m, tri.data());
    lsh_mesh_destroy(m);
    return lsa_mesh_from_arrays(c, name, xyz.data(), nv, tri.data(), nt, 16);
}

unsigned lsa_mesh_vertices(lsa_ctx* c, const char* name) { return c->meshes.at(name).mesh->cloud.width * c->meshes.at(name).mesh->cloud.height; }
unsigned lsa_mesh_polygons(lsa_ctx* c, const char* name) { return static_cast<unsigned>(c->meshes.at(name).mesh->polygons.size()); }

// MeshProjector::affineMeshCallback (MeshProjector.cpp:306-307) rewrites the cloud of the tracked mesh IN PLACE
int lsa_mesh_set_vertices(lsa_ctx* c, const char* name, const float* xyz, unsigned seq)
{
    auto& cloud = c->meshes.at(name).mesh->cloud;
    const unsigned n = cloud.width * cloud.height;
    for (unsigned i = 0; i < n; ++i) std::memcpy(cloud.data.data() + static_cast<std::size_t>(i) * cloud.point_step, xyz + 3 * i, 12);
    cloud.header.seq = seq;
    return 0;
}

void lsa_mesh_set_displacement(lsa_ctx* c, const char* name, const float* linear, const float* angular)
{
    auto& tm = c->meshes.at(name);
    tm.linear = Eigen::Vector3f(linear[0], linear[1], linear[2]);
    tm.angular = Eigen::Vector3f(angular[0], angular[1], angular[2]);
}

// MeshProjector::addMeshToScene (MeshProjector.cpp:334)
int lsa_add_geometry(lsa_ctx* c, const char* name, int type)
{
    auto& tm = c->meshes.at(name);
    LSA_TRY(c->itracer->addGeometry(name, static_cast<RTCGeometryType>(type), tm.mesh->cloud.width * tm.mesh->cloud.height,
                                    static_cast<int>(tm.mesh->polygons.size())));
}
int lsa_remove_geometry(lsa_ctx* c, const char* name) { LSA_TRY(c->itracer->removeGeometry(name)); }
long lsa_geometry_count(lsa_ctx* c) { return c->itracer->getGeometryCount(); }

int lsa_update_components(lsa_ctx* c, const char* name)
{
    auto& tm = c->meshes.at(name);
    LSA_TRY(c->itracer->updateGeometry(name, tm.linear, tm.angular, tm.mesh));
}

int lsa_update_affine(lsa_ctx* c, const char* name, const float* affine3x4)
{
    Eigen::Affine3f T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) T(i, j) = affine3x4[4 * i + j];
    LSA_TRY(c->itracer->updateGeometry(name, T, c->meshes.at(name).mesh));
}

int lsa_commit(lsa_ctx* c) { LSA_TRY(c->itracer->commitScene()); }
int lsa_trace(lsa_ctx* c, unsigned frame) { LSA_TRY(c->itracer->traceScene(frame)); }

// MeshProjector::traceAffineMesh (MeshProjector.cpp:446-464), `frames` times; returns seconds per frame
double lsa_frame_loop(lsa_ctx* c, unsigned frames)
{
    try {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned f = 0; f < frames; ++f) {
            for (auto& kv : c->meshes) c->itracer->updateGeometry(kv.first, kv.second.linear, kv.second.angular, kv.second.mesh);
            c->itracer->commitScene();
            c->itracer->traceScene(++c->frameIndex);
        }
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (frames ? frames : 1);
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1.0;
    }
}

// the shared PointCloud2: out8 = width, height, point_step, row_step, seq, n_fields, is_bigendian, is_dense
const void* lsa_cloud(lsa_ctx* c, unsigned* out8, unsigned long long* data_bytes)
{
    auto cloud = c->itracer->getTraceCloud();
    out8[0] = cloud->width; out8[1] = cloud->height; out8[2] = cloud->point_step; out8[3] = cloud->row_step;
    out8[4] = cloud->header.seq; out8[5] = static_cast<unsigned>(cloud->fields.size());
    out8[6] = cloud->is_bigendian; out8[7] = cloud->is_dense;
    *data_bytes = cloud->data.size();
    return cloud->data.data();
}

void lsa_set_skip_unchanged(lsa_ctx* c, int on)
{
    c->tracer->setMeshPolicy(on ? HipTracer::MeshPolicy::SkipUnchanged : HipTracer::MeshPolicy::UploadAlways);
}
void lsa_upload_counts(lsa_ctx* c, unsigned long long* uploads, unsigned long long* skipped)
{
    *uploads = c->tracer->getUploadCount();
    *skipped = c->tracer->getSkippedUploadCount();
}
void* lsa_handle(lsa_ctx* c) { return c->tracer->getHandle(); }

// EmbreeTracer's per-name getters (EmbreeTracer.cpp:82-113, :369-415).  what: 0 getGeometryId, 1 getGeometryType,
// 2 getVertexCount, 3 getElementCount.  A TraceException becomes -1000 - its code.
long lsa_getter(lsa_ctx* c, int what, const char* name)
{
    try {
        switch (what) {
        case 0: return c->tracer->getGeometryId(name);
        case 1: return static_cast<long>(c->tracer->getGeometryType(name));
        case 2: return c->tracer->getVertexCount(name);
        case 3: return c->tracer->getElementCount(name);
        default: return -2;
        }
    } catch (const TraceException& e) {
        g_error = e.what();
        return -1000 - static_cast<long>(e.getErrorCode());
    }
}
// HipTracer::invalidateMesh: how an in-place vertex edit that leaves the header alone is announced under SkipUnchanged
void lsa_invalidate_mesh(lsa_ctx* c, const char* name) { c->tracer->invalidateMesh(name); }

// ITracer::setSensorConfig (ITracer.cpp:48) with a NEW LidarDevice made from `sensor_config`, through the base-class pointer
int lsa_set_sensor_config(lsa_ctx* c, const char* sensor_config)
{
    try {
        c->device = LidarDevice::create(sensor_config);
        c->itracer->setSensorConfig(c->device);
        return 0;
    } catch (const std::exception& e) { g_error = e.what(); return -100; }
}
// LidarDevice::initialize (LidarDevice.hpp:116-117): the SAME device object reads another configuration
int lsa_reinitialize_sensor(lsa_ctx* c, const char* sensor_config)
{
    try { c->device->initialize(sensor_config); return 0; } catch (const std::exception& e) { g_error = e.what(); return -100; }
}
void lsa_invalidate_sensor(lsa_ctx* c) { c->tracer->invalidateSensor(); }
unsigned long long lsa_sensor_probe_count(lsa_ctx* c) { return c->tracer->getSensorProbeCount(); }

}  // extern "C"
#pragma GCC visibility pop
