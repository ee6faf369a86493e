// HipTracer.hpp -- lidarshooter::ITracer backend for AMD MI355X, over the C ABI of liblidarshooter_hip.so.
//
// Drop this file into ros_ws/src/lidarshooter/src/ next to EmbreeTracer.hpp / OptixTracer.hpp; nothing else
// of lidarshooter changes except the backend-selection line (mainwindow.cpp:256-258) and three CMake lines
// (INTEGRATION.md).  It compiles against the UNMODIFIED reference headers: everything it needs from the
// sensor it gets through LidarDevice's public interface (LidarDevice.hpp:116-300) --
//   * ray directions: one walk of nextRay1() over the V*H rays at construction (LidarDevice.cpp:160-198),
//     from which the exact libm factor tables sin/cos(theta_v), sin/cos(phi_h) are recovered and verified
//     bit for bit against every walked direction;
//   * sensor pose: originToSensorInverse(0) = t exactly, and originToSensor(t + 2^k e_i) = 2^k * Rinv[:,i]
//     exactly (LidarDevice.cpp:383-401);
//   * header: initMessage() every frame (LidarDevice.cpp:94-115), as EmbreeTracer::traceScene does.
// The sensor is not a snapshot: EmbreeTracer::traceScene reads its LidarDevice every frame (EmbreeTracer.cpp:299-307), so
// ITracer::setSensorConfig (ITracer.cpp:48) or a LidarDevice initialised again in place (LidarDevice.hpp:116-117) takes
// effect at the next trace.  Here every traceScene compares a fingerprint of the current device -- its address, ray /
// channel / column counts, the pose (t and Rinv through originToSensor[Inverse]) and the first three walked rays --
// with the probed one; a difference walks the new device and hands the library the new tables (ls_tracer_set_sensor_tables:
// the geometries stay).  A change the fingerprint cannot see -- an in-place edit of the channel table beyond channel 0
// with everything else equal -- is announced with invalidateSensor().
//
// Per frame (MeshProjector.cpp:446-464: updateGeometry for every mesh, commitScene, traceScene):
//   * polygons are flattened and uploaded once per mesh and again only when their fingerprint changes (the
//     reference never edits them after pcl::io::loadPolygonFileSTL, mainwindow.cpp:146);
//   * vertices are uploaded every frame by default (MeshProjector::affineMeshCallback rewrites the cloud in
//     place, MeshProjector.cpp:306-307, and checking 8 MB costs what sending 8 MB costs).  setMeshPolicy(
//     SkipUnchanged) or LIDARSHOOTER_HIP_SKIP_UNCHANGED=1 is a CONTRACT, not a guess: an update whose cloud
//     is the same buffer of the same size with the same header.seq and header.stamp as the one uploaded last
//     is taken to hold the same vertices and becomes a transform-only update -- no copy at all, what a
//     joystick-driven pose change is (AffineMesh.cpp:108-128).  Whoever edits vertices in place under that
//     policy bumps header.seq / header.stamp or calls invalidateMesh(name) (one line in
//     MeshProjector::affineMeshCallback); the tracer does not sample the data to second-guess it;
//   * the vertex transform (MeshTransformer.cpp:142-205), ray generation, closest hit and 32-byte point
//     packing (XYZIRBytes.cpp:24-40) run on the GPU; the points land in pinned host memory (16 bytes each:
//     the other 16 of a record are constants) with one host wait and are expanded into PointCloud2::data by
//     the library's copy threads.
#pragma once

#include <lidarshooter_hip.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "Exceptions.hpp"
#include "ITracer.hpp"
#include "LidarDevice.hpp"

namespace lidarshooter
{

class HipTracer : public ITracer, public std::enable_shared_from_this<HipTracer>
{
public:
    using Ptr = std::shared_ptr<HipTracer>;
    using ConstPtr = std::shared_ptr<HipTracer const>;

    /// What an updateGeometry() call does with the vertex data it is handed
    enum class MeshPolicy {
        UploadAlways,   ///< default: vertices are sent every frame (~0.15 ms per 8 MB), whatever the caller did to them
        SkipUnchanged   ///< contract: same buffer + size + header.seq + header.stamp as last time == same vertices ->
                        ///< transform-only update; in-place edits are announced by the header or invalidateMesh()
    };

    /// Same signature as EmbreeTracer::create (EmbreeTracer.hpp:51) / OptixTracer::create (OptixTracer.hpp:91)
    static HipTracer::Ptr create(LidarDevice::Ptr _sensorConfig, sensor_msgs::PointCloud2::Ptr _traceStorage = nullptr,
                                 std::shared_ptr<spdlog::logger> _logger = nullptr, int _hipDevice = 0)
    {
        return HipTracer::Ptr(new HipTracer(_sensorConfig, _traceStorage, _logger, _hipDevice));
    }

    ITracer::Ptr getPtr() override { return shared_from_this(); }

    ~HipTracer() override
    {
        if (_handle != nullptr) ls_tracer_destroy(_handle);
    }

    /// ITracer.hpp:50; returns the geometry id like EmbreeTracer (lowest free id), 0 for an unsupported type
    int addGeometry(const std::string& _meshName, enum RTCGeometryType _geometryType, int _numVertices, int _numElements) override
    {
        std::lock_guard<std::mutex> lock(_mutex);
        const int rc = ls_add_geometry(_handle, _meshName.c_str(), static_cast<int>(_geometryType), _numVertices, _numElements);
        if (rc == LS_ERR_UNSUPPORTED_TYPE) return 0;   // EmbreeTracer.cpp:200-201 `return false`
        if (rc < 0) return rc;
        MeshState state;
        state.verticesPerElement = _geometryType == RTC_GEOMETRY_TYPE_QUAD ? 4u : 3u;
        state.numVertices = static_cast<std::size_t>(_numVertices);
        state.numElements = static_cast<std::size_t>(_numElements);
        _meshes[_meshName] = std::move(state);
        setGeometryCount(getGeometryCount() + 1);
        return rc;
    }

    /// ITracer.hpp:59; -1 when the name is unknown (EmbreeTracer.cpp:224-225); the scene is re-committed (:252)
    int removeGeometry(const std::string& _meshName) override
    {
        std::lock_guard<std::mutex> lock(_mutex);
        const bool known = _meshes.count(_meshName) != 0;
        const int rc = ls_remove_geometry(_handle, _meshName.c_str());
        if (known && rc != -1) {
            // the library erased the geometry before it re-committed: the bookkeeping follows even when that commit failed
            _meshes.erase(_meshName);
            setGeometryCount(getGeometryCount() - 1);
        }
        if (rc < -1) throwCommitError(rc);   // the commit inside failed, as commitScene() reports it
        return rc;
    }

    /// ITracer.hpp:69
    int updateGeometry(const std::string& _meshName, Eigen::Affine3f _transform, pcl::PolygonMesh::Ptr& _mesh) override
    {
        float affine[12];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) affine[4 * i + j] = _transform(i, j);
        return updateFromMesh(_meshName, affine, _mesh);
    }

    /// ITracer.hpp:80; T = Translation * Rz * Ry * Rx (MeshTransformer.cpp:467-477), built by the library
    int updateGeometry(const std::string& _meshName, Eigen::Vector3f _translation, Eigen::Vector3f _rotation,
                       pcl::PolygonMesh::Ptr& _mesh) override
    {
        const float lin[3] = {_translation.x(), _translation.y(), _translation.z()};
        const float ang[3] = {_rotation.x(), _rotation.y(), _rotation.z()};
        float affine[12];
        ls_affine_from_components(lin, ang, affine);
        return updateFromMesh(_meshName, affine, _mesh);
    }

    /// ITracer.hpp:87; -1 on an empty scene (OptixTracer.cpp:266-267)
    int commitScene() override
    {
        std::lock_guard<std::mutex> lock(_mutex);
        const int rc = ls_commit_scene(_handle);
        if (rc < -1) throwCommitError(rc);
        return rc;
    }

    /// ITracer.hpp:94; fills the shared PointCloud2 in place (EmbreeTracer.cpp:297-367)
    int traceScene(std::uint32_t _frameIndex) override
    {
        std::lock_guard<std::mutex> lock(_mutex);
        followSensor();   // (setSensorConfig / a re-initialised LidarDevice: EmbreeTracer.cpp:299-307 reads _config every frame)
        auto cloud = getTraceCloud();
        getSensorConfig()->initMessage(cloud, static_cast<int>(_frameIndex));
        getSensorConfig()->reset();
        // two steps: the hit count comes back while the points are still being packed on the device; the cloud is sized
        // meanwhile, then filled -- its first half while the second half is still crossing PCIe
        std::uint32_t n_points = 0;
        const int rc = ls_trace_scene_begin(_handle, _frameIndex, &n_points);
        if (rc < -1) {
            cloud->data.clear();
            throw TraceException(__FILE__, ls_last_error(_handle), rc);
        }
        // no clear() first: shrinking costs nothing and growing value-initialises only the difference
        cloud->data.resize(static_cast<std::size_t>(n_points) * 32u);
        if (rc >= 0) {
            const int erc = ls_trace_scene_expand(_handle, cloud->data.data());
            if (erc < 0) {
                cloud->data.clear();
                throw TraceException(__FILE__, ls_last_error(_handle), erc);
            }
        }
        cloud->width = n_points;   // EmbreeTracer.cpp:364; height 1, row_step 0 from initMessage
        return rc;
    }

    // ---- the per-name getters of EmbreeTracer, with its behaviour for unknown names
    /// EmbreeTracer.cpp:82-89: -1 when the name is unknown (no exception)
    int getGeometryId(const std::string& _meshName) const
    {
        const int rc = ls_geometry_id(_handle, _meshName.c_str());
        return rc < 0 ? -1 : rc;
    }
    /// EmbreeTracer.cpp:103-113 (test/EmbreeTracer_test.cpp:116-120): TraceException code 8 when unknown
    RTCGeometryType getGeometryType(const std::string& _meshName)
    {
        const int rc = ls_geometry_type(_handle, _meshName.c_str());
        if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in geometry types map", 8);
        return static_cast<RTCGeometryType>(rc);
    }
    /// EmbreeTracer.cpp:369-379: code 1
    long getVertexCount(const std::string& _meshName)
    {
        const long rc = ls_vertex_count(_handle, _meshName.c_str());
        if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in vertex count map", 1);
        return rc;
    }
    /// EmbreeTracer.cpp:405-415: code 4
    long getElementCount(const std::string& _meshName)
    {
        const long rc = ls_element_count(_handle, _meshName.c_str());
        if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in element count map", 4);
        return rc;
    }

    void setMeshPolicy(MeshPolicy _policy) { _policy_ = _policy; }
    MeshPolicy getMeshPolicy() const { return _policy_; }
    /// Forget what is known about a mesh: its next update re-sends vertices and polygons.  Under SkipUnchanged this is how
    /// an in-place edit that leaves the cloud's header alone is announced.
    void invalidateMesh(const std::string& _meshName)
    {
        std::lock_guard<std::mutex> lock(_mutex);
        auto it = _meshes.find(_meshName);
        if (it != _meshes.end()) it->second.haveVertices = it->second.haveElements = false;
    }
    /// The next traceScene walks the sensor again, whatever its fingerprint says (an in-place edit the fingerprint cannot see)
    void invalidateSensor()
    {
        std::lock_guard<std::mutex> lock(_mutex);
        _sensorValid = false;
    }
    /// How many times the sensor was probed (1 = at construction only)
    std::uint64_t getSensorProbeCount() const { return _sensorProbes; }
    /// Number of updateGeometry calls that copied vertex data / that were transform-only (diagnostics)
    std::uint64_t getUploadCount() const { return _uploads; }
    std::uint64_t getSkippedUploadCount() const { return _skipped; }
    /// The C-ABI handle (asynchronous / multi-GPU extensions, include/lidarshooter_hip.h)
    ls_tracer* getHandle() { return _handle; }

private:
    struct MeshState {
        std::size_t numVertices = 0, numElements = 0, verticesPerElement = 3;   // 4: RTC_GEOMETRY_TYPE_QUAD
        bool haveVertices = false, haveElements = false;
        // identity of the data handed over last time
        const void* polygonStorage = nullptr;
        std::size_t polygonCount = 0;
        std::uint64_t polygonProbe = 0;
        const void* vertexStorage = nullptr;
        std::size_t vertexBytes = 0;
        std::uint64_t stamp = 0;
        std::uint32_t seq = 0, pointStep = 0;
        std::uint64_t vertexSample = 0;    // unstamped clouds under SkipUnchanged: hash of 64 sampled vertices
        std::vector<std::uint32_t> flat;   // scratch of the one-off polygon flatten
    };

    HipTracer(LidarDevice::Ptr _sensorConfig, sensor_msgs::PointCloud2::Ptr _traceStorage, std::shared_ptr<spdlog::logger> __logger,
              int _hipDevice)
        : ITracer(_sensorConfig, _traceStorage, __logger)
    {
        if (const char* e = std::getenv("LIDARSHOOTER_HIP_SKIP_UNCHANGED"))
            if (e[0] == '1') _policy_ = MeshPolicy::SkipUnchanged;
        // (the header this file was compiled against and the library it was linked with: the same ABI, or nothing is called)
        if (ls_abi_version() != LS_ABI_VERSION)
            throw TraceException(__FILE__, "liblidarshooter_hip.so speaks another ABI than the lidarshooter_hip.h this adapter was built with", ls_abi_version());
        SensorProbe probe(*_sensorConfig);
        const ls_sensor_tables tables = probe.tables();
        _sensorPrint = SensorFingerprint(*_sensorConfig);
        _sensorValid = true;
        _sensorProbes = 1;
        const int rc = ls_tracer_create_tables(&tables, _hipDevice, &_handle);
        if (rc != LS_OK)
            throw TraceException(__FILE__, rc == LS_ERR_NO_DEVICE ? "no usable HIP device for the MI355X tracer (there is no CPU fallback)"
                                                                  : "ls_tracer_create_tables rejected the probed sensor", rc);
        ls_tracer_set_option(_handle, LS_OPT_READBACK_HITS, 0);   // the cloud only carries the 32-byte points
        ls_tracer_set_option(_handle, LS_OPT_HOST_OUTPUT, 2);     // 16 bytes per point over PCIe, expanded into cloud->data
        if (_logger) _logger->debug("HipTracer: {} channels x {} azimuths, {} host copy threads", tables.n_vertical, tables.h_count,
                                    ls_get_info(_handle, LS_INFO_HOST_THREADS));
    }

    // What one frame can afford to ask the sensor (a dozen calls): enough to notice another device, another raster, another
    // pose, another azimuth table or another first channel.  Compared bit for bit.
    struct SensorFingerprint {
        const void* device = nullptr;
        unsigned rays = 0, channels = 0, columns = 0;
        float words[3 + 9 + 9] = {};   // t; Rinv applied to the unit vectors (offset by t); the first three walked directions
        SensorFingerprint() = default;
        explicit SensorFingerprint(LidarDevice& d)
        {
            device = &d;
            rays = d.getTotalRays();
            channels = d.getTotalChannels();
            columns = d.getScanRayCount();
            Eigen::Vector3f p(0.0f, 0.0f, 0.0f);
            d.originToSensorInverse(p);
            words[0] = p.x(); words[1] = p.y(); words[2] = p.z();
            for (int i = 0; i < 3; ++i) {
                Eigen::Vector3f e(i == 0 ? 1.0f : 0.0f, i == 1 ? 1.0f : 0.0f, i == 2 ? 1.0f : 0.0f);
                d.originToSensor(e);
                words[3 + 3 * i] = e.x(); words[4 + 3 * i] = e.y(); words[5 + 3 * i] = e.z();
            }
            d.reset();
            const unsigned n = rays < 3u ? rays : 3u;
            for (unsigned r = 0; r < n; ++r) {
                RTCRayHit ray;
                int valid = -1;
                d.nextRay1(ray, &valid);
                words[12 + 3 * r] = ray.ray.dir_x; words[13 + 3 * r] = ray.ray.dir_y; words[14 + 3 * r] = ray.ray.dir_z;
            }
            d.reset();
        }
        bool operator==(const SensorFingerprint& o) const
        {
            return device == o.device && rays == o.rays && channels == o.channels && columns == o.columns &&
                   std::memcmp(words, o.words, sizeof(words)) == 0;
        }
    };

    // called with _mutex held, at the top of traceScene
    void followSensor()
    {
        LidarDevice::Ptr device = getSensorConfig();
        if (!device) throw TraceException(__FILE__, "traceScene without a sensor (setSensorConfig(nullptr))", 1);
        const SensorFingerprint now(*device);
        if (_sensorValid && now == _sensorPrint) return;
        SensorProbe probe(*device);   // throws when the device cannot be expressed as factor tables
        const ls_sensor_tables tables = probe.tables();
        const int rc = ls_tracer_set_sensor_tables(_handle, &tables);
        if (rc < -1) throw TraceException(__FILE__, ls_last_error(_handle), rc);
        _sensorPrint = now;
        _sensorValid = true;
        ++_sensorProbes;
        if (_logger) _logger->debug("HipTracer: the sensor changed: {} channels x {} azimuths now", tables.n_vertical, tables.h_count);
    }

    // FNV-1a over a few words: identity probes, not integrity checks
    static std::uint64_t mix(std::uint64_t h, std::uint64_t x)
    {
        for (int i = 0; i < 8; ++i) { h ^= (x >> (8 * i)) & 0xFFu; h *= 1099511628211ull; }
        return h;
    }

    static std::uint64_t probePolygons(const pcl::PolygonMesh& mesh)
    {
        std::uint64_t h = 1469598103934665603ull;
        const std::size_t n = mesh.polygons.size(), step = n > 16 ? n / 16 : 1;   // every pcl::Vertices is its own heap block: a cache miss each
        for (std::size_t i = 0; i < n; i += step) {
            const auto& p = mesh.polygons[i].vertices;
            h = mix(h, p.size());
            for (std::size_t k = 0; k < p.size() && k < 4; ++k) h = mix(h, static_cast<std::uint64_t>(p[k]));
        }
        if (n) {
            const auto& p = mesh.polygons[n - 1].vertices;
            for (std::size_t k = 0; k < p.size() && k < 4; ++k) h = mix(h, static_cast<std::uint64_t>(p[k]));
        }
        return h;
    }

    static std::uint64_t probeVertices(const std::uint8_t* data, std::size_t n, std::size_t pointStep)
    {
        std::uint64_t h = 1469598103934665603ull;
        const std::size_t step = n > 64 ? n / 64 : 1;
        for (std::size_t i = 0; i < n; i += step) {
            std::uint32_t w[3];
            std::memcpy(w, data + i * pointStep, 12);
            h = mix(h, (static_cast<std::uint64_t>(w[0]) << 32) | w[1]);
            h = mix(h, w[2]);
        }
        return h;
    }

    int updateFromMesh(const std::string& _meshName, const float affine[12], pcl::PolygonMesh::Ptr& _mesh)
    {
        std::lock_guard<std::mutex> lock(_mutex);
        auto it = _meshes.find(_meshName);
        if (it == _meshes.end())
            throw TraceException(__FILE__, "Geometry key does not exist in geometry types map", 8);   // EmbreeTracer.cpp:102-113
        if (!_mesh) throw TraceException(__FILE__, "updateGeometry was handed a null mesh", 1);
        MeshState& st = it->second;
        const auto& cloud = _mesh->cloud;
        const std::size_t pointStep = cloud.point_step;
        if (pointStep < 12 || (pointStep & 3u) != 0)
            throw TraceException(__FILE__, "mesh cloud point_step must be a multiple of 4 and hold x, y, z float32 first", 1);
        if (cloud.data.size() < st.numVertices * pointStep)
            throw TraceException(__FILE__, "mesh cloud holds fewer vertices than addGeometry registered", 1);
        if (_mesh->polygons.size() != st.numElements)
            throw TraceException(__FILE__, "mesh holds a different number of polygons than addGeometry registered", 1);

        // ---- polygons: flatten + upload only when they are new (MeshTransformer.cpp:486-520 does it every frame)
        const std::uint32_t* indices = nullptr;
        const std::uint64_t polygonProbe = probePolygons(*_mesh);
        if (!st.haveElements || st.polygonStorage != static_cast<const void*>(_mesh->polygons.data()) ||
            st.polygonCount != _mesh->polygons.size() || st.polygonProbe != polygonProbe) {
            const std::size_t per = st.verticesPerElement;   // MeshTransformer.cpp:499-538: 3 for triangles, 4 for quads
            st.flat.resize(st.numElements * per);
            std::size_t k = 0;
            for (const auto& polygon : _mesh->polygons) {
                if (polygon.vertices.size() != per)
                    throw TraceException(__FILE__, "Geometry does not match element vertex count", per == 3 ? 1 : 2);
                for (std::size_t c = 0; c < per; ++c) st.flat[k++] = static_cast<std::uint32_t>(polygon.vertices[c]);
            }
            indices = st.flat.data();
        }

        // ---- vertices
        const std::uint8_t* vertexData = cloud.data.data();
        const std::size_t vertexBytes = st.numVertices * pointStep;
        const std::uint64_t stamp = static_cast<std::uint64_t>(cloud.header.stamp);
        const std::uint32_t seq = static_cast<std::uint32_t>(cloud.header.seq);
        // SkipUnchanged is a contract (see MeshPolicy): identity of buffer and header stands for identity of content
        // An UNSTAMPED cloud (seq and stamp both zero: a publisher that never fills the header, or a mesh loaded from a file
        // and copied over in place, MeshProjector.cpp:306-307) cannot announce an edit through its header: for those, and only
        // those, 64 evenly spaced vertices are hashed as a safety net -- it catches a rigid edit of the whole cloud, not an
        // edit of vertices it does not sample; invalidateMesh() stays the contract
        const bool unstamped = seq == 0u && stamp == 0u;
        std::uint64_t sample = 0;
        if (unstamped) sample = probeVertices(vertexData, st.numVertices, pointStep);   // (whatever the policy: it may change between frames)
        if (_policy_ == MeshPolicy::SkipUnchanged && unstamped) {
            if (!_warnedUnstamped && _logger) {
                _logger->warn("HipTracer: SkipUnchanged with an unstamped mesh cloud ('{}'): in-place vertex edits must be announced "
                              "with invalidateMesh(); a 64-vertex sample is compared as a safety net", _meshName);
                _warnedUnstamped = true;
            }
        }
        const bool unchanged = _policy_ == MeshPolicy::SkipUnchanged && st.haveVertices && indices == nullptr &&
                               st.vertexStorage == static_cast<const void*>(vertexData) && st.vertexBytes == vertexBytes &&
                               st.pointStep == pointStep && st.seq == seq && st.stamp == stamp && (!unstamped || st.vertexSample == sample);
        int rc;
        if (unchanged) {
            rc = ls_update_geometry_transform(_handle, _meshName.c_str(), affine);
            ++_skipped;
        } else {
            rc = ls_update_geometry(_handle, _meshName.c_str(), affine, vertexData, static_cast<std::uint32_t>(pointStep), indices);
            ++_uploads;
        }
        if (rc < 0) throw TraceException(__FILE__, ls_last_error(_handle), rc);
        st.haveVertices = true;
        st.vertexStorage = vertexData;
        st.vertexBytes = vertexBytes;
        st.pointStep = static_cast<std::uint32_t>(pointStep);
        st.seq = seq;
        st.stamp = stamp;
        st.vertexSample = sample;
        if (indices != nullptr) {
            st.haveElements = true;
            st.polygonStorage = _mesh->polygons.data();
            st.polygonCount = _mesh->polygons.size();
            st.polygonProbe = polygonProbe;
            std::vector<std::uint32_t>().swap(st.flat);   // 12 MB per million triangles: not kept
        }
        return 0;
    }

public:
    // ------------------------------------------------------------------------------------------------
    // Everything the tracer needs to know about the sensor, recovered through LidarDevice's public
    // interface and checked bit for bit against it.  The reference forms ray (v, h) as
    //   theta = float((90.0 - chi_v) * pi / 180), phi = float((begin + step * float(h)) * pi / 180),
    //   d = (sinf(theta) cosf(phi), sinf(theta) sinf(phi), cosf(theta))          (LidarDevice.cpp:176-186)
    // so two floats (begin, step) and one float per channel (theta_v) determine all V*H directions; they
    // are found by trying the few floats around an arctangent estimate and keeping the ones that reproduce
    // every walked direction exactly.
    // ------------------------------------------------------------------------------------------------
    class SensorProbe
    {
    public:
        explicit SensorProbe(LidarDevice& device)
        {
            V = device.getTotalChannels();
            H = device.getScanRayCount();
            if (V == 0 || H < 2 || static_cast<unsigned long long>(V) * H != device.getTotalRays())
                fail("the device reports an empty or inconsistent ray raster");
            walk(device);
            recoverChannels();
            recoverColumns();
            verify();
            recoverPose(device);
        }

        ls_sensor_tables tables() const
        {
            ls_sensor_tables t;
            std::memset(&t, 0, sizeof(t));
            t.sin_theta = sinTheta.data();
            t.cos_theta = cosTheta.data();
            t.elevation_deg = elevationDeg.data();
            t.n_vertical = V;
            t.sin_phi = sinPhi.data();
            t.cos_phi = cosPhi.data();
            t.h_count = H;
            t.h_begin_deg = beginDeg;
            t.h_step_deg = stepDeg;
            std::memcpy(t.Rinv, Rinv, sizeof(Rinv));
            std::memcpy(t.t, translation, sizeof(translation));
            return t;
        }

    private:
        static constexpr int kNeighbours = 6;   // floats tried on each side of an estimate
        static constexpr double kPi = 3.14159265358979323846;

        [[noreturn]] static void fail(const std::string& what)
        {
            throw TraceException(__FILE__, "HipTracer cannot derive the sensor from LidarDevice's public interface: " + what, 1);
        }
        static float stepFloat(float x, int n)
        {
            for (; n > 0; --n) x = std::nextafter(x, std::numeric_limits<float>::infinity());
            for (; n < 0; ++n) x = std::nextafter(x, -std::numeric_limits<float>::infinity());
            return x;
        }
        static bool sameBits(float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a == 0.0f && b == 0.0f); }

        void walk(LidarDevice& device)
        {
            const std::size_t n = static_cast<std::size_t>(V) * H;
            dx.resize(n);
            dy.resize(n);
            dz.resize(n);
            device.reset();
            for (std::size_t r = 0; r < n; ++r) {   // channel-major: r = v * H + h (LidarDevice.cpp:824-845)
                RTCRayHit ray;
                int valid = -1;
                device.nextRay1(ray, &valid);
                dx[r] = ray.ray.dir_x;
                dy[r] = ray.ray.dir_y;
                dz[r] = ray.ray.dir_z;
            }
            device.reset();
        }

        // per channel: every float theta near the estimate with cosf(theta) == dz; the sine is settled later
        void recoverChannels()
        {
            cosTheta.resize(V);
            thetaCandidates.resize(V);
            double best = -1.0;
            for (unsigned v = 0; v < V; ++v) {
                const std::size_t r0 = static_cast<std::size_t>(v) * H;
                for (unsigned h = 1; h < H; ++h)
                    if (!sameBits(dz[r0 + h], dz[r0])) fail("a channel's z direction varies with azimuth");
                cosTheta[v] = dz[r0];
                const double rho = std::hypot(static_cast<double>(dx[r0]), static_cast<double>(dy[r0]));
                const float estimate = static_cast<float>(std::atan2(rho, static_cast<double>(dz[r0])));
                for (int k = -kNeighbours; k <= kNeighbours; ++k) {
                    const float theta = stepFloat(estimate, k);
                    if (sameBits(std::cos(theta), cosTheta[v])) thetaCandidates[v].push_back(theta);
                }
                if (thetaCandidates[v].empty()) fail("no float polar angle reproduces a channel's z direction");
                if (rho > best) { best = rho; reference = v; }
            }
            if (!(best > 1e-3)) fail("every channel points along the vertical axis");
        }

        bool columnsMatch(unsigned v, float sine, const std::vector<float>& c, const std::vector<float>& s) const
        {
            const std::size_t r0 = static_cast<std::size_t>(v) * H;
            for (unsigned h = 0; h < H; ++h)
                if (!sameBits(sine * c[h], dx[r0 + h]) || !sameBits(sine * s[h], dy[r0 + h])) return false;
            return true;
        }

        void buildColumns(float begin, float step, std::vector<float>& c, std::vector<float>& s) const
        {
            for (unsigned h = 0; h < H; ++h) {
                const float prePhi = begin + step * static_cast<float>(h);
                const float phi = static_cast<float>(static_cast<double>(prePhi) * kPi / 180.0);
                c[h] = std::cos(phi);
                s[h] = std::sin(phi);
            }
        }

        // (begin, step): estimated from the arctangents of the reference channel, then the floats around the
        // estimates are tried until one pair reproduces that channel's H directions exactly
        void recoverColumns()
        {
            const std::size_t r0 = static_cast<std::size_t>(reference) * H;
            std::vector<double> az(H);
            for (unsigned h = 0; h < H; ++h) az[h] = std::atan2(static_cast<double>(dy[r0 + h]), static_cast<double>(dx[r0 + h]));
            double span = 0.0;
            for (unsigned h = 0; h + 1 < H; ++h) {
                double d = az[h + 1] - az[h];
                while (d > kPi) d -= 2.0 * kPi;
                while (d <= -kPi) d += 2.0 * kPi;
                span += d;
            }
            const double stepEstimate = span / static_cast<double>(H - 1) * 180.0 / kPi;
            const double beginEstimate = az[0] * 180.0 / kPi;
            std::vector<float> c(H), s(H);
            // a two-column raster cannot tell a step from its 360-degree aliases; nor can any raster tell begin
            // from begin +- 360: the aliases are tried too (they differ in the rounding of phi)
            for (int beginTurn = 0; beginTurn < 3; ++beginTurn) {
                static const double turns[3] = {0.0, 360.0, -360.0};
                for (int stepTurn = 0; stepTurn < (H == 2 ? 3 : 1); ++stepTurn) {
                    const float b0 = static_cast<float>(beginEstimate + turns[beginTurn]);
                    const float s0 = static_cast<float>(stepEstimate + turns[stepTurn]);
                    for (int kb = 0; kb <= 2 * kNeighbours; ++kb) {
                        const float begin = stepFloat(b0, (kb & 1) ? (kb + 1) / 2 : -(kb / 2));
                        for (int ks = 0; ks <= 2 * kNeighbours; ++ks) {
                            const float step = stepFloat(s0, (ks & 1) ? (ks + 1) / 2 : -(ks / 2));
                            buildColumns(begin, step, c, s);
                            for (float theta : thetaCandidates[reference]) {
                                if (!columnsMatch(reference, std::sin(theta), c, s)) continue;
                                if (!allChannelsMatch(c, s)) continue;
                                beginDeg = begin;
                                stepDeg = step;
                                cosPhi = c;
                                sinPhi = s;
                                return;
                            }
                        }
                    }
                }
            }
            fail("no (begin, step) pair reproduces the walked azimuth directions");
        }

        // settles sin(theta_v) for every channel against the given columns
        bool allChannelsMatch(const std::vector<float>& c, const std::vector<float>& s)
        {
            std::vector<float> sines(V), elevations(V);
            for (unsigned v = 0; v < V; ++v) {
                bool found = false;
                for (float theta : thetaCandidates[v]) {
                    const float sine = std::sin(theta);
                    if (!columnsMatch(v, sine, c, s)) continue;
                    sines[v] = sine;
                    elevations[v] = static_cast<float>(90.0 - static_cast<double>(theta) * 180.0 / kPi);
                    found = true;
                    break;
                }
                if (!found) return false;
            }
            sinTheta = sines;
            elevationDeg = elevations;
            return true;
        }

        void verify() const
        {
            for (unsigned v = 0; v < V; ++v)
                for (unsigned h = 0; h < H; ++h) {
                    const std::size_t r = static_cast<std::size_t>(v) * H + h;
                    if (!sameBits(sinTheta[v] * cosPhi[h], dx[r]) || !sameBits(sinTheta[v] * sinPhi[h], dy[r]) || !sameBits(cosTheta[v], dz[r]))
                        fail("the recovered tables do not reproduce every ray direction");
                }
        }

        // LidarDevice.cpp:383-401: originToSensorInverse(q) = R q + t, originToSensor(p) = Rinv (p - t)
        void recoverPose(LidarDevice& device)
        {
            Eigen::Vector3f origin(0.0f, 0.0f, 0.0f);
            device.originToSensorInverse(origin);   // R * 0 + t = t, exactly
            translation[0] = origin.x();
            translation[1] = origin.y();
            translation[2] = origin.z();
            for (int i = 0; i < 3; ++i) {
                // a power of two that adds to and subtracts from t_i without rounding: (t_i + s) - t_i == s
                float scale = 0.0f, shifted = 0.0f;
                for (int k = 0; k <= 40 && scale == 0.0f; ++k) {
                    for (int sign = 0; sign < 2 && scale == 0.0f; ++sign) {
                        const float s = std::ldexp(1.0f, sign ? -k : k);
                        volatile float x = translation[i] + s;
                        volatile float back = x - translation[i];
                        if (back == s && std::isfinite(x)) { scale = s; shifted = x; }
                    }
                }
                if (scale == 0.0f) fail("the sensor translation has no exactly invertible offset");
                Eigen::Vector3f probe(translation[0], translation[1], translation[2]);
                if (i == 0) probe = Eigen::Vector3f(shifted, translation[1], translation[2]);
                if (i == 1) probe = Eigen::Vector3f(translation[0], shifted, translation[2]);
                if (i == 2) probe = Eigen::Vector3f(translation[0], translation[1], shifted);
                device.originToSensor(probe);        // Rinv * (0, .., s, .., 0) = s * column i, exactly
                Rinv[0 + i] = probe.x() / scale;
                Rinv[3 + i] = probe.y() / scale;
                Rinv[6 + i] = probe.z() / scale;
            }
            for (float r : Rinv)
                if (!std::isfinite(r)) fail("the sensor rotation is not finite");
        }

        unsigned V = 0, H = 0, reference = 0;
        std::vector<float> dx, dy, dz;
        std::vector<std::vector<float>> thetaCandidates;
        std::vector<float> sinTheta, cosTheta, elevationDeg, sinPhi, cosPhi;
        float beginDeg = 0.0f, stepDeg = 0.0f;
        float Rinv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, translation[3] = {0, 0, 0};
    };

private:

    std::mutex _mutex;   // add/remove come from the GUI thread, update/commit/trace from the ROS spinner (mainwindow.cpp:153,320)
    /// a failed commit: LS_ERR_OUT_OF_RANGE is the library refusing a mesh -- a triangle names a vertex the geometry does not
    /// have; Embree would read it out of the shared buffers unchecked (EmbreeTracer.cpp:140-176), the GPU would fault --
    /// which is what BadGeometryException (Exceptions.hpp:143-157) is for; anything else stays a TraceException
    [[noreturn]] void throwCommitError(int rc)
    {
        if (rc == LS_ERR_OUT_OF_RANGE) {
            // the refused geometry's own type (the library's message names it: "geometry '<name>': ...")
            const std::string msg = ls_last_error(_handle);
            RTCGeometryType type = RTC_GEOMETRY_TYPE_TRIANGLE;
            const std::size_t a = msg.find('\''), b = a == std::string::npos ? a : msg.find('\'', a + 1);
            if (b != std::string::npos) {
                const auto it = _meshes.find(msg.substr(a + 1, b - a - 1));
                if (it != _meshes.end() && it->second.verticesPerElement == 4) type = RTC_GEOMETRY_TYPE_QUAD;
            }
            throw BadGeometryException(__FILE__, msg, rc, type);
        }
        throw TraceException(__FILE__, ls_last_error(_handle), rc);
    }

    std::map<std::string, MeshState> _meshes;
    MeshPolicy _policy_ = MeshPolicy::UploadAlways;
    std::uint64_t _uploads = 0, _skipped = 0;
    SensorFingerprint _sensorPrint;
    bool _sensorValid = false;
    std::uint64_t _sensorProbes = 0;
    bool _warnedUnstamped = false;
    ls_tracer* _handle = nullptr;
};

}  // namespace lidarshooter
