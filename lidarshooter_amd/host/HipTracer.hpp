// HipTracer.hpp -- the host-side tracer class: same surface as lidarshooter::ITracer
// (ITracer.hpp:29-152) / EmbreeTracer (EmbreeTracer.hpp:40-260), implemented over the C ABI of
// liblidarshooter_hip.so.  This is the class a user switches to; with ROS/PCL present the only
// differences are the message types (see INTEGRATION.md for the ROS-typed adapter).
#pragma once

#include <array>
#include <map>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lidarshooter_hip.h"
#include "HostTypes.hpp"
#include "LidarDevice.hpp"

namespace lidarshooter {

// Embree's enum values the reference passes through ITracer::addGeometry (ITracer.hpp:50)
enum RTCGeometryType { RTC_GEOMETRY_TYPE_TRIANGLE = 0, RTC_GEOMETRY_TYPE_QUAD = 1 };

using Affine3f = std::array<float, 12>;  // row-major 3x4 [linear | translation] (Eigen::Affine3f)
using Vector3f = std::array<float, 3>;

inline Affine3f AffineIdentity() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}; }

class HipTracer : public std::enable_shared_from_this<HipTracer> {
public:
    using Ptr = std::shared_ptr<HipTracer>;

    // EmbreeTracer::create(LidarDevice::Ptr, PointCloud2::Ptr = nullptr) (EmbreeTracer.hpp:51)
    static Ptr create(LidarDevice::Ptr _sensorConfig, std::shared_ptr<PointCloud2> _traceStorage = nullptr, int _hipDevice = 0);
    Ptr getPtr() { return shared_from_this(); }
    ~HipTracer();

    int addGeometry(const std::string& _meshName, RTCGeometryType _geometryType, int _numVertices, int _numElements);
    int removeGeometry(const std::string& _meshName);
    int updateGeometry(const std::string& _meshName, const Affine3f& _transform, const PolygonMesh& _mesh);
    int updateGeometry(const std::string& _meshName, const Vector3f& _translation, const Vector3f& _rotation, const PolygonMesh& _mesh);
    int commitScene();
    int traceScene(std::uint32_t _frameIndex);

    long getGeometryCount() const;
    // EmbreeTracer.cpp:82-113, :369-415: getGeometryId returns -1 for an unknown name, the others throw TraceException
    // (codes 8, 1, 4)
    int getGeometryId(const std::string& _meshName) const;
    RTCGeometryType getGeometryType(const std::string& _meshName);
    long getVertexCount(const std::string& _meshName);
    long getElementCount(const std::string& _meshName);

    std::shared_ptr<PointCloud2> getTraceCloud() { return _traceCloud; }
    void setTraceCloud(std::shared_ptr<PointCloud2> _traceStorage) { _traceCloud = std::move(_traceStorage); }
    LidarDevice::Ptr getSensorConfig() { return _config; }

    // per-hit records of the last frame (ray index, geomID, primID, t), same order as the cloud
    const ls_hit* getHits() const { return _lastHits; }
    std::uint32_t getHitCount() const { return _lastHitCount; }
    ls_tracer* handle() { return _handle; }

private:
    HipTracer(LidarDevice::Ptr _sensorConfig, std::shared_ptr<PointCloud2> _traceStorage, int _hipDevice);
    std::vector<std::uint32_t> flattenPolygons(const std::string& _meshName, const PolygonMesh& _mesh) const;
    std::map<std::string, std::size_t> _verticesPerElement;   // 3, or 4 for RTC_GEOMETRY_TYPE_QUAD

    LidarDevice::Ptr _config;
    std::shared_ptr<PointCloud2> _traceCloud;
    ls_tracer* _handle = nullptr;
    const ls_hit* _lastHits = nullptr;
    std::uint32_t _lastHitCount = 0;
};

}  // namespace lidarshooter
