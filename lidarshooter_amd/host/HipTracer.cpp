// HipTracer.cpp -- see HipTracer.hpp.  Pure plumbing: every numeric step happens behind the C ABI
// on the GPU; when the library cannot create a device context the constructor throws.
#include "HipTracer.hpp"

#include <cstring>

namespace lidarshooter {

HipTracer::Ptr HipTracer::create(LidarDevice::Ptr _sensorConfig, std::shared_ptr<PointCloud2> _traceStorage, int _hipDevice)
{
    return Ptr(new HipTracer(std::move(_sensorConfig), std::move(_traceStorage), _hipDevice));
}

HipTracer::HipTracer(LidarDevice::Ptr _sensorConfig, std::shared_ptr<PointCloud2> _traceStorage, int _hipDevice)
    : _config(std::move(_sensorConfig)), _traceCloud(std::move(_traceStorage))
{
    // ITracer.cpp:17-22: allocate the cloud when none is given, zero width/height
    if (!_traceCloud) {
        _traceCloud = std::make_shared<PointCloud2>();
        _traceCloud->width = 0;
        _traceCloud->height = 0;
    }
    const ls_sensor_desc desc = _config->sensorDesc();
    const int rc = ls_tracer_create(&desc, _hipDevice, &_handle);
    if (rc != LS_OK) throw TraceException(__FILE__, "cannot create the HIP tracer (no MI355X device or invalid sensor)", rc);
}

HipTracer::~HipTracer()
{
    if (_handle) ls_tracer_destroy(_handle);
}

int HipTracer::addGeometry(const std::string& _meshName, RTCGeometryType _geometryType, int _numVertices, int _numElements)
{
    const int rc = ls_add_geometry(_handle, _meshName.c_str(), static_cast<int>(_geometryType), _numVertices, _numElements);
    if (rc == LS_ERR_UNSUPPORTED_TYPE) return 0;  // EmbreeTracer.cpp:200-201 `return false`
    if (rc >= 0) _verticesPerElement[_meshName] = _geometryType == RTC_GEOMETRY_TYPE_QUAD ? 4u : 3u;
    return rc;
}

int HipTracer::removeGeometry(const std::string& _meshName)
{
    _verticesPerElement.erase(_meshName);
    return ls_remove_geometry(_handle, _meshName.c_str());
}

std::vector<std::uint32_t> HipTracer::flattenPolygons(const std::string& _meshName, const PolygonMesh& _mesh) const
{
    // MeshTransformer.cpp:486-538 copyElementsIntoBuffer: three indices per triangle, four per quad
    const auto it = _verticesPerElement.find(_meshName);
    const std::size_t per = it == _verticesPerElement.end() ? 3u : it->second;
    if (!_mesh.polygons.empty() && _mesh.polygons[0].vertices.size() != per)
        throw TraceException(__FILE__, "Geometry does not match element vertex count", per == 3 ? 1 : 2);
    std::vector<std::uint32_t> idx(_mesh.polygons.size() * per);
    std::size_t k = 0;
    for (const auto& poly : _mesh.polygons)
        for (std::size_t c = 0; c < per; ++c) idx[k++] = poly.vertices[c];
    return idx;
}

int HipTracer::updateGeometry(const std::string& _meshName, const Affine3f& _transform, const PolygonMesh& _mesh)
{
    const std::vector<std::uint32_t> idx = flattenPolygons(_meshName, _mesh);
    const int rc = ls_update_geometry(_handle, _meshName.c_str(), _transform.data(), _mesh.cloud.data.data(),
                                      _mesh.cloud.point_step, idx.data());
    if (rc == LS_ERR_UNKNOWN_GEOMETRY) throw TraceException(__FILE__, "Geometry key does not exist in geometry types map", 8);
    return rc;
}

int HipTracer::updateGeometry(const std::string& _meshName, const Vector3f& _translation, const Vector3f& _rotation, const PolygonMesh& _mesh)
{
    const std::vector<std::uint32_t> idx = flattenPolygons(_meshName, _mesh);
    const int rc = ls_update_geometry_components(_handle, _meshName.c_str(), _translation.data(), _rotation.data(),
                                                 _mesh.cloud.data.data(), _mesh.cloud.point_step, idx.data());
    if (rc == LS_ERR_UNKNOWN_GEOMETRY) throw TraceException(__FILE__, "Geometry key does not exist in geometry types map", 8);
    return rc;
}

int HipTracer::commitScene() { return ls_commit_scene(_handle); }

int HipTracer::traceScene(std::uint32_t _frameIndex)
{
    // EmbreeTracer.cpp:299-301: header, clear, then trace; width = number of points (:364)
    _config->initMessage(*_traceCloud, static_cast<int>(_frameIndex));
    _traceCloud->data.clear();
    _config->reset();
    ls_frame frame;
    const int rc = ls_trace_scene(_handle, _frameIndex, &frame);
    _lastHits = frame.hits;
    _lastHitCount = frame.n_points;
    if (rc < -1) throw TraceException(__FILE__, ls_last_error(_handle), rc);
    if (frame.n_points) _traceCloud->data.assign(frame.points32, frame.points32 + static_cast<std::size_t>(frame.n_points) * 32);
    _traceCloud->width = frame.n_points;
    return rc;
}

long HipTracer::getGeometryCount() const { return ls_geometry_count(_handle); }

// EmbreeTracer.cpp:82-89: -1 when the name is unknown (no exception)
int HipTracer::getGeometryId(const std::string& _meshName) const
{
    const int rc = ls_geometry_id(_handle, _meshName.c_str());
    return rc < 0 ? -1 : rc;
}

// EmbreeTracer.cpp:103-113: TraceException code 8 when the name is unknown
RTCGeometryType HipTracer::getGeometryType(const std::string& _meshName)
{
    const int rc = ls_geometry_type(_handle, _meshName.c_str());
    if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in geometry types map", 8);
    return static_cast<RTCGeometryType>(rc);
}

// EmbreeTracer.cpp:369-379: code 1
long HipTracer::getVertexCount(const std::string& _meshName)
{
    const long rc = ls_vertex_count(_handle, _meshName.c_str());
    if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in vertex count map", 1);
    return rc;
}

// EmbreeTracer.cpp:405-415: code 4
long HipTracer::getElementCount(const std::string& _meshName)
{
    const long rc = ls_element_count(_handle, _meshName.c_str());
    if (rc < 0) throw TraceException(__FILE__, "Geometry key does not exist in element count map", 4);
    return rc;
}

}  // namespace lidarshooter
