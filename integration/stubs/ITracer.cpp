// stub of src/ITracer.cpp: constructor semantics of ITracer.cpp:3-28 (cloud allocated when none is given,
// width/height zeroed, geometry count 0) and the trivial accessors
#include "ITracer.hpp"

namespace lidarshooter
{

ITracer::ITracer(LidarDevice::Ptr _sensorConfig, sensor_msgs::PointCloud2::Ptr _traceStorage, std::shared_ptr<spdlog::logger> __logger)
    : _logger(__logger ? __logger : spdlog::stdout_color_mt(LIDARSHOOTER_APPLICATION_NAME)), _config(_sensorConfig), _geometryCount(0),
      _traceCloud(_traceStorage ? _traceStorage : sensor_msgs::PointCloud2::Ptr(new sensor_msgs::PointCloud2()))
{
    _traceCloud->width = 0;
    _traceCloud->height = 0;
}

long ITracer::getGeometryCount() const { return _geometryCount; }
sensor_msgs::PointCloud2::Ptr ITracer::getTraceCloud() { return _traceCloud; }
void ITracer::setTraceCloud(sensor_msgs::PointCloud2::Ptr _traceStorage) { _traceCloud = _traceStorage; }
LidarDevice::Ptr ITracer::getSensorConfig() { return _config; }
void ITracer::setSensorConfig(LidarDevice::Ptr __config) { _config = __config; }
void ITracer::setGeometryCount(long _count) { _geometryCount = _count; }

}  // namespace lidarshooter
