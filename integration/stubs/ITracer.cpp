// Stand-in for src/ITracer.cpp: what the base class does at construction (ITracer.cpp:3-28 in the reference: a cloud
// is made when none is handed in, its width / height start at zero, no geometries yet) and its trivial accessors.
#include "ITracer.hpp"

namespace lidarshooter
{

ITracer::ITracer(LidarDevice::Ptr sensor, CloudPtr storage, LoggerPtr logger) : sensor_(sensor), cloud_(storage)
{
    _logger = logger ? logger : spdlog::stdout_color_mt(LIDARSHOOTER_APPLICATION_NAME);
    if (!cloud_) cloud_ = std::make_shared<sensor_msgs::PointCloud2>();
    cloud_->height = 0;
    cloud_->width = 0;
}

long ITracer::getGeometryCount() const { return geometries_; }
void ITracer::setGeometryCount(long n) { geometries_ = n; }
ITracer::CloudPtr ITracer::getTraceCloud() { return cloud_; }
void ITracer::setTraceCloud(CloudPtr storage) { cloud_ = storage; }
LidarDevice::Ptr ITracer::getSensorConfig() { return sensor_; }
void ITracer::setSensorConfig(LidarDevice::Ptr sensor) { sensor_ = sensor; }

}  // namespace lidarshooter
