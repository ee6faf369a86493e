// stub with the shape of lidarshooter's src/ITracer.hpp:29-152 -- the abstract tracer the adapter derives from.
// Declarations only mirror what a backend must override or may call; see the reference for the real file.
#pragma once

#include <cstdint>
#include <memory>
#include <string>

#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <pcl/io/vtk_lib_io.h>
#include <sensor_msgs/PointCloud2.h>
#include <spdlog/spdlog.h>

#include "LidarDevice.hpp"
#include "LidarShooter.hpp"

namespace lidarshooter
{

class ITracer
{
public:
    using Ptr = std::shared_ptr<ITracer>;
    using ConstPtr = std::shared_ptr<ITracer const>;

    ITracer(LidarDevice::Ptr _sensorConfig, sensor_msgs::PointCloud2::Ptr _traceStorage = nullptr, std::shared_ptr<spdlog::logger> _logger = nullptr);
    virtual ITracer::Ptr getPtr() = 0;
    virtual ~ITracer() = default;

    virtual int addGeometry(const std::string& _meshName, enum RTCGeometryType _geometryType, int _numVertices, int _numElements) = 0;
    virtual int removeGeometry(const std::string& _meshName) = 0;
    virtual int updateGeometry(const std::string& _meshName, Eigen::Affine3f _transform, pcl::PolygonMesh::Ptr& _mesh) = 0;
    virtual int updateGeometry(const std::string& _meshName, Eigen::Vector3f _translation, Eigen::Vector3f _rotation, pcl::PolygonMesh::Ptr& _mesh) = 0;
    virtual int commitScene() = 0;
    virtual int traceScene(std::uint32_t _frameIndex) = 0;
    virtual long getGeometryCount() const;

    sensor_msgs::PointCloud2::Ptr getTraceCloud();
    void setTraceCloud(sensor_msgs::PointCloud2::Ptr _traceStorage);
    LidarDevice::Ptr getSensorConfig();
    void setSensorConfig(LidarDevice::Ptr __config);

protected:
    void setGeometryCount(long _count);
    std::shared_ptr<spdlog::logger> _logger;

private:
    LidarDevice::Ptr _config;
    long _geometryCount;
    sensor_msgs::PointCloud2::Ptr _traceCloud;
};

}  // namespace lidarshooter
