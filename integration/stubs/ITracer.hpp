// Stand-in for lidarshooter's src/ITracer.hpp (see the reference, lines 29-152, for the real header): only the
// SHAPE a tracer backend derives from -- six pure virtuals, the count / cloud / config accessors -- so that
// integration/HipTracer.hpp can be compiled and exercised in an image without ROS.  Test infrastructure.
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
    typedef std::shared_ptr<ITracer> Ptr;
    typedef std::shared_ptr<const ITracer> ConstPtr;
    typedef sensor_msgs::PointCloud2::Ptr CloudPtr;
    typedef std::shared_ptr<spdlog::logger> LoggerPtr;

    ITracer(LidarDevice::Ptr sensor, CloudPtr storage = nullptr, LoggerPtr logger = nullptr);
    virtual ~ITracer() = default;
    virtual Ptr getPtr() = 0;

    // geometry registry
    virtual int addGeometry(const std::string& mesh,
                            enum RTCGeometryType elementType,
                            int vertexCount,
                            int elementCount) = 0;
    virtual int removeGeometry(const std::string& mesh) = 0;
    // per frame: every mesh with its pose, then commit, then trace
    virtual int updateGeometry(const std::string& mesh,
                               Eigen::Affine3f pose,
                               pcl::PolygonMesh::Ptr& data) = 0;
    virtual int updateGeometry(const std::string& mesh,
                               Eigen::Vector3f shift,
                               Eigen::Vector3f eulerXYZ,
                               pcl::PolygonMesh::Ptr& data) = 0;
    virtual int commitScene() = 0;
    virtual int traceScene(std::uint32_t frame) = 0;

    virtual long getGeometryCount() const;
    CloudPtr getTraceCloud();
    void setTraceCloud(CloudPtr storage);
    LidarDevice::Ptr getSensorConfig();
    void setSensorConfig(LidarDevice::Ptr sensor);

protected:
    void setGeometryCount(long n);
    LoggerPtr _logger;

private:
    LidarDevice::Ptr sensor_;
    long geometries_ = 0;
    CloudPtr cloud_;
};

}  // namespace lidarshooter
