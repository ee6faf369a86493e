// stub with the PUBLIC shape of lidarshooter's src/LidarDevice.hpp:55-300 (nothing private is declared the
// same way, on purpose: the adapter may only use what the reference exposes).  Implemented in
// StubLidarDevice.cpp over this repository's host mirror of the sensor JSON parser.
#pragma once

#include <memory>
#include <string>

#include <Eigen/Dense>
#include <embree3/rtcore.h>
#include <sensor_msgs/PointCloud2.h>
#include <spdlog/spdlog.h>

#include "LidarShooter.hpp"

namespace lidarshooter
{

class LidarDevice : public std::enable_shared_from_this<LidarDevice>
{
public:
    using Ptr = std::shared_ptr<LidarDevice>;
    using ConstPtr = std::shared_ptr<LidarDevice const>;

    static Ptr create(const std::string& _config, std::shared_ptr<spdlog::logger> __logger = nullptr);
    static Ptr create(const std::string& _config, const std::string& _sensorUid, std::shared_ptr<spdlog::logger> __logger = nullptr);
    Ptr getPtr();
    ~LidarDevice();

    void initialize(const std::string& _config);                                // (LidarDevice.hpp:116-117: the same object, another configuration)
    void initialize(const std::string& _config, const std::string& _sensorUid);
    void initMessage(sensor_msgs::PointCloud2Ptr _msg, int _frameIndex);
    int nextRay1(RTCRayHit& _ray, int* _valid);
    void originToSensor(Eigen::Vector3f& _sensor) const;
    void originToSensorInverse(Eigen::Vector3f& _sensor) const;
    void reset();
    unsigned int getTotalRays();
    unsigned int getTotalChannels();
    unsigned int getScanRayCount();
    void getCurrentIndex(int* _verticalIndex, int* _horizontalIndex);
    const std::string& getSensorUid() const;

private:
    LidarDevice(const std::string& _config, const std::string& _sensorUid);
    struct Impl;
    std::unique_ptr<Impl> _impl;
};

}  // namespace lidarshooter
