// LidarDevice.hpp -- host mirror of lidarshooter::LidarDevice (LidarDevice.hpp:55-480) for the
// tracer hot path: sensor JSON -> channel / azimuth tables, sensor pose, PointCloud2 header.
//
// Kept: create/initialize (LidarDevice.cpp:23-92), loadConfiguration (:482-633),
//       loadTransformationFromFile/Json (:720-822), initMessage (:94-115), originToSensor[Inverse]
//       (:383-401), getTotalRays/getTotalChannels (:411-419), reset/getCurrentIndex (:403-409).
// Moved to the GPU: nextRay1/4/8/16, allRays, allRaysGPU (:160-381, LidarDeviceKernels.cu) -- ray
//       generation is fused into the trace kernel; `rayDirection` below is the scalar formula for
//       host-side checks.
// Out of scope: loadTransformationFromUrl (HTTP via Poco, :635-718).
#pragma once

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lidarshooter_hip.h"
#include "HostTypes.hpp"
#include "Json.hpp"

namespace lidarshooter {

class LidarDevice : public std::enable_shared_from_this<LidarDevice> {
public:
    using Ptr = std::shared_ptr<LidarDevice>;

    static Ptr create() { return Ptr(new LidarDevice()); }
    static Ptr create(const std::string& _config) { return Ptr(new LidarDevice(_config, "")); }
    static Ptr create(const std::string& _config, const std::string& _sensorUid) { return Ptr(new LidarDevice(_config, _sensorUid)); }
    Ptr getPtr() { return shared_from_this(); }

    void initialize(const std::string& _config) { initialize(_config, ""); }
    void initialize(const std::string& _config, const std::string& _sensorUid);

    // LidarDevice.cpp:94-115
    void initMessage(PointCloud2& _msg, int _frameIndex) const;

    // LidarDevice.cpp:383-401 (row-major R / Rinv applied to a 3-vector in place)
    void originToSensor(float _sensor[3]) const;
    void originToSensorInverse(float _sensor[3]) const;

    void reset() { _verticalIndex = 0; _horizontalIndex = 0; }
    void getCurrentIndex(int* _v, int* _h) const { *_v = static_cast<int>(_verticalIndex); *_h = static_cast<int>(_horizontalIndex); }
    unsigned int getTotalRays() const { return _channels.count; }
    unsigned int getTotalChannels() const { return static_cast<unsigned int>(_channels.vertical.size()); }
    unsigned int getScanRayCount() const { return _channels.horizontal.count; }
    const std::string& getSensorUid() const { return _device.sensorUid; }

    // direction of ray (channel v, azimuth column h): LidarDevice.cpp:306-316
    void rayDirection(unsigned int v, unsigned int h, float dir[3]) const;

    // What the C ABI needs (include/lidarshooter_hip.h: ls_sensor_desc); pointers stay owned here.
    ls_sensor_desc sensorDesc() const;

    const std::vector<float>& verticalAngles() const { return _channels.vertical; }
    float horizontalBegin() const { return _channels.horizontal.range.begin; }
    float horizontalEnd() const { return _channels.horizontal.range.end; }
    float horizontalStep() const { return _channels.horizontal.step; }
    const std::array<float, 9>& rotation() const { return _device.transform.sensorToBase.R; }
    const std::array<float, 9>& rotationInverse() const { return _device.transform.sensorToBase.Rinv; }
    std::array<float, 3> translation() const
    {
        return {_device.transform.baseToOrigin.tx, _device.transform.baseToOrigin.ty, _device.transform.sensorToBase.tz};
    }

private:
    LidarDevice() = default;
    LidarDevice(const std::string& _config, const std::string& _sensorUid) { initialize(_config, _sensorUid); }

    int loadConfiguration(const std::string& _config, const std::string& _sensorUid);
    int loadTransformationFromFile(const std::string& _transformFile);
    int loadTransformationFromJson(const json::Value& _transformJson);

    struct {
        std::string sensorUid, sensorApiUrl;
        unsigned int sensorApiPort = 0;
        struct {
            struct { float tx = 0.f, ty = 0.f; } baseToOrigin;
            struct {
                float qw = 0.f, qx = 0.f, qy = 0.f, qz = 0.f, tz = 0.f;
                std::array<float, 9> R{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, Rinv{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
            } sensorToBase;
        } transform;
    } _device;

    struct {
        std::string frameId = "PandarXT-32";
        int pointStep = 32;
        bool isBigendian = false, isDense = true;
        std::vector<PointField> fields;
    } _message;

    struct {
        std::vector<float> vertical;
        struct {
            struct { float begin = 0.f, end = 360.f; } range;
            unsigned int count = 0;
            float step = 0.f;
        } horizontal;
        unsigned int count = 0;
    } _channels;

    std::string _outputFolder = ".";
    unsigned int _verticalIndex = 0, _horizontalIndex = 0;
    bool _configLoaded = false;
};

}  // namespace lidarshooter
