// The stub LidarDevice (integration/stubs/LidarDevice.hpp) implemented over this repository's host mirror of
// the sensor JSON parser, through liblidarshooter_host.so's C entry points (lidarshooter_amd/host/host_capi.cpp):
// same public behaviour as the reference's LidarDevice for the calls the tracer surface makes --
// nextRay1 (LidarDevice.cpp:160-198), originToSensor[Inverse] (:383-401), initMessage (:94-115),
// advanceRayIndex (:824-845).  Test infrastructure only.
#include "LidarDevice.hpp"

#include <mutex>
#include <stdexcept>

extern "C" {
struct lsh_device;
lsh_device* lsh_device_create(const char* config_path, const char* sensor_uid);
void lsh_device_destroy(lsh_device* d);
unsigned lsh_device_total_rays(lsh_device* d);
unsigned lsh_device_total_channels(lsh_device* d);
unsigned lsh_device_scan_ray_count(lsh_device* d);
const char* lsh_device_uid(lsh_device* d);
void lsh_device_ray_direction(lsh_device* d, unsigned v, unsigned h, float* dir3);
void lsh_device_origin_to_sensor(lsh_device* d, float* p3, int inverse);
void lsh_device_init_message(lsh_device* d, int frame, unsigned* out6, unsigned* flags, char* frame_id, unsigned frame_id_cap);
int lsh_device_message_field(lsh_device* d, unsigned i, char* name, unsigned name_cap, unsigned* offset, unsigned* datatype, unsigned* count);
const char* lsh_last_error(void);
}

namespace lidarshooter
{

struct LidarDevice::Impl {
    lsh_device* device = nullptr;
    std::string uid;
    unsigned vertical = 0, horizontal = 0;   // iterator state of nextRay*
    std::mutex rayMutex;
};

LidarDevice::Ptr LidarDevice::create(const std::string& _config, std::shared_ptr<spdlog::logger>) { return Ptr(new LidarDevice(_config, "")); }
LidarDevice::Ptr LidarDevice::create(const std::string& _config, const std::string& _sensorUid, std::shared_ptr<spdlog::logger>)
{
    return Ptr(new LidarDevice(_config, _sensorUid));
}
LidarDevice::Ptr LidarDevice::getPtr() { return shared_from_this(); }

LidarDevice::LidarDevice(const std::string& _config, const std::string& _sensorUid) : _impl(new Impl())
{
    _impl->device = lsh_device_create(_config.c_str(), _sensorUid.c_str());
    if (!_impl->device) throw std::runtime_error(std::string("stub LidarDevice: ") + lsh_last_error());
    _impl->uid = lsh_device_uid(_impl->device);
}

void LidarDevice::initialize(const std::string& _config) { initialize(_config, ""); }
void LidarDevice::initialize(const std::string& _config, const std::string& _sensorUid)
{
    lsh_device* fresh = lsh_device_create(_config.c_str(), _sensorUid.c_str());
    if (!fresh) throw std::runtime_error(std::string("stub LidarDevice: ") + lsh_last_error());
    if (_impl->device) lsh_device_destroy(_impl->device);
    _impl->device = fresh;
    _impl->uid = lsh_device_uid(fresh);
    _impl->vertical = _impl->horizontal = 0;
}

LidarDevice::~LidarDevice()
{
    if (_impl && _impl->device) lsh_device_destroy(_impl->device);
}

void LidarDevice::initMessage(sensor_msgs::PointCloud2Ptr _msg, int _frameIndex)
{
    unsigned head[6], flags = 0;
    char frameId[128];
    lsh_device_init_message(_impl->device, _frameIndex, head, &flags, frameId, sizeof(frameId));
    _msg->fields.clear();
    for (unsigned i = 0; i < head[5]; ++i) {
        char name[64];
        unsigned offset = 0, datatype = 0, count = 0;
        lsh_device_message_field(_impl->device, i, name, sizeof(name), &offset, &datatype, &count);
        sensor_msgs::PointField field;
        field.name = name;
        field.offset = offset;
        field.datatype = static_cast<std::uint8_t>(datatype);
        field.count = count;
        _msg->fields.push_back(field);
    }
    _msg->header.frame_id = frameId;
    _msg->header.stamp = ros::Time::now();
    _msg->header.seq = head[0];
    _msg->height = head[1];
    _msg->width = head[2];
    _msg->point_step = head[3];
    _msg->row_step = head[4];
    _msg->is_bigendian = (flags & 1u) != 0;
    _msg->is_dense = (flags & 2u) != 0;
}

int LidarDevice::nextRay1(RTCRayHit& _ray, int*)
{
    std::lock_guard<std::mutex> lock(_impl->rayMutex);
    float dir[3];
    lsh_device_ray_direction(_impl->device, _impl->vertical, _impl->horizontal, dir);
    _ray.ray.org_x = _ray.ray.org_y = _ray.ray.org_z = 0.f;
    _ray.ray.dir_x = dir[0];
    _ray.ray.dir_y = dir[1];
    _ray.ray.dir_z = dir[2];
    _ray.ray.tnear = 0.f;
    _ray.ray.tfar = __builtin_inff();
    _ray.hit.geomID = RTC_INVALID_GEOMETRY_ID;
    // channel-major walk; 1 once the last ray has been handed out
    if (_impl->horizontal + 1 == getScanRayCount()) {
        _impl->horizontal = 0;
        if (_impl->vertical + 1 == getTotalChannels()) {
            _impl->vertical = 0;
            return 1;
        }
        ++_impl->vertical;
    } else {
        ++_impl->horizontal;
    }
    return 0;
}

void LidarDevice::originToSensor(Eigen::Vector3f& _sensor) const { lsh_device_origin_to_sensor(_impl->device, _sensor.data(), 0); }
void LidarDevice::originToSensorInverse(Eigen::Vector3f& _sensor) const { lsh_device_origin_to_sensor(_impl->device, _sensor.data(), 1); }
void LidarDevice::reset() { _impl->vertical = _impl->horizontal = 0; }
unsigned int LidarDevice::getTotalRays() { return lsh_device_total_rays(_impl->device); }
unsigned int LidarDevice::getTotalChannels() { return lsh_device_total_channels(_impl->device); }
unsigned int LidarDevice::getScanRayCount() { return lsh_device_scan_ray_count(_impl->device); }
void LidarDevice::getCurrentIndex(int* _verticalIndex, int* _horizontalIndex)
{
    *_verticalIndex = static_cast<int>(_impl->vertical);
    *_horizontalIndex = static_cast<int>(_impl->horizontal);
}
const std::string& LidarDevice::getSensorUid() const { return _impl->uid; }

}  // namespace lidarshooter
