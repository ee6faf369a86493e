// LidarDevice.cpp -- see LidarDevice.hpp.  All float arithmetic follows the reference's
// expressions term by term (cited per function) so that the tables handed to the GPU are the
// ones the reference's CPU path would compute.
#include "LidarDevice.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace lidarshooter {

namespace {

std::string slurp(const std::string& path, bool& ok)
{
    std::ifstream f(path, std::ios::binary);
    ok = static_cast<bool>(f);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// Eigen/src/Geometry/Quaternion.h QuaternionBase::toRotationMatrix (float), row-major out.
void quatToRotation(float w, float x, float y, float z, std::array<float, 9>& R)
{
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R = {{1.0f - (tyy + tzz), txy - twz, txz + twy,
          txy + twz, 1.0f - (txx + tzz), tyz - twx,
          txz - twy, tyz + twx, 1.0f - (txx + tyy)}};
}

// Eigen/src/LU/InverseImpl.h compute_inverse for 3x3: cofactors / determinant (float).
void inverse3(const std::array<float, 9>& m, std::array<float, 9>& inv)
{
    auto M = [&](int i, int j) { return m[3 * i + j]; };
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    };
    const float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const float det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
    const float invdet = 1.0f / det;
    inv = {{c00 * invdet, c10 * invdet, c20 * invdet,
            cof(0, 1) * invdet, cof(1, 1) * invdet, cof(2, 1) * invdet,
            cof(0, 2) * invdet, cof(1, 2) * invdet, cof(2, 2) * invdet}};
}

}  // namespace

void LidarDevice::initialize(const std::string& _config, const std::string& _sensorUid)
{
    // LidarDevice.cpp:83-92
    _channels.count = 0;
    if (!_config.empty()) loadConfiguration(_config, _sensorUid);
    reset();
}

int LidarDevice::loadConfiguration(const std::string& _config, const std::string& _sensorUid)
{
    // LidarDevice.cpp:482-633
    bool ok = false;
    const std::string text = slurp(_config, ok);
    if (!ok) throw ConfigurationException(_config, __FILE__, "File not found", 1);
    const json::Value jsonData = json::parse(text);

    if (jsonData.isMember("device")) {
        const json::Value& dev = jsonData["device"];
        if (!_sensorUid.empty()) _device.sensorUid = _sensorUid;
        else if (dev.isMember("sensorUid")) _device.sensorUid = dev.getString("sensorUid", "");

        if (dev.isMember("sensorConfig")) {
            loadTransformationFromJson(dev["sensorConfig"]);  // inline pose has the highest precedence (:505-510)
        } else if (dev.isMember("sensorConfigFile")) {
            const std::string fullPath = dev.getString("sensorConfigFile", "transform-device.json");
            bool exists = false;
            (void)slurp(fullPath, exists);
            if (!exists) throw ConfigurationException(fullPath, __FILE__, "File not found", 1);  // :518-527
            loadTransformationFromFile(fullPath);
        } else {
            // :533-554 fetches the pose from a SENSR REST endpoint over HTTP: out of scope (no network)
            _device.sensorUid = dev.getString("sensorUid", "lidar_0000");
            _device.sensorApiUrl = dev.getString("sensorApiUrl", "localhost");
            _device.sensorApiPort = dev.getUInt("sensorApiPort", 9080);
            if (!_sensorUid.empty()) _device.sensorUid = _sensorUid;
            throw ConfigurationException(_config, __FILE__, "sensor pose over HTTP is not supported by this backend", 2);
        }
    }

    if (jsonData.isMember("message")) {
        const json::Value& msg = jsonData["message"];
        _message.frameId = msg.getString("frameId", "PandarXT-32");
        _message.pointStep = msg.getInt("pointStep", 32);
        _message.isBigendian = msg.getBool("isBigendian", false);
        _message.isDense = msg.getBool("isDense", true);
    }
    _message.fields.clear();
    if (jsonData.isMember("message") && jsonData["message"].isMember("pointFields")) {
        for (const json::Value& field : jsonData["message"]["pointFields"].items()) {
            PointField pf;
            pf.name = field["name"].asString();
            pf.offset = static_cast<std::uint32_t>(field["offset"].asInt());
            pf.datatype = static_cast<std::uint8_t>(field["datatype"].asInt());
            pf.count = static_cast<std::uint32_t>(field["count"].asInt());
            _message.fields.push_back(pf);
        }
    }

    if (jsonData.isMember("channels")) {
        const json::Value& ch = jsonData["channels"];
        _channels.vertical.clear();
        if (ch.isMember("vertical"))
            for (const json::Value& angle : ch["vertical"].items()) _channels.vertical.push_back(angle.asFloat());
        if (ch.isMember("horizontal")) {
            const json::Value& hz = ch["horizontal"];
            if (hz.isMember("range")) {
                _channels.horizontal.range.begin = hz["range"].getFloat("begin", 0.0f);
                _channels.horizontal.range.end = hz["range"].getFloat("end", 360.0f);
            }
            _channels.horizontal.count = hz.getUInt("count", 128);
            // :611  float / (unsigned - 1) -> float
            _channels.horizontal.step = (_channels.horizontal.range.end - _channels.horizontal.range.begin) /
                                        static_cast<float>(_channels.horizontal.count - 1u);
        }
        _channels.count = static_cast<unsigned int>(_channels.vertical.size()) * _channels.horizontal.count;  // :617
    }

    if (jsonData.isMember("outputFolder")) _outputFolder = jsonData.getString("outputFolder", ".");
    _configLoaded = true;
    return 0;
}

int LidarDevice::loadTransformationFromFile(const std::string& _transformFile)
{
    bool ok = false;
    const std::string text = slurp(_transformFile, ok);
    if (!ok) return -1;
    return loadTransformationFromJson(json::parse(text));
}

int LidarDevice::loadTransformationFromJson(const json::Value& _transformJson)
{
    // LidarDevice.cpp:758-822
    if (_transformJson.isMember("uid")) {
        const std::string uidValue = _transformJson.getString("uid", "");
        if (_device.sensorUid.empty() && !uidValue.empty()) _device.sensorUid = uidValue;
    }
    if (_transformJson.isMember("base_to_origin")) {
        _device.transform.baseToOrigin.tx = _transformJson["base_to_origin"].getFloat("tx", 0.0f);
        _device.transform.baseToOrigin.ty = _transformJson["base_to_origin"].getFloat("ty", 0.0f);
    }
    if (_transformJson.isMember("sensor_to_base")) {
        auto& s2b = _device.transform.sensorToBase;
        const json::Value& j = _transformJson["sensor_to_base"];
        s2b.qw = j.getFloat("qw", 0.0f);
        s2b.qx = j.getFloat("qx", 0.0f);
        s2b.qy = j.getFloat("qy", 0.0f);
        s2b.qz = j.getFloat("qz", 0.0f);
        s2b.tz = j.getFloat("tz", 0.0f);
        quatToRotation(s2b.qw, s2b.qx, s2b.qy, s2b.qz, s2b.R);  // :812, quaternion NOT normalised
        inverse3(s2b.R, s2b.Rinv);                               // :813
    }
    return 0;
}

void LidarDevice::initMessage(PointCloud2& _msg, int _frameIndex) const
{
    // LidarDevice.cpp:94-115 (stamp: ros::Time::now() in the reference; left to the caller here)
    _msg.fields = _message.fields;
    _msg.header.frame_id = _message.frameId;
    _msg.header.seq = static_cast<std::uint32_t>(_frameIndex);
    _msg.height = 1;
    _msg.width = 0;
    _msg.point_step = static_cast<std::uint32_t>(_message.pointStep);
    _msg.row_step = 0;
    _msg.is_bigendian = _message.isBigendian;
    _msg.is_dense = _message.isDense;
}

void LidarDevice::originToSensor(float s[3]) const
{
    const auto& Ri = _device.transform.sensorToBase.Rinv;
    const float a = s[0] - _device.transform.baseToOrigin.tx, b = s[1] - _device.transform.baseToOrigin.ty,
                c = s[2] - _device.transform.sensorToBase.tz;
    for (int i = 0; i < 3; ++i) s[i] = (Ri[3 * i] * a + Ri[3 * i + 1] * b) + Ri[3 * i + 2] * c;
}

void LidarDevice::originToSensorInverse(float s[3]) const
{
    const auto& R = _device.transform.sensorToBase.R;
    const float a = s[0], b = s[1], c = s[2];
    const float tr[3] = {_device.transform.baseToOrigin.tx, _device.transform.baseToOrigin.ty, _device.transform.sensorToBase.tz};
    for (int i = 0; i < 3; ++i) s[i] = ((R[3 * i] * a + R[3 * i + 1] * b) + R[3 * i + 2] * c) + tr[i];
}

void LidarDevice::rayDirection(unsigned int v, unsigned int h, float dir[3]) const
{
    // LidarDevice.cpp:306-316
    const float preChi = _channels.vertical[v];
    const float prePhi = _channels.horizontal.range.begin + _channels.horizontal.step * static_cast<float>(h);
    const float theta = static_cast<float>((90.0 - preChi) * M_PI / 180.0);
    const float phi = static_cast<float>(prePhi * M_PI / 180.0);
    dir[0] = std::sin(theta) * std::cos(phi);
    dir[1] = std::sin(theta) * std::sin(phi);
    dir[2] = std::cos(theta);
}

ls_sensor_desc LidarDevice::sensorDesc() const
{
    ls_sensor_desc d;
    std::memset(&d, 0, sizeof(d));
    d.vertical_deg = _channels.vertical.data();
    d.n_vertical = static_cast<uint32_t>(_channels.vertical.size());
    d.h_begin = _channels.horizontal.range.begin;
    d.h_end = _channels.horizontal.range.end;
    d.h_count = _channels.horizontal.count;
    for (int i = 0; i < 9; ++i) d.Rinv[i] = _device.transform.sensorToBase.Rinv[i];
    d.t[0] = _device.transform.baseToOrigin.tx;
    d.t[1] = _device.transform.baseToOrigin.ty;
    d.t[2] = _device.transform.sensorToBase.tz;
    return d;
}

// ---------------------------------------------------------------------------------------------
int loadPolygonFileSTL(const std::string& path, PolygonMesh& mesh)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return 0;
    char header[80];
    std::uint32_t nt = 0;
    f.read(header, 80);
    f.read(reinterpret_cast<char*>(&nt), 4);
    if (!f) return 0;
    std::vector<char> body(static_cast<std::size_t>(nt) * 50);
    f.read(body.data(), static_cast<std::streamsize>(body.size()));
    if (static_cast<std::size_t>(f.gcount()) != body.size()) return 0;

    struct Key {
        std::uint32_t a, b, c;
        bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; }
    };
    struct KeyHash {
        std::size_t operator()(const Key& k) const { return (static_cast<std::size_t>(k.a) * 0x9E3779B97F4A7C15ull) ^ (static_cast<std::size_t>(k.b) << 21) ^ (static_cast<std::size_t>(k.c) << 42) ^ k.c; }
    };
    std::unordered_map<Key, std::uint32_t, KeyHash> seen;
    std::vector<float> pts;
    mesh.polygons.clear();
    mesh.polygons.reserve(nt);
    for (std::uint32_t k = 0; k < nt; ++k) {
        Vertices poly;
        for (int c = 0; c < 3; ++c) {
            float p[3];
            std::memcpy(p, body.data() + static_cast<std::size_t>(k) * 50 + 12 + 12 * c, 12);
            Key key;
            std::memcpy(&key, p, 12);
            if (key.a == 0x80000000u) key.a = 0;  // -0.0 == +0.0
            if (key.b == 0x80000000u) key.b = 0;
            if (key.c == 0x80000000u) key.c = 0;
            auto it = seen.find(key);
            std::uint32_t idx;
            if (it == seen.end()) {
                idx = static_cast<std::uint32_t>(pts.size() / 3);
                seen.emplace(key, idx);
                pts.insert(pts.end(), p, p + 3);
            } else {
                idx = it->second;
            }
            poly.vertices.push_back(idx);
        }
        mesh.polygons.push_back(std::move(poly));
    }
    // pcl::PointXYZ cloud: 16 bytes per point (x, y, z, padding)
    const std::uint32_t n = static_cast<std::uint32_t>(pts.size() / 3);
    mesh.cloud = PointCloud2();
    mesh.cloud.height = 1;
    mesh.cloud.width = n;
    mesh.cloud.point_step = 16;
    mesh.cloud.row_step = 16 * n;
    mesh.cloud.is_dense = true;
    mesh.cloud.fields = {{"x", 0, 7, 1}, {"y", 4, 7, 1}, {"z", 8, 7, 1}};
    mesh.cloud.data.assign(static_cast<std::size_t>(n) * 16, 0);
    for (std::uint32_t j = 0; j < n; ++j) {
        std::memcpy(mesh.cloud.data.data() + static_cast<std::size_t>(j) * 16, &pts[3 * static_cast<std::size_t>(j)], 12);
        const float one = 1.0f;
        std::memcpy(mesh.cloud.data.data() + static_cast<std::size_t>(j) * 16 + 12, &one, 4);
    }
    return static_cast<int>(n);
}

}  // namespace lidarshooter
