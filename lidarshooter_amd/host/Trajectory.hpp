// Trajectory.hpp -- player for config/trajectory.json.
//
// The reference ships config/trajectory.json ({dt, linear[3], angular[3]} segments, "can be converted
// to bag format for streaming to the mesh projector") but no consumer.  The player below feeds the
// segments through the reference's own pose integration rule, AffineMesh::joystickCallback
// (AffineMesh.cpp:107-128) with AffineMesh::transformToGlobal (:275-284):
//     linear  += (Rz(ang.z) * Ry(ang.y) * Rx(ang.x)) * twist.linear        (rotation from the CURRENT angles)
//     angular += twist.angular
// one twist message per trace period (0.1 s, mainwindow.cpp:269-270), `dt / period` messages per
// segment.  The resulting (linear, angular) pairs are exactly what MeshProjector hands to
// ITracer::updateGeometry(name, translation, rotation, mesh) every frame (MeshProjector.cpp:451-456).
#pragma once

#include <array>
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "Json.hpp"

namespace lidarshooter {

struct Twist {
    float dt = 0.f;
    std::array<float, 3> linear{{0, 0, 0}}, angular{{0, 0, 0}};
};

struct Pose {
    std::array<float, 3> linear{{0, 0, 0}}, angular{{0, 0, 0}};
};

class Trajectory {
public:
    static Trajectory load(const std::string& path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("trajectory file not found: " + path);
        std::ostringstream ss;
        ss << f.rdbuf();
        const json::Value j = json::parse(ss.str());
        Trajectory t;
        for (const json::Value& seg : j["trajectory"].items()) {
            Twist tw;
            tw.dt = seg.getFloat("dt", 0.0f);
            for (int i = 0; i < 3; ++i) {
                tw.linear[i] = seg["linear"][static_cast<std::size_t>(i)].asFloat();
                tw.angular[i] = seg["angular"][static_cast<std::size_t>(i)].asFloat();
            }
            t._segments.push_back(tw);
        }
        return t;
    }

    const std::vector<Twist>& segments() const { return _segments; }

    // AffineMesh::joystickCallback: one twist message applied to the accumulated displacement
    static void applyTwist(Pose& p, const std::array<float, 3>& lin, const std::array<float, 3>& ang)
    {
        // transformToGlobal: Eigen ((Translation(0)*Rz)*Ry)*Rx with AngleAxis::toRotationMatrix, then R*v
        float rx[9], ry[9], rz[9], zy[9], r[9];
        axisRotation(p.angular[0], 0, rx);
        axisRotation(p.angular[1], 1, ry);
        axisRotation(p.angular[2], 2, rz);
        mul3(rz, ry, zy);
        mul3(zy, rx, r);
        for (int i = 0; i < 3; ++i) {
            const float g = ((r[3 * i] * lin[0] + r[3 * i + 1] * lin[1]) + r[3 * i + 2] * lin[2]) + 0.0f;
            p.linear[i] += g;
        }
        for (int i = 0; i < 3; ++i) p.angular[i] += ang[i];
    }

    // poses after each twist message, `period` seconds apart (the first pose is after the first message)
    std::vector<Pose> play(float period = 0.1f) const
    {
        std::vector<Pose> out;
        Pose p;
        for (const Twist& tw : _segments) {
            const long n = std::lround(static_cast<double>(tw.dt) / static_cast<double>(period));
            for (long k = 0; k < n; ++k) {
                applyTwist(p, tw.linear, tw.angular);
                out.push_back(p);
            }
        }
        return out;
    }

private:
    static void axisRotation(float angle, int axis, float* m)
    {
        // Eigen AngleAxis::toRotationMatrix for a unit axis (same expressions as ls_tracer.cpp)
        float ax[3] = {0.f, 0.f, 0.f};
        ax[axis] = 1.0f;
        const float s = std::sin(angle), c = std::cos(angle);
        const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
        const float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
        float tmp;
        tmp = ca[0] * ax[1]; m[1] = tmp - sa[2]; m[3] = tmp + sa[2];
        tmp = ca[0] * ax[2]; m[2] = tmp + sa[1]; m[6] = tmp - sa[1];
        tmp = ca[1] * ax[2]; m[5] = tmp - sa[0]; m[7] = tmp + sa[0];
        m[0] = ca[0] * ax[0] + c;
        m[4] = ca[1] * ax[1] + c;
        m[8] = ca[2] * ax[2] + c;
    }
    static void mul3(const float* a, const float* b, float* o)
    {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                o[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
    }

    std::vector<Twist> _segments;
};

}  // namespace lidarshooter
