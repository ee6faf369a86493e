// HostTypes.hpp -- plain stand-ins for the ROS / PCL message types that cross the tracer surface.
//
// The reference's ITracer takes pcl::PolygonMesh and fills a sensor_msgs::PointCloud2
// (ITracer.hpp:69-94).  Neither ROS nor PCL exists in this image, so the host mirror carries the
// same FIELDS in its own structs (same names, same meaning); the ROS-side adapter in INTEGRATION.md
// passes the real messages' buffers straight to the C ABI instead.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace lidarshooter {

// sensor_msgs/PointField
struct PointField {
    std::string name;
    std::uint32_t offset = 0;
    std::uint8_t datatype = 0;
    std::uint32_t count = 0;
};

// the part of sensor_msgs/PointCloud2 the path touches (LidarDevice.cpp:94-115)
struct PointCloud2 {
    struct {
        std::uint32_t seq = 0;
        double stamp = 0.0;
        std::string frame_id;
    } header;
    std::uint32_t height = 0, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    std::uint32_t point_step = 0, row_step = 0;
    std::vector<std::uint8_t> data;
    bool is_dense = false;
};

// pcl::Vertices / pcl::PolygonMesh: a PointCloud2-style vertex blob + polygons
struct Vertices {
    std::vector<std::uint32_t> vertices;
};

struct PolygonMesh {
    PointCloud2 cloud;  // point_step bytes per vertex, x,y,z float32 first (pcl::PointXYZ: 16)
    std::vector<Vertices> polygons;
};

// Exceptions.hpp:160-176 TraceException(location, message, code)
class TraceException : public std::runtime_error {
public:
    TraceException(const std::string& location, const std::string& message, long code)
        : std::runtime_error("Trace error: " + message + " in " + location + " (code " + std::to_string(code) + ")"),
          _code(code)
    {
    }
    long getErrorCode() const { return _code; }

private:
    long _code;
};

// Exceptions.hpp ConfigurationException
class ConfigurationException : public std::runtime_error {
public:
    ConfigurationException(const std::string& file, const std::string& location, const std::string& message, long code)
        : std::runtime_error("Configuration error: " + message + " (" + file + ") in " + location + " (code " +
                             std::to_string(code) + ")")
    {
    }
};

// pcl::io::loadPolygonFileSTL (test/EmbreeTracer_test.cpp:42-45): binary STL, exactly-equal
// vertices merged in first-seen order, triangle order kept (pinned by 98 vertices / 162 triangles
// for mesh/ground.stl, EmbreeTracer_test.cpp:86-91).  Returns the number of points.
int loadPolygonFileSTL(const std::string& path, PolygonMesh& mesh);

}  // namespace lidarshooter
