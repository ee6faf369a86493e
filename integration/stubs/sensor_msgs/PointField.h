// stub of sensor_msgs/PointField.h
#pragma once
#include <cstdint>
#include <string>
namespace sensor_msgs {
struct PointField {
    std::string name;
    std::uint32_t offset = 0;
    std::uint8_t datatype = 0;
    std::uint32_t count = 0;
};
}  // namespace sensor_msgs
