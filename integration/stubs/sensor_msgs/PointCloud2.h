// stub of sensor_msgs/PointCloud2.h (ROS 1: Ptr is boost::shared_ptr there; the adapter only uses the alias)
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include "sensor_msgs/PointField.h"
#include "std_msgs/Header.h"
namespace sensor_msgs {
struct PointCloud2 {
    using Ptr = std::shared_ptr<PointCloud2>;
    using ConstPtr = std::shared_ptr<PointCloud2 const>;
    std_msgs::Header header;
    std::uint32_t height = 0, width = 0;
    std::vector<PointField> fields;
    std::uint8_t is_bigendian = 0;
    std::uint32_t point_step = 0, row_step = 0;
    std::vector<std::uint8_t> data;
    std::uint8_t is_dense = 0;
};
using PointCloud2Ptr = PointCloud2::Ptr;
using PointCloud2ConstPtr = PointCloud2::ConstPtr;
}  // namespace sensor_msgs
