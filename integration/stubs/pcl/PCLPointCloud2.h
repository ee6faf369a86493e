// stub of pcl/PCLPointCloud2.h
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "pcl/PCLHeader.h"
namespace pcl {
struct PCLPointField {
    std::string name;
    std::uint32_t offset = 0;
    std::uint8_t datatype = 0;
    std::uint32_t count = 0;
};
struct PCLPointCloud2 {
    PCLHeader header;
    std::uint32_t height = 0, width = 0;
    std::vector<PCLPointField> fields;
    std::uint8_t is_bigendian = 0;
    std::uint32_t point_step = 0, row_step = 0;
    std::vector<std::uint8_t> data;
    std::uint8_t is_dense = 0;
};
}  // namespace pcl
