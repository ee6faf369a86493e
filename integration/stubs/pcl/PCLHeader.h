// stub of pcl/PCLHeader.h
#pragma once
#include <cstdint>
#include <string>
namespace pcl {
struct PCLHeader {
    std::uint32_t seq = 0;
    std::uint64_t stamp = 0;
    std::string frame_id;
};
}  // namespace pcl
