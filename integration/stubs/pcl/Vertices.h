// stub of pcl/Vertices.h (PCL 1.12: Indices = std::vector<index_t>, index_t = std::int32_t)
#pragma once
#include <cstdint>
#include <vector>
namespace pcl {
using index_t = std::int32_t;
using Indices = std::vector<index_t>;
struct Vertices {
    Indices vertices;
};
}  // namespace pcl
