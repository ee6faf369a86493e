// stub of pcl/PolygonMesh.h
#pragma once
#include <memory>
#include <vector>
#include "pcl/PCLPointCloud2.h"
#include "pcl/Vertices.h"
namespace pcl {
struct PolygonMesh {
    using Ptr = std::shared_ptr<PolygonMesh>;
    using ConstPtr = std::shared_ptr<PolygonMesh const>;
    PCLHeader header;
    PCLPointCloud2 cloud;
    std::vector<Vertices> polygons;
};
}  // namespace pcl
