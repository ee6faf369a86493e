// stub: ITracer.hpp includes it; nothing of it is used on the path
#pragma once
#include "pcl/PolygonMesh.h"
