// stub of src/LidarShooter.hpp.in: only what ITracer / the adapter reference
#pragma once
#define LIDARSHOOTER_APPLICATION_NAME "LiDARShooter"
