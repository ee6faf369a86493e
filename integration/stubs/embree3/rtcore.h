// stub of <embree3/rtcore.h>: the enum in ITracer::addGeometry's signature and the single-ray struct of nextRay1
#pragma once
#define RTC_INVALID_GEOMETRY_ID (static_cast<unsigned int>(-1))
enum RTCGeometryType {
    RTC_GEOMETRY_TYPE_TRIANGLE = 0,
    RTC_GEOMETRY_TYPE_QUAD = 1,
    RTC_GEOMETRY_TYPE_GRID = 2,
    RTC_GEOMETRY_TYPE_SUBDIVISION = 8
};
struct RTCRay {
    float org_x, org_y, org_z, tnear;
    float dir_x, dir_y, dir_z, time;
    float tfar;
    unsigned int mask, id, flags;
};
struct RTCHit {
    float Ng_x, Ng_y, Ng_z, u, v;
    unsigned int primID, geomID, instID[1];
};
struct RTCRayHit {
    RTCRay ray;
    RTCHit hit;
};
