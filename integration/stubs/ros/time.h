// stub of ros/time.h
#pragma once
#include <cstdint>
namespace ros {
struct Time {
    std::uint32_t sec = 0, nsec = 0;
    static Time now() { return Time(); }
    static void init() {}
};
}  // namespace ros
