// stub of <spdlog/spdlog.h>: a logger that drops everything
#pragma once
#include <memory>
#include <string>
namespace spdlog {
class logger {
public:
    template <typename... A> void debug(const char*, A&&...) {}
    template <typename... A> void info(const char*, A&&...) {}
    template <typename... A> void warn(const char*, A&&...) {}
    template <typename... A> void error(const char*, A&&...) {}
};
inline std::shared_ptr<logger> get(const std::string&) { return nullptr; }
inline std::shared_ptr<logger> stdout_color_mt(const std::string&) { return std::make_shared<logger>(); }
}  // namespace spdlog
