// stub of src/Exceptions.hpp:160-176: the one exception type the tracer backends throw
#pragma once
#include <stdexcept>
#include <string>

namespace lidarshooter
{

class TraceException : public std::runtime_error
{
public:
    TraceException(std::string _errorLocation, std::string _errorString, long _errorCode)
        : std::runtime_error("Trace error: " + _errorString + " in " + _errorLocation + " (code " + std::to_string(_errorCode) + ")"),
          _code(_errorCode)
    {
    }
    long getErrorCode() const { return _code; }
    std::string getError() const { return what(); }

private:
    long _code;
};

}  // namespace lidarshooter
