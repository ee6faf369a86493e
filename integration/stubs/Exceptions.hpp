// stub of src/Exceptions.hpp:160-176: the one exception type the tracer backends throw
#pragma once
#include <embree3/rtcore.h>
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

// stub of src/Exceptions.hpp:143-157: thrown by the adapter when the library refuses a mesh (LS_ERR_OUT_OF_RANGE from a commit:
// a triangle names a vertex the geometry does not have)
class BadGeometryException : public std::runtime_error
{
public:
    BadGeometryException(std::string _errorLocation, std::string _errorString, long _errorCode, RTCGeometryType _geometryType)
        : std::runtime_error("BadGeometry error (type " + std::to_string(static_cast<int>(_geometryType)) + "): " + _errorString + " in " + _errorLocation +
                             " (code " + std::to_string(_errorCode) + ")"),
          _code(_errorCode)
    {
    }
    long getErrorCode() const { return _code; }
    std::string getError() const { return what(); }

private:
    long _code;
};

}  // namespace lidarshooter
