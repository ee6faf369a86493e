// Json.hpp -- small tolerant JSON reader for the sensor configuration files.
//
// The reference parses its configs with jsoncpp with comments allowed (LidarDevice.cpp:485-493).
// The shipped files contain `//` comment lines AND a "http://..." string
// (config/hesai-pandar-XT-32-lidar_0000.json:4,78,92), so comments are skipped only outside string
// literals.  jsoncpp is not available in this image; this reader covers what those files use.
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lidarshooter {
namespace json {

class Value {
public:
    enum class Type { Null, Bool, Number, String, Array, Object };

    Value() = default;

    Type type() const { return _type; }
    bool isNull() const { return _type == Type::Null; }
    bool isObject() const { return _type == Type::Object; }
    bool isArray() const { return _type == Type::Array; }

    // jsoncpp-like accessors
    bool isMember(const std::string& key) const { return _type == Type::Object && _object.count(key) > 0; }
    const Value& operator[](const std::string& key) const
    {
        static const Value null;
        if (_type != Type::Object) return null;
        auto it = _object.find(key);
        return it == _object.end() ? null : it->second;
    }
    const Value& operator[](std::size_t idx) const
    {
        static const Value null;
        return (_type == Type::Array && idx < _array.size()) ? _array[idx] : null;
    }
    std::size_t size() const { return _type == Type::Array ? _array.size() : (_type == Type::Object ? _object.size() : 0); }
    const std::vector<Value>& items() const { return _array; }

    double asDouble(double dflt = 0.0) const { return _type == Type::Number ? _number : (_type == Type::Bool ? (_bool ? 1.0 : 0.0) : dflt); }
    float asFloat(float dflt = 0.0f) const { return _type == Type::Number ? static_cast<float>(_number) : dflt; }  // Json::Value::asFloat
    int asInt(int dflt = 0) const { return _type == Type::Number ? static_cast<int>(_number) : dflt; }
    unsigned asUInt(unsigned dflt = 0) const { return _type == Type::Number ? static_cast<unsigned>(_number) : dflt; }
    bool asBool(bool dflt = false) const { return _type == Type::Bool ? _bool : (_type == Type::Number ? _number != 0.0 : dflt); }
    std::string asString(const std::string& dflt = "") const { return _type == Type::String ? _string : dflt; }

    // get(key, default) family
    float getFloat(const std::string& key, float dflt) const { return isMember(key) ? (*this)[key].asFloat(dflt) : dflt; }
    int getInt(const std::string& key, int dflt) const { return isMember(key) ? (*this)[key].asInt(dflt) : dflt; }
    unsigned getUInt(const std::string& key, unsigned dflt) const { return isMember(key) ? (*this)[key].asUInt(dflt) : dflt; }
    bool getBool(const std::string& key, bool dflt) const { return isMember(key) ? (*this)[key].asBool(dflt) : dflt; }
    std::string getString(const std::string& key, const std::string& dflt) const { return isMember(key) ? (*this)[key].asString(dflt) : dflt; }

private:
    friend class Parser;
    Type _type = Type::Null;
    bool _bool = false;
    double _number = 0.0;
    std::string _string;
    std::vector<Value> _array;
    std::map<std::string, Value> _object;
};

class Parser {
public:
    explicit Parser(const std::string& text) : _s(text) {}

    Value parse()
    {
        Value v = value();
        skip();
        if (_i != _s.size()) error("trailing characters");
        return v;
    }

private:
    const std::string& _s;
    std::size_t _i = 0;

    [[noreturn]] void error(const char* what) const
    {
        throw std::runtime_error(std::string("JSON parse error at byte ") + std::to_string(_i) + ": " + what);
    }

    void skip()
    {
        while (_i < _s.size()) {
            const char c = _s[_i];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { ++_i; continue; }
            if (c == '/' && _i + 1 < _s.size() && _s[_i + 1] == '/') {
                while (_i < _s.size() && _s[_i] != '\n') ++_i;
                continue;
            }
            if (c == '/' && _i + 1 < _s.size() && _s[_i + 1] == '*') {
                const std::size_t e = _s.find("*/", _i + 2);
                _i = (e == std::string::npos) ? _s.size() : e + 2;
                continue;
            }
            break;
        }
    }

    Value value()
    {
        skip();
        if (_i >= _s.size()) error("unexpected end");
        const char c = _s[_i];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Value v; v._type = Value::Type::String; v._string = string(); return v; }
        if (_s.compare(_i, 4, "true") == 0) { _i += 4; Value v; v._type = Value::Type::Bool; v._bool = true; return v; }
        if (_s.compare(_i, 5, "false") == 0) { _i += 5; Value v; v._type = Value::Type::Bool; v._bool = false; return v; }
        if (_s.compare(_i, 4, "null") == 0) { _i += 4; return Value(); }
        return number();
    }

    Value number()
    {
        const char* b = _s.c_str() + _i;
        char* e = nullptr;
        const double d = std::strtod(b, &e);
        if (e == b) error("bad number");
        _i += static_cast<std::size_t>(e - b);
        Value v;
        v._type = Value::Type::Number;
        v._number = d;
        return v;
    }

    std::string string()
    {
        std::string out;
        ++_i;  // opening quote
        while (_i < _s.size() && _s[_i] != '"') {
            char c = _s[_i++];
            if (c == '\\' && _i < _s.size()) {
                const char e = _s[_i++];
                switch (e) {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'u':
                    if (_i + 4 <= _s.size()) {
                        const unsigned cp = static_cast<unsigned>(std::strtoul(_s.substr(_i, 4).c_str(), nullptr, 16));
                        _i += 4;
                        if (cp < 0x80) out.push_back(static_cast<char>(cp));
                        else if (cp < 0x800) { out.push_back(static_cast<char>(0xC0 | (cp >> 6))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
                        else { out.push_back(static_cast<char>(0xE0 | (cp >> 12))); out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F))); out.push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
                    }
                    continue;
                default: c = e; break;
                }
            }
            out.push_back(c);
        }
        if (_i >= _s.size()) error("unterminated string");
        ++_i;  // closing quote
        return out;
    }

    Value array()
    {
        Value v;
        v._type = Value::Type::Array;
        ++_i;
        skip();
        if (_i < _s.size() && _s[_i] == ']') { ++_i; return v; }
        while (true) {
            v._array.push_back(value());
            skip();
            if (_i >= _s.size()) error("unterminated array");
            if (_s[_i] == ',') { ++_i; continue; }
            if (_s[_i] == ']') { ++_i; break; }
            error("expected , or ]");
        }
        return v;
    }

    Value object()
    {
        Value v;
        v._type = Value::Type::Object;
        ++_i;
        skip();
        if (_i < _s.size() && _s[_i] == '}') { ++_i; return v; }
        while (true) {
            skip();
            if (_i >= _s.size() || _s[_i] != '"') error("expected key");
            const std::string key = string();
            skip();
            if (_i >= _s.size() || _s[_i] != ':') error("expected :");
            ++_i;
            v._object[key] = value();
            skip();
            if (_i >= _s.size()) error("unterminated object");
            if (_s[_i] == ',') { ++_i; continue; }
            if (_s[_i] == '}') { ++_i; break; }
            error("expected , or }");
        }
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace lidarshooter
