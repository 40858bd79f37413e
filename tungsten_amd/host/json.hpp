// Minimal JSON DOM for the Tungsten scene schema (objects, arrays, strings,
// numbers, booleans, null).  The reference parses with rapidjson into doubles
// and casts to float at the point of use (src/core/io/JsonPtr.hpp:53-65); we
// keep numbers as double for the same reason.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace tungsten_amd {

struct JsonLoadException : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj; // keeps file order

    bool isNull() const { return type == Null; }
    bool isNumber() const { return type == Number; }
    bool isString() const { return type == String; }
    bool isArray() const { return type == Array; }
    bool isObject() const { return type == Object; }
    bool isBool() const { return type == Bool; }
    explicit operator bool() const { return type != Null; }

    const Json &operator[](const std::string &key) const
    {
        static const Json null;
        if (type != Object) return null;
        for (auto &kv : obj)
            if (kv.first == key) return kv.second;
        return null;
    }
    const Json &operator[](size_t i) const { return arr.at(i); }
    size_t size() const { return type == Array ? arr.size() : obj.size(); }

    float asFloat() const
    {
        if (type != Number) throw JsonLoadException("JSON: expected a number");
        return float(num);
    }
    int asInt() const
    {
        if (type != Number) throw JsonLoadException("JSON: expected a number");
        return int(num);
    }
    bool asBool() const
    {
        if (type != Bool) throw JsonLoadException("JSON: expected a boolean");
        return b;
    }
    const std::string &asString() const
    {
        if (type != String) throw JsonLoadException("JSON: expected a string");
        return str;
    }

    // getField semantics of the reference: leave dst untouched when the key is absent
    bool getField(const std::string &key, float &dst) const
    {
        const Json &v = (*this)[key];
        if (!v) return false;
        dst = v.asFloat();
        return true;
    }
    bool getField(const std::string &key, int &dst) const
    {
        const Json &v = (*this)[key];
        if (!v) return false;
        dst = v.asInt();
        return true;
    }
    bool getField(const std::string &key, bool &dst) const
    {
        const Json &v = (*this)[key];
        if (!v) return false;
        dst = v.asBool();
        return true;
    }
    bool getField(const std::string &key, std::string &dst) const
    {
        const Json &v = (*this)[key];
        if (!v) return false;
        dst = v.asString();
        return true;
    }
    // scalar -> broadcast, array -> per component (JsonPtr.hpp:53-65)
    bool getVec(const std::string &key, float *dst, int n) const
    {
        const Json &v = (*this)[key];
        if (!v) return false;
        v.toVec(dst, n);
        return true;
    }
    void toVec(float *dst, int n) const
    {
        if (type == Number) {
            for (int i = 0; i < n; ++i) dst[i] = float(num);
        } else if (type == Array) {
            if (int(arr.size()) != n) throw JsonLoadException("JSON: vector has wrong number of elements");
            for (int i = 0; i < n; ++i) dst[i] = arr[i].asFloat();
        } else {
            throw JsonLoadException("JSON: expected a number or an array");
        }
    }

    static Json parse(const std::string &text)
    {
        Parser p{text, 0};
        Json v = p.value();
        p.ws();
        if (p.pos != text.size()) p.fail("trailing characters");
        return v;
    }

private:
    struct Parser {
        const std::string &s;
        size_t pos;
        [[noreturn]] void fail(const char *msg)
        {
            throw JsonLoadException("JSON parse error at offset " + std::to_string(pos) + ": " + msg);
        }
        void ws()
        {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\n' || s[pos] == '\t' || s[pos] == '\r')) ++pos;
        }
        Json value()
        {
            ws();
            if (pos >= s.size()) fail("unexpected end");
            char c = s[pos];
            if (c == '{') return object();
            if (c == '[') return array();
            if (c == '"') { Json j; j.type = String; j.str = string(); return j; }
            if (s.compare(pos, 4, "true") == 0) { pos += 4; Json j; j.type = Bool; j.b = true; return j; }
            if (s.compare(pos, 5, "false") == 0) { pos += 5; Json j; j.type = Bool; j.b = false; return j; }
            if (s.compare(pos, 4, "null") == 0) { pos += 4; return Json(); }
            return number();
        }
        Json number()
        {
            const char *begin = s.c_str() + pos;
            char *end = nullptr;
            double d = std::strtod(begin, &end);
            if (end == begin) fail("invalid value");
            pos += size_t(end - begin);
            Json j; j.type = Number; j.num = d;
            return j;
        }
        std::string string()
        {
            ++pos; // opening quote
            std::string out;
            while (true) {
                if (pos >= s.size()) fail("unterminated string");
                char c = s[pos++];
                if (c == '"') break;
                if (c == '\\') {
                    if (pos >= s.size()) fail("bad escape");
                    char e = s[pos++];
                    switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (pos + 4 > s.size()) fail("bad \\u escape");
                        unsigned cp = unsigned(std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16));
                        pos += 4;
                        if (cp < 0x80) out += char(cp);
                        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
                        else { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e; break;
                    }
                } else {
                    out += c;
                }
            }
            return out;
        }
        Json array()
        {
            Json j; j.type = Array;
            ++pos; ws();
            if (pos < s.size() && s[pos] == ']') { ++pos; return j; }
            while (true) {
                j.arr.push_back(value());
                ws();
                if (pos >= s.size()) fail("unterminated array");
                if (s[pos] == ',') { ++pos; continue; }
                if (s[pos] == ']') { ++pos; break; }
                fail("expected ',' or ']'");
            }
            return j;
        }
        Json object()
        {
            Json j; j.type = Object;
            ++pos; ws();
            if (pos < s.size() && s[pos] == '}') { ++pos; return j; }
            while (true) {
                ws();
                if (pos >= s.size() || s[pos] != '"') fail("expected string key");
                std::string key = string();
                ws();
                if (pos >= s.size() || s[pos] != ':') fail("expected ':'");
                ++pos;
                Json v = value();
                j.obj.emplace_back(std::move(key), std::move(v));
                ws();
                if (pos >= s.size()) fail("unterminated object");
                if (s[pos] == ',') { ++pos; continue; }
                if (s[pos] == '}') { ++pos; break; }
                fail("expected ',' or '}'");
            }
            return j;
        }
    };
};

} // namespace tungsten_amd
