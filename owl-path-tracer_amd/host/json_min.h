// json_min.h -- minimal JSON DOM (objects, arrays, strings, numbers, true/false/null) for the host entry point.
// Replaces nlohmann/json (an empty submodule in the reference, path_tracer/src/utils/parser.cpp:5-17).  Like
// nlohmann's .get<>() on a missing key, at() / as_*() throw std::runtime_error instead of inventing defaults.
#pragma once
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace jsonmin {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj; // insertion order kept (material order = JSON order)

    bool is_array() const { return kind == Array; }
    const Value& at(const std::string& key) const
    {
        if (kind != Object) throw std::runtime_error("json: not an object while looking up '" + key + "'");
        for (auto const& kv : obj)
            if (kv.first == key) return kv.second;
        throw std::runtime_error("json: missing key '" + key + "'");
    }
    const Value& at(size_t i) const
    {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[i];
    }
    float as_float() const
    {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return (float)num;
    }
    int as_int() const
    {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return (int)num;
    }
    bool as_bool() const
    {
        if (kind != Bool) throw std::runtime_error("json: boolean expected");
        return b;
    }
    const std::string& as_string() const
    {
        if (kind != String) throw std::runtime_error("json: string expected");
        return str;
    }
};

class Parser {
public:
    explicit Parser(const std::string& text) : s(text) {}
    Value parse()
    {
        Value v = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t p = 0;
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p)); }
    void ws()
    {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p;
        if (p + 2 < s.size() && (unsigned char)s[p] == 0xEF && (unsigned char)s[p + 1] == 0xBB && (unsigned char)s[p + 2] == 0xBF) { p += 3; ws(); }
    }
    Value value()
    {
        ws();
        if (p >= s.size()) fail("unexpected end");
        char c = s[p];
        Value v;
        if (c == '{') {
            v.kind = Value::Object;
            ++p;
            ws();
            if (p < s.size() && s[p] == '}') { ++p; return v; }
            for (;;) {
                ws();
                if (p >= s.size() || s[p] != '"') fail("key expected");
                std::string k = string();
                ws();
                if (p >= s.size() || s[p] != ':') fail("':' expected");
                ++p;
                Value e = value();
                v.obj.emplace_back(std::move(k), std::move(e));
                ws();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == '}') { ++p; break; }
                fail("',' or '}' expected");
            }
        } else if (c == '[') {
            v.kind = Value::Array;
            ++p;
            ws();
            if (p < s.size() && s[p] == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == ']') { ++p; break; }
                fail("',' or ']' expected");
            }
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = string();
        } else if (s.compare(p, 4, "true") == 0) {
            v.kind = Value::Bool; v.b = true; p += 4;
        } else if (s.compare(p, 5, "false") == 0) {
            v.kind = Value::Bool; v.b = false; p += 5;
        } else if (s.compare(p, 4, "null") == 0) {
            p += 4;
        } else {
            char* end = nullptr;
            v.num = std::strtod(s.c_str() + p, &end);
            if (end == s.c_str() + p) fail("value expected");
            v.kind = Value::Number;
            p = (size_t)(end - s.c_str());
        }
        return v;
    }
    std::string string()
    {
        std::string out;
        ++p; // opening quote
        while (p < s.size() && s[p] != '"') {
            char c = s[p++];
            if (c == '\\') {
                if (p >= s.size()) fail("bad escape");
                char e = s[p++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (p + 4 > s.size()) fail("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(s.substr(p, 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e;
                }
            } else {
                out += c;
            }
        }
        if (p >= s.size()) fail("unterminated string");
        ++p;
        return out;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

} // namespace jsonmin
