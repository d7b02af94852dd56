// Minimal JSON reader for the three scene-directory files (settings.json, scene.json,
// denoising.json).  The reference links nlohmann/json for this (src/scene.cpp:1,15);
// only parsing into a tree and typed access are needed here, so this is a small
// recursive-descent parser that keeps object members in file order.
#pragma once
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ptkjson {

struct Value;
using Member = std::pair<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Value> items;
    std::vector<Member> members;

    bool contains(const std::string& key) const {
        if (kind != Object) return false;
        for (const auto& m : members)
            if (m.first == key) return true;
        return false;
    }
    const Value& at(const std::string& key) const {
        for (const auto& m : members)
            if (m.first == key) return m.second;
        throw std::runtime_error("JSON structure missing key '" + key + "'");
    }
    size_t size() const { return kind == Array ? items.size() : (kind == Object ? members.size() : 1); }
    double as_number() const {
        if (kind == Number) return num;
        if (kind == Bool) return b ? 1.0 : 0.0;
        throw std::runtime_error("JSON value is not a number");
    }
    bool as_bool() const {
        if (kind == Bool) return b;
        if (kind == Number) return num != 0;
        throw std::runtime_error("JSON value is not a boolean");
    }
    const std::string& as_string() const {
        if (kind != String) throw std::runtime_error("JSON value is not a string");
        return str;
    }
    std::string dump() const {
        switch (kind) {
            case Null: return "null";
            case Bool: return b ? "true" : "false";
            case Number: {
                char buf[64];
                if (num == std::floor(num) && std::fabs(num) < 1e15)
                    snprintf(buf, sizeof buf, "%lld", (long long) num);
                else
                    snprintf(buf, sizeof buf, "%.17g", num);
                return buf;
            }
            case String: return "\"" + str + "\"";
            case Array: {
                std::string s = "[";
                for (size_t i = 0; i < items.size(); i++) s += (i ? "," : "") + items[i].dump();
                return s + "]";
            }
            case Object: {
                std::string s = "{";
                for (size_t i = 0; i < members.size(); i++)
                    s += (i ? "," : "") + ("\"" + members[i].first + "\":") + members[i].second.dump();
                return s + "}";
            }
        }
        return "";
    }
};

class Parser {
  public:
    explicit Parser(const std::string& text) : s(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string& s;
    size_t p = 0;

    [[noreturn]] void fail(const std::string& what) const {
        throw std::runtime_error("JSON parse error at byte " + std::to_string(p) + ": " + what);
    }
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
    }
    Value value() {
        ws();
        if (p >= s.size()) fail("unexpected end of input");
        char c = s[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            Value v;
            v.kind = Value::String;
            v.str = string();
            return v;
        }
        if (s.compare(p, 4, "true") == 0) {
            p += 4;
            Value v;
            v.kind = Value::Bool;
            v.b = true;
            return v;
        }
        if (s.compare(p, 5, "false") == 0) {
            p += 5;
            Value v;
            v.kind = Value::Bool;
            v.b = false;
            return v;
        }
        if (s.compare(p, 4, "null") == 0) {
            p += 4;
            return Value();
        }
        return number();
    }
    Value number() {
        const char* begin = s.c_str() + p;
        char* end = nullptr;
        double d = std::strtod(begin, &end);
        if (end == begin) fail("invalid value");
        p += static_cast<size_t>(end - begin);
        Value v;
        v.kind = Value::Number;
        v.num = d;
        return v;
    }
    std::string string() {
        std::string out;
        p++; // opening quote
        while (true) {
            if (p >= s.size()) fail("unterminated string");
            char c = s[p++];
            if (c == '"') break;
            if (c == '\\') {
                if (p >= s.size()) fail("unterminated escape");
                char e = s[p++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (p + 4 > s.size()) fail("bad \\u escape");
                        unsigned code = static_cast<unsigned>(std::strtoul(s.substr(p, 4).c_str(), nullptr, 16));
                        p += 4;
                        if (code < 0x80) {
                            out += static_cast<char>(code);
                        }
                        else if (code < 0x800) {
                            out += static_cast<char>(0xC0 | (code >> 6));
                            out += static_cast<char>(0x80 | (code & 0x3F));
                        }
                        else {
                            out += static_cast<char>(0xE0 | (code >> 12));
                            out += static_cast<char>(0x80 | ((code >> 6) & 0x3F));
                            out += static_cast<char>(0x80 | (code & 0x3F));
                        }
                        break;
                    }
                    default: out += e; // \" \\ \/
                }
            }
            else {
                out += c;
            }
        }
        return out;
    }
    Value array() {
        Value v;
        v.kind = Value::Array;
        p++;
        ws();
        if (p < s.size() && s[p] == ']') {
            p++;
            return v;
        }
        while (true) {
            v.items.push_back(value());
            ws();
            if (p >= s.size()) fail("unterminated array");
            if (s[p] == ',') {
                p++;
                continue;
            }
            if (s[p] == ']') {
                p++;
                break;
            }
            fail("expected ',' or ']'");
        }
        return v;
    }
    Value object() {
        Value v;
        v.kind = Value::Object;
        p++;
        ws();
        if (p < s.size() && s[p] == '}') {
            p++;
            return v;
        }
        while (true) {
            ws();
            if (p >= s.size() || s[p] != '"') fail("expected member name");
            std::string key = string();
            ws();
            if (p >= s.size() || s[p] != ':') fail("expected ':'");
            p++;
            Value member = value();
            bool replaced = false;
            for (auto& m : v.members) {
                if (m.first == key) { // later duplicates win, as in nlohmann/json
                    m.second = member;
                    replaced = true;
                }
            }
            if (!replaced) v.members.emplace_back(key, std::move(member));
            ws();
            if (p >= s.size()) fail("unterminated object");
            if (s[p] == ',') {
                p++;
                continue;
            }
            if (s[p] == '}') {
                p++;
                break;
            }
            fail("expected ',' or '}'");
        }
        return v;
    }
};

inline Value parse(const std::string& text) {
    return Parser(text).parse();
}

} // namespace ptkjson
