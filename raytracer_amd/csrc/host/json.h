// Minimal JSON DOM reader for the scene dialect (all scene scalars are strings; numbers, booleans
// and null are still accepted and kept as their source text).  Header-only.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rtjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    std::string str;                                   // String / Number / Bool source text
    std::vector<Value> arr;                            // Array
    std::vector<std::pair<std::string, Value>> obj;    // Object (document order kept)

    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool has(const char *k) const { return find(k) != nullptr; }
    const Value *find(const char *k) const {
        if (kind != Object) return nullptr;
        for (auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    // Missing members read as a Null value so that chained lookups do not crash.
    const Value &operator[](const char *k) const {
        static const Value null_value;
        const Value *v = find(k);
        return v ? *v : null_value;
    }
    const char *c_str() const { return str.c_str(); }
    // "single object or array" convention of the dialect: visit one or many.
    template <class F> void each(F &&f) const {
        if (kind == Array) {
            for (auto &v : arr) f(v);
        } else if (kind == Object) {
            f(*this);
        }
    }
};

class Parser {
  public:
    explicit Parser(const std::string &text) : s(text), p(0) {}
    bool parse(Value &out, std::string &err) {
        skip();
        if (!value(out, err)) return false;
        return true;
    }

  private:
    const std::string &s;
    size_t p;
    void skip() {
        while (p < s.size()) {
            char c = s[p];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                ++p;
            } else if (c == '/' && p + 1 < s.size() && s[p + 1] == '/') {
                while (p < s.size() && s[p] != '\n') ++p;
            } else if (c == '/' && p + 1 < s.size() && s[p + 1] == '*') {
                p += 2;
                while (p + 1 < s.size() && !(s[p] == '*' && s[p + 1] == '/')) ++p;
                p += 2;
            } else {
                break;
            }
        }
    }
    bool fail(std::string &err, const char *what) {
        char buf[128];
        snprintf(buf, sizeof buf, "JSON parse error at offset %zu: %s", p, what);
        err = buf;
        return false;
    }
    bool string(std::string &out, std::string &err) {
        ++p;  // opening quote
        while (p < s.size() && s[p] != '"') {
            char c = s[p++];
            if (c == '\\' && p < s.size()) {
                char e = s[p++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        unsigned cp = 0;
                        for (int i = 0; i < 4 && p < s.size(); ++i) cp = cp * 16 + hex(s[p++]);
                        if (cp < 0x80) {
                            out += (char)cp;
                        } else if (cp < 0x800) {
                            out += (char)(0xC0 | (cp >> 6));
                            out += (char)(0x80 | (cp & 0x3F));
                        } else {
                            out += (char)(0xE0 | (cp >> 12));
                            out += (char)(0x80 | ((cp >> 6) & 0x3F));
                            out += (char)(0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        if (p >= s.size()) return fail(err, "unterminated string");
        ++p;
        return true;
    }
    static unsigned hex(char c) {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return 0;
    }
    bool value(Value &v, std::string &err) {
        skip();
        if (p >= s.size()) return fail(err, "unexpected end");
        char c = s[p];
        if (c == '{') {
            v.kind = Value::Object;
            ++p;
            skip();
            if (p < s.size() && s[p] == '}') {
                ++p;
                return true;
            }
            while (true) {
                skip();
                if (p >= s.size() || s[p] != '"') return fail(err, "expected member name");
                std::string key;
                if (!string(key, err)) return false;
                skip();
                if (p >= s.size() || s[p] != ':') return fail(err, "expected ':'");
                ++p;
                Value child;
                if (!value(child, err)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                skip();
                if (p < s.size() && s[p] == ',') {
                    ++p;
                    continue;
                }
                if (p < s.size() && s[p] == '}') {
                    ++p;
                    return true;
                }
                return fail(err, "expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Value::Array;
            ++p;
            skip();
            if (p < s.size() && s[p] == ']') {
                ++p;
                return true;
            }
            while (true) {
                Value child;
                if (!value(child, err)) return false;
                v.arr.emplace_back(std::move(child));
                skip();
                if (p < s.size() && s[p] == ',') {
                    ++p;
                    continue;
                }
                if (p < s.size() && s[p] == ']') {
                    ++p;
                    return true;
                }
                return fail(err, "expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = Value::String;
            return string(v.str, err);
        }
        size_t b = p;
        while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '-' || s[p] == '+' || s[p] == '.')) ++p;
        if (p == b) return fail(err, "unexpected character");
        v.str = s.substr(b, p - b);
        if (v.str == "true" || v.str == "false")
            v.kind = Value::Bool;
        else if (v.str == "null")
            v.kind = Value::Null;
        else
            v.kind = Value::Number;
        return true;
    }
};

inline bool parse_file(const std::string &path, Value &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    Parser ps(text);
    return ps.parse(out, err);
}

}  // namespace rtjson
