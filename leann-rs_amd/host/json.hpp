// json.hpp — small JSON value / parser / writer for the on-disk formats of the index directory
// (documents.leann.meta.json, *.passages.jsonl, *.passages.idx.json) and the `--format json` output.
// Objects keep keys sorted (serde_json's default BTreeMap behaviour, which the reference relies on
// for its pretty output, src/cli/search.rs:211-223).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lj {

struct Value;
using Array = std::vector<Value>;
using Object = std::map<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Int, Float, String, Arr, Obj } kind = Null;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::shared_ptr<Array> a;
    std::shared_ptr<Object> o;

    Value() = default;
    static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
    static Value integer(int64_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value number(double v) { Value x; x.kind = Float; x.f = v; return x; }
    static Value string(std::string v) { Value x; x.kind = String; x.s = std::move(v); return x; }
    static Value array() { Value x; x.kind = Arr; x.a = std::make_shared<Array>(); return x; }
    static Value object() { Value x; x.kind = Obj; x.o = std::make_shared<Object>(); return x; }

    bool is_null() const { return kind == Null; }
    bool is_string() const { return kind == String; }
    bool is_number() const { return kind == Int || kind == Float; }
    bool is_object() const { return kind == Obj; }
    double as_f64() const { return kind == Int ? (double)i : f; }
    const Value *get(const std::string &key) const {
        if (kind != Obj) return nullptr;
        auto it = o->find(key);
        return it == o->end() ? nullptr : &it->second;
    }
    Value &operator[](const std::string &key) { return (*o)[key]; }
};

// ---- parser ----------------------------------------------------------------------------------
class Parser {
  public:
    explicit Parser(const std::string &t) : p_(t.data()), e_(t.data() + t.size()) {}
    Value parse() {
        Value v = value();
        ws();
        if (p_ != e_) fail("trailing characters");
        return v;
    }

  private:
    const char *p_, *e_;
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("JSON: ") + m); }
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
    bool lit(const char *w) {
        size_t n = strlen(w);
        if ((size_t)(e_ - p_) >= n && !memcmp(p_, w, n)) { p_ += n; return true; }
        return false;
    }
    static void utf8(std::string &out, uint32_t cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (e_ - p_ < 4) fail("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string str() {
        if (p_ >= e_ || *p_ != '"') fail("expected string");
        ++p_;
        std::string out;
        while (p_ < e_ && *p_ != '"') {
            char c = *p_++;
            if (c != '\\') { out += c; continue; }
            if (p_ >= e_) fail("bad escape");
            char x = *p_++;
            switch (x) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        uint32_t lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(out, cp);
                    break;
                }
                default: out += x; // \" \\ \/
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return out;
    }
    Value value() {
        ws();
        if (p_ >= e_) fail("unexpected end");
        char c = *p_;
        if (c == '{') {
            ++p_;
            Value v = Value::object();
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return v; }
            for (;;) {
                ws();
                std::string k = str();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                (*v.o)[k] = value();
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            Value v = Value::array();
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return v; }
            for (;;) {
                v.a->push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') return Value::string(str());
        if (lit("true")) return Value::boolean(true);
        if (lit("false")) return Value::boolean(false);
        if (lit("null")) return Value();
        const char *s = p_;
        bool is_float = false;
        if (p_ < e_ && (*p_ == '-' || *p_ == '+')) ++p_;
        while (p_ < e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) {
            if (*p_ == '.' || *p_ == 'e' || *p_ == 'E') is_float = true;
            ++p_;
        }
        if (s == p_) fail("unexpected character");
        std::string num(s, p_);
        if (!is_float) {
            try { return Value::integer(std::stoll(num)); } catch (...) { is_float = true; }
        }
        return Value::number(std::stod(num));
    }
};
inline Value parse(const std::string &t) { return Parser(t).parse(); }

// ---- writer ----------------------------------------------------------------------------------
inline void write_string(std::string &out, const std::string &s) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
                else out += (char)c;
        }
    }
    out += '"';
}
// shortest round-trip decimal of a double, in serde_json/ryu style ("1.0", "1e-7", "0.8999999761581421")
inline std::string format_f64(double v) {
    if (!std::isfinite(v)) return "null";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    size_t e = s.find('e');
    if (e != std::string::npos) { // normalise exponent: e-07 -> e-7, e+20 -> e20
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        bool neg = !ex.empty() && ex[0] == '-';
        if (!ex.empty() && (ex[0] == '-' || ex[0] == '+')) ex.erase(0, 1);
        while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
        return mant + "e" + (neg ? "-" : "") + ex;
    }
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}
inline void write(std::string &out, const Value &v, int indent = -1, int depth = 0) {
    auto nl = [&](int d) {
        if (indent < 0) return;
        out += '\n';
        out.append((size_t)(indent * d), ' ');
    };
    switch (v.kind) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Int: out += std::to_string(v.i); break;
        case Value::Float: out += format_f64(v.f); break;
        case Value::String: write_string(out, v.s); break;
        case Value::Arr:
            if (v.a->empty()) { out += "[]"; break; }
            out += '[';
            for (size_t k = 0; k < v.a->size(); k++) {
                if (k) out += ',';
                nl(depth + 1);
                write(out, (*v.a)[k], indent, depth + 1);
            }
            nl(depth);
            out += ']';
            break;
        case Value::Obj: {
            if (v.o->empty()) { out += "{}"; break; }
            out += '{';
            bool first = true;
            for (auto &kv : *v.o) {
                if (!first) out += ',';
                first = false;
                nl(depth + 1);
                write_string(out, kv.first);
                out += indent < 0 ? ":" : ": ";
                write(out, kv.second, indent, depth + 1);
            }
            nl(depth);
            out += '}';
            break;
        }
    }
}
inline std::string to_string(const Value &v) { std::string s; write(s, v); return s; }
inline std::string to_string_pretty(const Value &v) { std::string s; write(s, v, 2); return s; }

} // namespace lj
