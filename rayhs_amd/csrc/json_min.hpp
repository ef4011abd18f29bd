// json_min.hpp -- small strict JSON reader used by the scene loader.
// Enough of RFC 8259 for aeson's `decode` (src/RayHs.hs:197-200): one value,
// surrounding whitespace only; numbers go through strtod (correctly rounded,
// like aeson's Scientific -> Double).
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rayhs {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;  // insertion order; last duplicate wins on lookup

    const JsonValue *find(const char *key) const {
        const JsonValue *r = nullptr;
        for (auto &kv : obj)
            if (kv.first == key) r = &kv.second;
        return r;
    }
};

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class JsonReader {
  public:
    explicit JsonReader(const std::string &text) : s_(text.data()), n_(text.size()) {}
    JsonValue parse_document() {
        JsonValue v = value(0);
        ws();
        if (i_ != n_) throw JsonError("trailing characters after JSON value");
        return v;
    }

  private:
    const char *s_;
    size_t n_, i_ = 0;
    void ws() {
        while (i_ < n_ && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) i_++;
    }
    [[noreturn]] void bad(const char *what) { throw JsonError(std::string(what) + " at byte " + std::to_string(i_)); }
    bool lit(const char *w) {
        size_t l = strlen(w);
        if (n_ - i_ >= l && memcmp(s_ + i_, w, l) == 0) { i_ += l; return true; }
        return false;
    }
    JsonValue value(int depth) {
        if (depth > 512) bad("nesting too deep");
        ws();
        if (i_ >= n_) bad("unexpected end");
        JsonValue v;
        char c = s_[i_];
        if (c == '{') {
            v.kind = JsonValue::Object;
            i_++; ws();
            if (i_ < n_ && s_[i_] == '}') { i_++; return v; }
            for (;;) {
                ws();
                if (i_ >= n_ || s_[i_] != '"') bad("expected object key");
                std::string k = string();
                ws();
                if (i_ >= n_ || s_[i_] != ':') bad("expected ':'");
                i_++;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (i_ < n_ && s_[i_] == ',') { i_++; continue; }
                if (i_ < n_ && s_[i_] == '}') { i_++; break; }
                bad("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JsonValue::Array;
            i_++; ws();
            if (i_ < n_ && s_[i_] == ']') { i_++; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (i_ < n_ && s_[i_] == ',') { i_++; continue; }
                if (i_ < n_ && s_[i_] == ']') { i_++; break; }
                bad("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JsonValue::String;
            v.str = string();
        } else if (lit("true")) { v.kind = JsonValue::Bool; v.b = true; }
        else if (lit("false")) { v.kind = JsonValue::Bool; v.b = false; }
        else if (lit("null")) { v.kind = JsonValue::Null; }
        else {
            v.kind = JsonValue::Number;
            v.num = number();
        }
        return v;
    }
    double number() {
        size_t b = i_;
        if (i_ < n_ && s_[i_] == '-') i_++;
        if (i_ >= n_) bad("bad number");
        if (s_[i_] == '0') i_++;
        else if (s_[i_] >= '1' && s_[i_] <= '9') { while (i_ < n_ && isdig(s_[i_])) i_++; }
        else bad("bad number");
        if (i_ < n_ && s_[i_] == '.') {
            i_++;
            if (i_ >= n_ || !isdig(s_[i_])) bad("bad fraction");
            while (i_ < n_ && isdig(s_[i_])) i_++;
        }
        if (i_ < n_ && (s_[i_] == 'e' || s_[i_] == 'E')) {
            i_++;
            if (i_ < n_ && (s_[i_] == '+' || s_[i_] == '-')) i_++;
            if (i_ >= n_ || !isdig(s_[i_])) bad("bad exponent");
            while (i_ < n_ && isdig(s_[i_])) i_++;
        }
        std::string t(s_ + b, i_ - b);
        return strtod(t.c_str(), nullptr);
    }
    static bool isdig(char c) { return c >= '0' && c <= '9'; }
    static void utf8(std::string &o, unsigned cp) {
        if (cp < 0x80) o += char(cp);
        else if (cp < 0x800) { o += char(0xC0 | (cp >> 6)); o += char(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += char(0xE0 | (cp >> 12)); o += char(0x80 | ((cp >> 6) & 0x3F)); o += char(0x80 | (cp & 0x3F)); }
        else { o += char(0xF0 | (cp >> 18)); o += char(0x80 | ((cp >> 12) & 0x3F)); o += char(0x80 | ((cp >> 6) & 0x3F)); o += char(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (n_ - i_ < 4) bad("bad \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            char c = s_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= unsigned(c - '0');
            else if (c >= 'a' && c <= 'f') v |= unsigned(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= unsigned(c - 'A' + 10);
            else bad("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        std::string o;
        i_++;  // opening quote
        for (;;) {
            if (i_ >= n_) bad("unterminated string");
            unsigned char c = (unsigned char)s_[i_++];
            if (c == '"') break;
            if (c < 0x20) bad("control character in string");
            if (c != '\\') { o += char(c); continue; }
            if (i_ >= n_) bad("bad escape");
            char e = s_[i_++];
            switch (e) {
            case '"': o += '"'; break;
            case '\\': o += '\\'; break;
            case '/': o += '/'; break;
            case 'b': o += '\b'; break;
            case 'f': o += '\f'; break;
            case 'n': o += '\n'; break;
            case 'r': o += '\r'; break;
            case 't': o += '\t'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF && n_ - i_ >= 6 && s_[i_] == '\\' && s_[i_ + 1] == 'u') {
                    i_ += 2;
                    unsigned lo = hex4();
                    if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(o, cp);
                break;
            }
            default: bad("bad escape");
            }
        }
        return o;
    }
};

}  // namespace rayhs
