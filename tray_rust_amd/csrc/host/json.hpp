// Minimal JSON reader for the scene loader. Replaces serde_json::Value walking in the reference
// (src/scene.rs:108-112); only what the scene schema needs: objects, arrays, strings, numbers,
// true/false/null. Numbers remember whether they were written as unsigned integers so that
// as_u64() fails on "800.0" the way serde_json's does.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace trayh {

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0.0;
    bool is_uint = false;
    unsigned long long u = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    bool is_number() const { return type == Number; }
    bool is_string() const { return type == String; }

    // serde_json Value::get: nullptr when absent or when this is not an object
    const Json* get(const char* key) const {
        if (type != Object) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool as_f64(double& out) const {
        if (type != Number) return false;
        out = num;
        return true;
    }
    bool as_u64(unsigned long long& out) const {
        if (type != Number || !is_uint) return false;
        out = u;
        return true;
    }

    static Json parse(const std::string& text) {
        Parser p{text, 0};
        p.skip_ws();
        Json v = p.value();
        p.skip_ws();
        if (p.pos != text.size()) p.fail("trailing characters");
        return v;
    }

private:
    struct Parser {
        const std::string& s;
        size_t pos;
        [[noreturn]] void fail(const char* msg) {
            size_t line = 1, col = 1;
            for (size_t i = 0; i < pos && i < s.size(); ++i) {
                if (s[i] == '\n') { ++line; col = 1; } else ++col;
            }
            throw JsonError(std::string("JSON parsing error: ") + msg + " at line " +
                            std::to_string(line) + " column " + std::to_string(col));
        }
        void skip_ws() {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) ++pos;
        }
        char peek() { return pos < s.size() ? s[pos] : '\0'; }
        Json value() {
            skip_ws();
            char c = peek();
            if (c == '{') return object();
            if (c == '[') return array();
            if (c == '"') { Json j; j.type = String; j.str = string(); return j; }
            if (c == 't' || c == 'f' || c == 'n') return literal();
            if (c == '-' || (c >= '0' && c <= '9')) return number();
            fail("unexpected character");
        }
        Json literal() {
            Json j;
            if (s.compare(pos, 4, "true") == 0) { j.type = Bool; j.b = true; pos += 4; }
            else if (s.compare(pos, 5, "false") == 0) { j.type = Bool; j.b = false; pos += 5; }
            else if (s.compare(pos, 4, "null") == 0) { j.type = Null; pos += 4; }
            else fail("invalid literal");
            return j;
        }
        Json number() {
            size_t start = pos;
            bool integral = true, negative = false;
            if (peek() == '-') { negative = true; ++pos; }
            if (!(peek() >= '0' && peek() <= '9')) fail("invalid number");
            while (peek() >= '0' && peek() <= '9') ++pos;
            if (peek() == '.') {
                integral = false;
                ++pos;
                if (!(peek() >= '0' && peek() <= '9')) fail("invalid number");
                while (peek() >= '0' && peek() <= '9') ++pos;
            }
            if (peek() == 'e' || peek() == 'E') {
                integral = false;
                ++pos;
                if (peek() == '+' || peek() == '-') ++pos;
                if (!(peek() >= '0' && peek() <= '9')) fail("invalid number");
                while (peek() >= '0' && peek() <= '9') ++pos;
            }
            std::string tok = s.substr(start, pos - start);
            Json j;
            j.type = Number;
            j.num = std::strtod(tok.c_str(), nullptr);
            if (integral && !negative && tok.size() <= 19) {
                j.is_uint = true;
                j.u = std::strtoull(tok.c_str(), nullptr, 10);
            }
            return j;
        }
        static void append_utf8(std::string& out, unsigned cp) {
            if (cp < 0x80) out += char(cp);
            else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) {
                out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F));
            } else {
                out += char(0xF0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3F));
                out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F));
            }
        }
        unsigned hex4() {
            if (pos + 4 > s.size()) fail("truncated \\u escape");
            unsigned v = 0;
            for (int i = 0; i < 4; ++i) {
                char c = s[pos++];
                v <<= 4;
                if (c >= '0' && c <= '9') v |= unsigned(c - '0');
                else if (c >= 'a' && c <= 'f') v |= unsigned(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= unsigned(c - 'A' + 10);
                else fail("bad hex digit");
            }
            return v;
        }
        std::string string() {
            std::string out;
            ++pos;  // opening quote
            for (;;) {
                if (pos >= s.size()) fail("unterminated string");
                char c = s[pos++];
                if (c == '"') break;
                if (c == '\\') {
                    if (pos >= s.size()) fail("unterminated escape");
                    char e = s[pos++];
                    switch (e) {
                        case '"': out += '"'; break;
                        case '\\': out += '\\'; break;
                        case '/': out += '/'; break;
                        case 'b': out += '\b'; break;
                        case 'f': out += '\f'; break;
                        case 'n': out += '\n'; break;
                        case 'r': out += '\r'; break;
                        case 't': out += '\t'; break;
                        case 'u': {
                            unsigned cp = hex4();
                            if (cp >= 0xD800 && cp <= 0xDBFF && pos + 1 < s.size() && s[pos] == '\\' && s[pos + 1] == 'u') {
                                pos += 2;
                                unsigned lo = hex4();
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            }
                            append_utf8(out, cp);
                            break;
                        }
                        default: fail("bad escape");
                    }
                } else {
                    out += c;
                }
            }
            return out;
        }
        Json array() {
            Json j;
            j.type = Array;
            ++pos;
            skip_ws();
            if (peek() == ']') { ++pos; return j; }
            for (;;) {
                j.arr.push_back(value());
                skip_ws();
                if (peek() == ',') { ++pos; continue; }
                if (peek() == ']') { ++pos; break; }
                fail("expected ',' or ']'");
            }
            return j;
        }
        Json object() {
            Json j;
            j.type = Object;
            ++pos;
            skip_ws();
            if (peek() == '}') { ++pos; return j; }
            for (;;) {
                skip_ws();
                if (peek() != '"') fail("expected object key");
                std::string k = string();
                skip_ws();
                if (peek() != ':') fail("expected ':'");
                ++pos;
                Json v = value();
                // serde_json keeps the last duplicate
                bool replaced = false;
                for (auto& kv : j.obj)
                    if (kv.first == k) { kv.second = v; replaced = true; break; }
                if (!replaced) j.obj.emplace_back(std::move(k), std::move(v));
                skip_ws();
                if (peek() == ',') { ++pos; continue; }
                if (peek() == '}') { ++pos; break; }
                fail("expected ',' or '}'");
            }
            return j;
        }
    };
};

}  // namespace trayh
