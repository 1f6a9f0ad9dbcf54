// Minimal JSON DOM parser (objects keep member order): enough for a HuggingFace tokenizer.json and a safetensors header.
#pragma once
#include "common.h"

namespace wrs {
namespace json {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* key) const {
        for (auto& kv : o)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

inline void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
}

struct JParser {
    const char* p;
    const char* end;
    int depth = 0;
    [[noreturn]] void fail(const char* what) { WRS_THROW(WRS_EINVAL, "json: %s, %ld bytes before the end", what, (long)(end - p)); }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    uint32_t hex4() {
        if (end - p < 4) fail("short \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string str() {
        if (p >= end || *p != '"') fail("expected string");
        ++p;
        std::string out;
        while (true) {
            if (p >= end) fail("unterminated string");
            const char c = *p++;
            if (c == '"') break;
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= end) fail("bad escape");
            const char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2;
                        const uint32_t lo = hex4();
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else { append_utf8(out, 0xFFFD); cp = lo; }
                    }
                    append_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        return out;
    }
    JVal val() {
        if (++depth > 64) fail("nesting too deep");
        ws();
        if (p >= end) fail("unexpected end");
        JVal v;
        const char c = *p;
        if (c == '{') {
            v.kind = JVal::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; --depth; return v; }
            while (true) {
                ws();
                std::string k = str();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v.o.emplace_back(std::move(k), val());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JVal::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; --depth; return v; }
            while (true) {
                v.a.push_back(val());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JVal::Str;
            v.s = str();
        } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; }
        else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; }
        else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; }
        else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* q = p;
            while (q < end && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) ++q;
            v.kind = JVal::Num;
            v.num = strtod(std::string(p, q).c_str(), nullptr);
            p = q;
        } else fail("unexpected character");
        --depth;
        return v;
    }
};

inline JVal parse(const char* bytes, size_t n) {
    JParser jp{bytes, bytes + n};
    return jp.val();
}

}  // namespace json
}  // namespace wrs
