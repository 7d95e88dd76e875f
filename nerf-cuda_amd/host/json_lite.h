// json_lite.h -- minimal JSON reader into mpk::Value (the tree msgpack_lite.h builds), for camera paths in
// `transforms.json` files (NeRF-synthetic / instant-ngp layout).  The reference reads .json network configs with
// nlohmann::json (src/nerf_render.cu:30-44,66-91); nothing else of that library is needed here.
#pragma once
#include <cctype>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "msgpack_lite.h"

namespace mpk {

class JsonReader {
 public:
  JsonReader(const char* p, size_t n) : p_(p), end_(p + n) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != end_) throw std::runtime_error("json: trailing characters");
    return v;
  }

 private:
  const char* p_;
  const char* end_;
  int depth_ = 0;
  void ws() {
    while (p_ < end_ && std::isspace((unsigned char)*p_)) ++p_;
  }
  char peek() {
    ws();
    if (p_ >= end_) throw std::runtime_error("json: truncated input");
    return *p_;
  }
  void expect(char c) {
    if (peek() != c) throw std::runtime_error(std::string("json: expected '") + c + "'");
    ++p_;
  }
  bool literal(const char* s) {
    const size_t n = std::strlen(s);
    if ((size_t)(end_ - p_) >= n && std::memcmp(p_, s, n) == 0) { p_ += n; return true; }
    return false;
  }
  std::string string() {
    expect('"');
    std::string out;
    while (true) {
      if (p_ >= end_) throw std::runtime_error("json: unterminated string");
      const char c = *p_++;
      if (c == '"') break;
      if (c == '\\') {
        if (p_ >= end_) throw std::runtime_error("json: unterminated escape");
        const char e = *p_++;
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // \uXXXX -> UTF-8 (surrogate pairs are not joined: each half becomes its own 3-byte sequence)
            if (end_ - p_ < 4) throw std::runtime_error("json: bad \\u escape");
            unsigned cp = 0;
            for (int i = 0; i < 4; ++i) {
              const char h = p_[i];
              const int dgt = h >= '0' && h <= '9' ? h - '0' : (h >= 'a' && h <= 'f' ? h - 'a' + 10 : (h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1));
              if (dgt < 0) throw std::runtime_error("json: bad \\u escape");
              cp = cp * 16 + (unsigned)dgt;
            }
            p_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
          } break;
          default: out += e;  // \" \\ \/
        }
      } else {
        out += c;
      }
    }
    return out;
  }
  Value value() {
    if (++depth_ > 64) throw std::runtime_error("json: nesting deeper than 64 levels");
    Value v;
    const char c = peek();
    if (c == '{') {
      ++p_;
      v.type = Value::Map;
      if (peek() == '}') { ++p_; --depth_; return v; }
      while (true) {
        std::string k = string();
        expect(':');
        v.map.emplace(std::move(k), value());
        if (peek() == ',') { ++p_; continue; }
        expect('}');
        break;
      }
    } else if (c == '[') {
      ++p_;
      v.type = Value::NumArray;  // demoted to Array by the first non-number
      if (peek() == ']') { ++p_; --depth_; return v; }
      while (true) {
        Value e = value();
        if (v.type == Value::NumArray && e.type == Value::Number) {
          v.nums.push_back((float)e.num);
        } else {
          if (v.type == Value::NumArray) {
            v.type = Value::Array;
            for (float f : v.nums) { Value n; n.type = Value::Number; n.num = f; v.arr.push_back(n); }
            v.nums.clear();
          }
          v.arr.push_back(std::move(e));
        }
        if (peek() == ',') { ++p_; continue; }
        expect(']');
        break;
      }
    } else if (c == '"') {
      v.type = Value::String;
      v.str = string();
    } else if (literal("true")) {
      v.type = Value::Bool; v.b = true;
    } else if (literal("false")) {
      v.type = Value::Bool; v.b = false;
    } else if (literal("null")) {
      v.type = Value::Nil;
    } else {
      // a JSON number and nothing else: -?digits[.digits][(e|E)[+-]digits] -- strtod alone would also take "inf", "nan",
      // hex floats and leading '+', and a token cut to fit a buffer would silently split a long number in two
      const char* q = p_;
      auto digits = [&] {
        const char* b = q;
        while (q < end_ && *q >= '0' && *q <= '9') ++q;
        return q > b;
      };
      if (q < end_ && *q == '-') ++q;
      if (!digits()) throw std::runtime_error("json: unexpected character");
      if (q < end_ && *q == '.') {
        ++q;
        if (!digits()) throw std::runtime_error("json: digits expected after the decimal point");
      }
      if (q < end_ && (*q == 'e' || *q == 'E')) {
        ++q;
        if (q < end_ && (*q == '+' || *q == '-')) ++q;
        if (!digits()) throw std::runtime_error("json: digits expected in the exponent");
      }
      if (q - p_ > 400) throw std::runtime_error("json: number token longer than 400 characters");
      const std::string tok(p_, (size_t)(q - p_));
      const double d = std::strtod(tok.c_str(), nullptr);
      if (!(d - d == 0.0)) throw std::runtime_error("json: number out of range");  // inf from overflow (nan cannot be written)
      p_ = q;
      v.type = Value::Number;
      v.num = d;
    }
    --depth_;
    return v;
  }
};

}  // namespace mpk
