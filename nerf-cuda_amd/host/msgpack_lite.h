// msgpack_lite.h -- minimal msgpack reader for the reference's snapshot files.
//
// Replaces nlohmann::json::from_msgpack as used by NerfRender::load_network_config
// (reference src/nerf_render.cu:83-88).  Numeric arrays (the 12.2 M-element
// "params" and the 2 M-element "density_grid" are plain msgpack arrays of
// numbers) are stored as flat float vectors instead of
// one node per element.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mpk {

struct Value {
  enum Type { Nil, Bool, Number, String, Array, NumArray, Map, Bin } type = Nil;
  bool b = false;
  double num = 0.0;
  std::string str;                 // String / Bin
  std::vector<Value> arr;          // Array (mixed)
  std::vector<float> nums;         // NumArray (all elements numeric)
  std::map<std::string, Value> map;

  bool contains(const std::string& k) const { return type == Map && map.count(k); }
  const Value& at(const std::string& k) const {
    auto it = map.find(k);
    if (type != Map || it == map.end()) throw std::runtime_error("snapshot: missing key '" + k + "'");
    return it->second;
  }
  bool is_number() const { return type == Number; }
  double number(double dflt) const { return type == Number ? num : dflt; }
  template <typename T>
  T value(const std::string& k, T dflt) const {
    if (!contains(k)) return dflt;
    const Value& v = map.at(k);
    if (v.type == Number) return (T)v.num;
    if (v.type == Bool) return (T)v.b;
    return dflt;
  }
  std::string value(const std::string& k, const char* dflt) const {
    if (!contains(k) || map.at(k).type != String) return dflt;
    return map.at(k).str;
  }
  size_t size() const { return type == NumArray ? nums.size() : (type == Array ? arr.size() : map.size()); }
};

class Reader {
 public:
  Reader(const uint8_t* p, size_t n) : p_(p), end_(p + n) {}
  Value parse() { return item(); }

 private:
  const uint8_t* p_;
  const uint8_t* end_;
  int depth_ = 0;  // nesting of containers: hostile input must not recurse the stack away
  static constexpr int kMaxDepth = 64;
  struct Nest {
    int& d;
    explicit Nest(int& depth) : d(depth) {
      if (++d > kMaxDepth) throw std::runtime_error("msgpack: nesting deeper than 64 levels");
    }
    ~Nest() { --d; }
  };
  void need(size_t n) const {
    if ((size_t)(end_ - p_) < n) throw std::runtime_error("msgpack: truncated input");
  }
  template <typename T>
  T be() {
    need(sizeof(T));
    T v = 0;
    for (size_t i = 0; i < sizeof(T); ++i) v = (T)((v << 8) | p_[i]);
    p_ += sizeof(T);
    return v;
  }
  std::string bytes(size_t n) {
    need(n);
    std::string s((const char*)p_, n);
    p_ += n;
    return s;
  }
  Value number(double d) {
    Value v;
    v.type = Value::Number;
    v.num = d;
    return v;
  }
  Value array(size_t n) {
    Nest guard(depth_);
    Value v;
    v.type = Value::NumArray;
    // every element takes at least one byte: a length field larger than what is left is a lie, not a reservation
    if (n > (size_t)(end_ - p_)) throw std::runtime_error("msgpack: array length exceeds the input");
    v.nums.reserve(n);
    bool numeric = true;
    for (size_t i = 0; i < n; ++i) {
      if (numeric) {
        // fast path for the common encodings of numbers
        need(1);
        const uint8_t t = *p_;
        if (t == 0xca) { ++p_; uint32_t u = be<uint32_t>(); float f; std::memcpy(&f, &u, 4); v.nums.push_back(f); continue; }
        if (t == 0xcb) { ++p_; uint64_t u = be<uint64_t>(); double d; std::memcpy(&d, &u, 8); v.nums.push_back((float)d); continue; }
        if (t <= 0x7f) { ++p_; v.nums.push_back((float)t); continue; }
        if (t >= 0xe0) { ++p_; v.nums.push_back((float)(int8_t)t); continue; }
      }
      Value e = item();
      if (numeric && e.type == Value::Number) {
        v.nums.push_back((float)e.num);
      } else {
        if (numeric) {  // demote to a mixed array
          numeric = false;
          v.type = Value::Array;
          for (float f : v.nums) v.arr.push_back(number(f));
          v.nums.clear();
        }
        v.arr.push_back(std::move(e));
      }
    }
    return v;
  }
  Value map(size_t n) {
    Nest guard(depth_);
    Value v;
    v.type = Value::Map;
    if (n > (size_t)(end_ - p_) / 2) throw std::runtime_error("msgpack: map length exceeds the input");
    for (size_t i = 0; i < n; ++i) {
      Value k = item();
      if (k.type != Value::String) throw std::runtime_error("msgpack: non-string map key");
      v.map.emplace(std::move(k.str), item());
    }
    return v;
  }
  Value str(size_t n, Value::Type t = Value::String) {
    Value v;
    v.type = t;
    v.str = bytes(n);
    return v;
  }
  Value item() {
    need(1);
    const uint8_t t = *p_++;
    if (t <= 0x7f) return number(t);
    if (t >= 0xe0) return number((int8_t)t);
    if ((t & 0xf0) == 0x80) return map(t & 0x0f);
    if ((t & 0xf0) == 0x90) return array(t & 0x0f);
    if ((t & 0xe0) == 0xa0) return str(t & 0x1f);
    switch (t) {
      case 0xc0: return Value{};
      case 0xc2: case 0xc3: { Value v; v.type = Value::Bool; v.b = t == 0xc3; return v; }
      case 0xc4: return str(be<uint8_t>(), Value::Bin);
      case 0xc5: return str(be<uint16_t>(), Value::Bin);
      case 0xc6: return str(be<uint32_t>(), Value::Bin);
      case 0xca: { uint32_t u = be<uint32_t>(); float f; std::memcpy(&f, &u, 4); return number(f); }
      case 0xcb: { uint64_t u = be<uint64_t>(); double d; std::memcpy(&d, &u, 8); return number(d); }
      case 0xcc: return number(be<uint8_t>());
      case 0xcd: return number(be<uint16_t>());
      case 0xce: return number(be<uint32_t>());
      case 0xcf: return number((double)be<uint64_t>());
      case 0xd0: return number((int8_t)be<uint8_t>());
      case 0xd1: return number((int16_t)be<uint16_t>());
      case 0xd2: return number((int32_t)be<uint32_t>());
      case 0xd3: return number((double)(int64_t)be<uint64_t>());
      case 0xd9: return str(be<uint8_t>());
      case 0xda: return str(be<uint16_t>());
      case 0xdb: return str(be<uint32_t>());
      case 0xdc: return array(be<uint16_t>());
      case 0xdd: return array(be<uint32_t>());
      case 0xde: return map(be<uint16_t>());
      case 0xdf: return map(be<uint32_t>());
      default: throw std::runtime_error("msgpack: unsupported type byte");
    }
  }
};

}  // namespace mpk
