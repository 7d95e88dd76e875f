// parser_fuzz.cpp -- sanitizer harness for the host code that faces untrusted bytes: the msgpack snapshot reader
// (msgpack_lite.h + NerfRender::load_snapshot / reset_network, both snapshot layouts), the JSON reader (json_lite.h +
// load_camera_path) and the PNG writer (png_lite.h).  Built with -fsanitize=address,undefined by `make asan`
// (host only: NerfRender(-1) creates no device context); any sanitizer report ends the process with a non-zero status.
// The reference has no such target (SURVEY 5 "Race detection / sanitizers": none) -- and no parser of its own either.
//   parser_fuzz file <snapshot.msgpack | transforms.json>     parse one file (exceptions are fine, reports are not)
//   parser_fuzz fuzz <seed> <cases> <tmpdir>                  seeded structural fuzz of both parsers
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "json_lite.h"
#include "nerf_render.h"
#include "png_lite.h"

namespace {

// ---- a tiny msgpack writer (test side only)
struct Pack {
  std::string s;
  void u8(unsigned v) { s.push_back((char)v); }
  void be32(uint32_t v) { for (int i = 3; i >= 0; --i) u8((v >> (8 * i)) & 255); }
  void map(uint32_t n) { if (n < 16) u8(0x80 | n); else { u8(0xdf); be32(n); } }
  void arr(uint32_t n) { if (n < 16) u8(0x90 | n); else { u8(0xdd); be32(n); } }
  void str(const char* t) { const size_t n = std::strlen(t); if (n < 32) u8(0xa0 | n); else { u8(0xdb); be32((uint32_t)n); } s.append(t, n); }
  void f32(float f) { uint32_t u; std::memcpy(&u, &f, 4); u8(0xca); be32(u); }
  void uint(uint32_t v) { if (v < 128) u8(v); else { u8(0xce); be32(v); } }
  void bin(const std::string& b) { u8(0xc6); be32((uint32_t)b.size()); s += b; }
};

// a small but complete snapshot in the reference's array form (or with binary blobs / in instant-ngp's layout)
std::string make_snapshot(std::mt19937& rng, int flavour) {
  const uint32_t H = 4, L = 2, F = 2, log2T = 4, base = 2, Wn = 16;
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  // param count of this shape: density 16*? ... let the loader complain when it is off -- half of the cases are meant to
  const uint32_t n_grid = 2 * 16 * F, n_mlp = (16 * Wn + Wn * 16) + (32 * Wn + Wn * 16);
  const uint32_t n_params = n_grid + n_mlp + (rng() % 4 == 0 ? rng() % 7 : 0);
  Pack p;
  p.map(5);
  p.str("encoding"); p.map(6);
  p.str("otype"); p.str("HashGrid");
  p.str("n_levels"); p.uint(L);
  p.str("n_features_per_level"); p.uint(F);
  p.str("log2_hashmap_size"); p.uint(log2T);
  p.str("base_resolution"); p.uint(base);
  p.str("per_level_scale"); p.f32(2.0f);
  p.str("network"); p.map(4);
  p.str("otype"); p.str("FullyFusedMLP"); p.str("n_neurons"); p.uint(Wn); p.str("n_hidden_layers"); p.uint(1); p.str("activation"); p.str("ReLU");
  p.str("rgb_network"); p.map(3);
  p.str("otype"); p.str("FullyFusedMLP"); p.str("n_neurons"); p.uint(Wn); p.str("n_hidden_layers"); p.uint(1);
  p.str("dir_encoding"); p.map(2);
  p.str("otype"); p.str("Composite");
  p.str("nested"); p.arr(1); p.map(3); p.str("n_dims_to_encode"); p.uint(3); p.str("otype"); p.str("SphericalHarmonics"); p.str("degree"); p.uint(4);
  p.str("snapshot");
  if (flavour == 2) {  // instant-ngp's own layout
    p.map(5);
    p.str("density_grid_size"); p.uint(H);
    p.str("nerf"); p.map(1); p.str("aabb_scale"); p.uint(2);
    std::string g((size_t)2 * H * H * H * 2, '\0'), w((size_t)n_params * 2, '\0');
    for (char& c : g) c = (char)(rng() & 0x3f);
    for (char& c : w) c = (char)(rng() & 0x3f);
    p.str("density_grid_binary"); p.bin(g);
    p.str("params_binary"); p.bin(w);
    p.str("aabb"); p.map(2); p.str("min"); p.arr(3); for (int i = 0; i < 3; ++i) p.f32(0.f); p.str("max"); p.arr(3); for (int i = 0; i < 3; ++i) p.f32(1.f);
  } else {
    p.map(7);
    p.str("aabb"); p.arr(6); for (int i = 0; i < 6; ++i) p.f32(i < 3 ? -1.f : 1.f);
    p.str("bound"); p.f32(1.0f);
    p.str("cascade"); p.uint(1);
    p.str("density_grid_size"); p.uint(H);
    p.str("mean_density"); p.f32(0.01f);
    if (flavour == 1) {
      std::string g((size_t)H * H * H * 4, '\0'), w((size_t)n_params * 4, '\0');
      for (size_t i = 0; i + 4 <= g.size(); i += 4) { const float f = U(rng); std::memcpy(&g[i], &f, 4); }
      for (size_t i = 0; i + 4 <= w.size(); i += 4) { const float f = U(rng); std::memcpy(&w[i], &f, 4); }
      p.str("density_grid_binary"); p.bin(g);
      p.str("params_binary"); p.bin(w);
      // (the *_type keys are left out: "__half" is assumed -> a size check has to catch the mismatch or accept it)
    } else {
      p.str("density_grid"); p.arr(H * H * H); for (uint32_t i = 0; i < H * H * H; ++i) p.f32(U(rng));
      p.str("params"); p.arr(n_params); for (uint32_t i = 0; i < n_params; ++i) p.f32(U(rng));
    }
  }
  return p.s;
}

std::string make_transforms(std::mt19937& rng) {
  std::string s = "{\"camera_angle_x\": 0.6911, \"w\": 800, \"h\": 800, \"frames\": [";
  const int n = 1 + (int)(rng() % 3);
  for (int f = 0; f < n; ++f) {
    s += f ? ", " : "";
    s += "{\"file_path\": \"./r_\\u00e9" + std::to_string(f) + "\", \"transform_matrix\": [";
    for (int r = 0; r < 4; ++r) {
      s += r ? ", [" : "[";
      for (int c = 0; c < 4; ++c) s += (c ? ", " : "") + std::to_string((int)(rng() % 2000) / 1000.0 - 1.0) + (rng() % 5 == 0 ? "e-1" : "");
      s += "]";
    }
    s += "]}";
  }
  return s + "]}";
}

void mutate(std::string& b, std::mt19937& rng) {
  if (b.empty()) return;
  const int ops = 1 + (int)(rng() % 4);
  for (int k = 0; k < ops; ++k) {
    const size_t at = rng() % b.size();
    switch (rng() % 7) {
      case 0: b[at] = (char)(rng() & 255); break;                                     // a random byte
      case 1: b[at] ^= (char)(1u << (rng() % 8)); break;                               // a flipped bit
      case 2: b.resize(at); break;                                                    // truncation
      case 3: b.insert(at, std::string(1 + rng() % 8, (char)(rng() & 255))); break;    // inserted bytes
      case 4: b.erase(at, 1 + rng() % 16); break;                                      // removed bytes
      case 5: {                                                                        // a huge length field
        static const unsigned char big[5] = {0xdd, 0xff, 0xff, 0xff, 0xff};
        b.replace(at, std::min<size_t>(5, b.size() - at), std::string((const char*)big, 5));
      } break;
      default: {                                                                       // a copied slice somewhere else
        const size_t from = rng() % b.size(), n = std::min<size_t>(1 + rng() % 32, b.size() - from);
        b.insert(at, b.substr(from, n));
      }
    }
    if (b.empty()) return;
  }
}

// 0: parsed and loaded; 1: rejected with an exception (both are fine)
int try_snapshot(const std::string& path) {
  try {
    ngp::NerfRender r(-1);  // host-only instance
    r.load_snapshot(path);
    r.reset_network();
    uint64_t expect = 0;
    (void)nrf_expected_n_params(&r.model_desc(), &expect);
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}
int try_transforms(const std::string& path) {
  try {
    std::vector<ngp::Camera> cams;
    std::vector<ngp::Matrix4f> poses;
    ngp::load_camera_path(path, 64, 48, cams, poses);
    return cams.size() == poses.size() ? 0 : 2;
  } catch (const std::exception&) {
    return 1;
  }
}
void write_file(const std::string& path, const std::string& bytes) {
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  f.write(bytes.data(), (std::streamsize)bytes.size());
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 3 && std::strcmp(argv[1], "file") == 0) {
    const std::string path = argv[2];
    const bool json = path.size() > 5 && path.substr(path.size() - 5) == ".json";
    const int rc = json ? try_transforms(path) : try_snapshot(path);
    std::printf("%s: %s\n", path.c_str(), rc == 0 ? "accepted" : "rejected");
    return 0;
  }
  if (argc >= 5 && std::strcmp(argv[1], "fuzz") == 0) {
    std::mt19937 rng((uint32_t)std::strtoul(argv[2], nullptr, 10));
    const long cases = std::strtol(argv[3], nullptr, 10);
    const std::string dir = argv[4];
    long accepted[2] = {0, 0}, rejected[2] = {0, 0};
    for (long i = 0; i < cases; ++i) {
      const bool json = i % 3 == 2;
      std::string bytes = json ? make_transforms(rng) : make_snapshot(rng, (int)(i % 3 == 0 ? rng() % 2 : 2));
      if (i % 8 != 0) mutate(bytes, rng);  // every eighth case stays as generated
      const std::string path = dir + (json ? "/case.json" : "/case.msgpack");
      write_file(path, bytes);
      const int rc = json ? try_transforms(path) : try_snapshot(path);
      if (rc == 2) { std::fprintf(stderr, "inconsistent camera path\n"); return 3; }
      (rc == 0 ? accepted : rejected)[json ? 1 : 0]++;
    }
    // the PNG writer on a few shapes (odd sizes, one pixel, several channel counts)
    for (int w : {1, 3, 17}) for (int h : {1, 2, 9}) for (int ch : {1, 3}) {
      std::vector<unsigned char> px((size_t)w * h * ch);
      for (auto& v : px) v = (unsigned char)(rng() & 255);
      pnglite::write((dir + "/case.png").c_str(), w, h, ch, px.data());
    }
    std::printf("fuzz: %ld cases; snapshots accepted %ld rejected %ld; camera paths accepted %ld rejected %ld\n", cases, accepted[0], rejected[0],
                accepted[1], rejected[1]);
    return (accepted[0] > 0 && accepted[1] > 0 && rejected[0] > 0 && rejected[1] > 0) ? 0 : 4;  // a fuzz that only ever rejects (or accepts) tests nothing
  }
  std::fprintf(stderr, "usage: parser_fuzz file <path> | parser_fuzz fuzz <seed> <cases> <tmpdir>\n");
  return 2;
}
