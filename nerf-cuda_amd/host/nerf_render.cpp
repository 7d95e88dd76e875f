// nerf_render.cpp -- ngp::NerfRender on top of include/nerfhip.h (no HIP in this file).
// Mirrors reference src/nerf_render.cu: load_network_config :66-91, reload_network_from_file
// :93-109, reset_network :111-184, set_resolution :186-236, render_frame :238-367,
// load_snapshot :431-473.
#include "nerf_render.h"

#include "json_lite.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace ngp {

namespace {

std::string to_lower(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
  return s;
}
std::string extension(const std::string& path) {
  const size_t dot = path.find_last_of('.');
  const size_t slash = path.find_last_of('/');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  return to_lower(path.substr(dot + 1));
}
bool exists(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  return (bool)f;
}
uint32_t activation(const mpk::Value& block, const char* key, const char* dflt) {  // T/src/network.cu:41-60
  const std::string s = to_lower(block.value(key, dflt));
  if (s == "none") return NRF_ACT_NONE;
  if (s == "relu") return NRF_ACT_RELU;
  if (s == "exponential") return NRF_ACT_EXPONENTIAL;
  if (s == "sigmoid") return NRF_ACT_SIGMOID;
  if (s == "squareplus") return NRF_ACT_SQUAREPLUS;
  if (s == "softplus") return NRF_ACT_SOFTPLUS;
  if (s == "sine") return NRF_ACT_SINE;
  throw std::runtime_error{"Invalid activation name: " + s};
}
std::string blob_of(const mpk::Value& v, const std::string& what);  // (below: the two binary forms tcnn reads)
}  // namespace

void NerfRender::check(int rc, const char* what) const {
  if (rc != NRF_OK) throw std::runtime_error{std::string(what) + ": " + nrf_last_error()};
}

NerfRender::NerfRender(const std::vector<int>& devices) {
  if (devices.empty()) throw std::runtime_error{"NerfRender: empty device list"};
  check(nrf_group_create((int)devices.size(), devices.data(), &m_group), "nrf_group_create");
  for (int i = 0; i < (int)devices.size(); ++i) m_ctx.push_back(nrf_group_member(m_group, i));
}

NerfRender::NerfRender(int n_gpus) {
  if (n_gpus < 0) return;  // host-only instance (snapshot tooling / CPU tests): no device context
  if (n_gpus == 0) {
    const char* e = std::getenv("NERF_NGPU");
    n_gpus = e ? std::max(1, std::atoi(e)) : 1;
  }
  std::vector<int> devices;
  if (const char* list = std::getenv("NERF_DEVICES")) {  // e.g. "0,0" rehearses two members on one device
    for (const char* p = list; *p;) {
      devices.push_back(std::atoi(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  if ((int)devices.size() != n_gpus) {
    devices.clear();
    for (int gpu = 0; gpu < n_gpus; ++gpu) devices.push_back(gpu);  // nerf_render.cu:49-56: one stream / state per device
  }
  check(nrf_group_create(n_gpus, devices.data(), &m_group), "nrf_group_create");
  for (int i = 0; i < n_gpus; ++i) m_ctx.push_back(nrf_group_member(m_group, i));
}

NerfRender::~NerfRender() {
  if (m_group) nrf_group_destroy(m_group);
}

mpk::Value NerfRender::load_network_config(const std::string& network_config_path) {
  if (!network_config_path.empty()) m_network_config_path = network_config_path;
  std::printf("Loading network config from: %s\n", network_config_path.c_str());
  if (network_config_path.empty() || !exists(network_config_path)) {
    throw std::runtime_error{std::string{"Network config \""} + network_config_path + "\" does not exist."};
  }
  mpk::Value result;
  if (extension(network_config_path) == "msgpack") {
    std::ifstream f{network_config_path, std::ios::in | std::ios::binary};
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    result = mpk::Reader(buf.data(), buf.size()).parse();
  }
  // (.json with parent merging exists in the reference but nothing reaches it: reload_network_from_file
  //  only accepts .msgpack; a non-msgpack path yields an empty config exactly as there)
  return result;
}

void NerfRender::reload_network_from_file(const std::string& network_config_path) {
  if (network_config_path.empty()) return;
  m_network_config_path = network_config_path;
  if (extension(m_network_config_path) == "msgpack") {
    load_snapshot(network_config_path);
    reset_network();
    // NerfNetwork::deserialize (nerf_network.h:424-443): size check + fp32 -> fp16 + upload
    if (m_group) check(nrf_group_load_model(m_group, &m_desc), "Can't set params");
    // a snapshot without a density grid (the reference's load_snapshot would throw on the missing key,
    // nerf_render.cu:447): evaluate one from the network
    if (m_group && m_density_grid.empty()) generate_density_grid();
  } else {
    throw std::runtime_error{"Input file with wrong extension!"};
  }
}

void NerfRender::load_snapshot(const std::string& filepath_string) {
  std::printf("Reading snapshot\n");
  mpk::Value config = load_network_config(filepath_string);
  if (!config.contains("snapshot")) {
    throw std::runtime_error{"File " + filepath_string + " does not contain a snapshot."};
  }
  const mpk::Value& snapshot = config.at("snapshot");
  nrf_model_desc& d = m_desc;
  d = nrf_model_desc{};
  d.abi_version = NRF_ABI_VERSION;
  d.gather_copy_budget_mb = m_gather_copy_budget_mb;
  m_ngp_per_level_scale = 0.0f;
  m_ngp_rgb_sigmoid = false;
  // instant-ngp's own snapshot layout (the reference reads only its array form): see load_ngp_snapshot
  if (snapshot.contains("nerf") || (snapshot.contains("aabb") && snapshot.at("aabb").type == mpk::Value::Map)) {
    load_ngp_snapshot(config);
    m_network_config_path = filepath_string;
    m_network_config = std::move(config);
    m_have_snapshot = true;
    return;
  }
  const mpk::Value& aabb = snapshot.at("aabb");
  if (aabb.type != mpk::Value::NumArray || aabb.nums.size() != 6) throw std::runtime_error{"snapshot.aabb must hold 6 numbers"};
  for (int i = 0; i < 6; ++i) d.aabb[i] = aabb.nums[i];
  d.bound = snapshot.value("bound", 1.0f);                       // member defaults nerf_render.h:55-67
  d.scale = snapshot.value("scale", 0.33f);
  d.cascade = (uint32_t)snapshot.value("cascade", 1);
  d.density_grid_size = (uint32_t)snapshot.value("density_grid_size", 128);
  d.mean_density = snapshot.value("mean_density", 1.e-4f);
  // The reference reads `params` / `density_grid` as msgpack arrays of numbers (nerf_render.cu:447-466).  Accepted in
  // addition (instant-ngp's snapshot convention): `params_binary` / `density_grid_binary` = the same values, same
  // order, as one raw little-endian blob of `params_type` / `density_grid_type` "__half" (default) or "float".
  auto numbers = [&](const char* key, const char* what) -> std::vector<float> {
    if (snapshot.contains(key)) {
      const mpk::Value& v = snapshot.at(key);
      if (v.type != mpk::Value::NumArray) throw std::runtime_error{std::string("snapshot.") + what + " must be an array of numbers"};
      return v.nums;
    }
    const std::string bkey = std::string(key) + "_binary", tkey = std::string(key) + "_type";
    if (!snapshot.contains(bkey.c_str())) throw std::runtime_error{std::string("snapshot.") + what + " is missing"};
    mpk::Value b;
    b.type = mpk::Value::Bin;
    b.str = blob_of(snapshot.at(bkey.c_str()), "snapshot." + bkey);
    const std::string type = snapshot.value(tkey.c_str(), "__half");
    std::vector<float> out;
    if (type == "float") {
      if (b.str.size() % 4) throw std::runtime_error{"snapshot." + bkey + ": size is not a multiple of 4"};
      out.resize(b.str.size() / 4);
      std::memcpy(out.data(), b.str.data(), b.str.size());
    } else if (type == "__half" || type == "half") {
      if (b.str.size() % 2) throw std::runtime_error{"snapshot." + bkey + ": size is not a multiple of 2"};
      out.resize(b.str.size() / 2);
      for (size_t i = 0; i < out.size(); ++i) {
        uint16_t h;
        std::memcpy(&h, b.str.data() + 2 * i, 2);
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
        uint32_t bits;
        if (e == 0) {
          if (m == 0) bits = sign;
          else {  // subnormal half: normalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; ++sh; }
            bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
          }
        } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
        else bits = sign | ((e + 112u) << 23) | (m << 13);
        std::memcpy(&out[i], &bits, 4);
      }
    } else {
      throw std::runtime_error{"snapshot." + tkey + ": unknown element type '" + type + "'"};
    }
    return out;
  };
  const uint64_t H = d.density_grid_size;
  if (snapshot.contains("density_grid") || snapshot.contains("density_grid_binary")) {
    m_density_grid = numbers("density_grid", "density_grid");
    if (m_density_grid.size() != H * H * H * d.cascade) {
      throw std::runtime_error{"Incompatible number of grid cascades."};
    }
  } else {
    m_density_grid.clear();  // generate_density_grid() after the network is up (reload_network_from_file)
  }
  m_params = numbers("params", "params");
  m_network_config_path = filepath_string;
  m_network_config = std::move(config);
  m_have_snapshot = true;
}

namespace {
float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) bits = sign;
    else {
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 1024u)) { mm <<= 1; ++sh; }
      bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
    }
  } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
  else bits = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}
// A binary value of a snapshot as bytes: tcnn writes its parameters as a nlohmann binary_t (msgpack `bin`,
// gpu_memory_to_json_binary) and reads either that or, from a text JSON, the object {"bytes": [..], "subtype": ..}
// (T/include/tiny-cuda-nn/gpu_memory_json.h:37-72).  Both forms are accepted.
std::string blob_of(const mpk::Value& v, const std::string& what) {
  if (v.type == mpk::Value::Bin) return v.str;
  if (v.type == mpk::Value::Map && v.contains("bytes") && v.at("bytes").type == mpk::Value::NumArray) {
    const std::vector<float>& n = v.at("bytes").nums;
    std::string out(n.size(), '\0');
    for (size_t i = 0; i < n.size(); ++i) out[i] = (char)(unsigned char)((unsigned)n[i] & 0xffu);
    return out;
  }
  throw std::runtime_error{what + ": Invalid json type: must be either binary or object"};  // gpu_memory_json.h:70
}
std::vector<float> blob_to_floats(const std::string& blob, bool is_float, const char* what) {
  std::vector<float> out;
  if (is_float) {
    if (blob.size() % 4) throw std::runtime_error{std::string(what) + ": size is not a multiple of 4"};
    out.resize(blob.size() / 4);
    std::memcpy(out.data(), blob.data(), blob.size());
  } else {
    if (blob.size() % 2) throw std::runtime_error{std::string(what) + ": size is not a multiple of 2"};
    out.resize(blob.size() / 2);
    for (size_t i = 0; i < out.size(); ++i) {
      uint16_t h;
      std::memcpy(&h, blob.data() + 2 * i, 2);
      out[i] = half_bits_to_float(h);
    }
  }
  return out;
}
uint32_t expand_bits(uint32_t v) {  // instant-ngp's morton3D (the reference carries the same helper, render_utils.h:157-170)
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
}  // namespace

// instant-ngp's snapshot layout -> the reference's conventions (same mapping as nerfhip.py ngp_snapshot_to_reference,
// which documents it): bound = aabb_scale / 2, x_ref = x_ngp - 0.5, reference cascade k = instant-ngp cascade k + 1
// (cascade 0 when aabb_scale == 1) max-pooled with its children, Morton -> x-major, mean over instant-ngp's cascade 0,
// per_level_scale from aabb_scale when the file leaves it out, logistic colour activation as the rgb output activation.
void NerfRender::load_ngp_snapshot(const mpk::Value& config) {
  const mpk::Value& snapshot = config.at("snapshot");
  static const mpk::Value empty_map = [] { mpk::Value v; v.type = mpk::Value::Map; return v; }();
  const mpk::Value& nerf = snapshot.contains("nerf") ? snapshot.at("nerf") : empty_map;
  const mpk::Value& dataset = nerf.contains("dataset") ? nerf.at("dataset") : empty_map;
  const uint32_t aabb_scale = nerf.value("aabb_scale", dataset.value("aabb_scale", 1u));
  if (aabb_scale < 1 || (aabb_scale & (aabb_scale - 1))) throw std::runtime_error{"instant-ngp snapshot: aabb_scale must be a power of two"};
  if (dataset.contains("offset")) {
    const mpk::Value& off = dataset.at("offset");
    if (off.type != mpk::Value::NumArray || off.nums.size() != 3) throw std::runtime_error{"instant-ngp snapshot: dataset.offset must hold 3 numbers"};
    for (float v : off.nums)
      if (std::fabs(v - 0.5f) > 1e-6f) throw std::runtime_error{"instant-ngp snapshot: dataset.offset other than 0.5 has no counterpart in the reference"};
  }
  const uint32_t H = snapshot.value("density_grid_size", 128u);
  if (H < 4 || (H & (H - 1))) throw std::runtime_error{"instant-ngp snapshot: density_grid_size must be a power of two (Morton order)"};
  uint32_t n_ngp = 0;
  for (uint32_t v = aabb_scale; v; v >>= 1) ++n_ngp;  // K + 1 cascades
  const uint64_t H3 = (uint64_t)H * H * H, cells = H3 * n_ngp;
  if (!snapshot.contains("density_grid_binary")) throw std::runtime_error{"instant-ngp snapshot: density_grid_binary is missing"};
  const std::string blob = blob_of(snapshot.at("density_grid_binary"), "snapshot.density_grid_binary");
  const std::string gtype = snapshot.value("density_grid_type", blob.size() == 4 * cells ? "float" : "__half");
  const std::vector<float> ngp = blob_to_floats(blob, gtype == "float", "snapshot.density_grid_binary");
  if (ngp.size() < cells) throw std::runtime_error{"Incompatible number of grid cascades."};
  nrf_model_desc& d = m_desc;
  d.bound = (float)aabb_scale / 2.0f;
  d.scale = dataset.value("scale", 0.33f);
  d.cascade = aabb_scale == 1 ? 1u : n_ngp - 1;
  d.density_grid_size = H;
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = 0.5f - d.bound; hi[a] = 0.5f + d.bound; }
  if (snapshot.contains("aabb") && snapshot.at("aabb").type == mpk::Value::Map) {
    const mpk::Value& bb = snapshot.at("aabb");
    if (bb.contains("min") && bb.at("min").type == mpk::Value::NumArray && bb.at("min").nums.size() == 3)
      for (int a = 0; a < 3; ++a) lo[a] = bb.at("min").nums[a];
    if (bb.contains("max") && bb.at("max").type == mpk::Value::NumArray && bb.at("max").nums.size() == 3)
      for (int a = 0; a < 3; ++a) hi[a] = bb.at("max").nums[a];
  }
  for (int a = 0; a < 3; ++a) { d.aabb[a] = lo[a] - 0.5f; d.aabb[a + 3] = hi[a] - 0.5f; }
  {
    double sum = 0.0;  // instant-ngp: mean of max(v, 0) over its cascade 0
    for (uint64_t i = 0; i < H3; ++i) sum += ngp[i] > 0.0f ? (double)ngp[i] : 0.0;
    d.mean_density = (float)(sum / (double)H3);
  }
  m_density_grid.assign((size_t)H3 * d.cascade, 0.0f);
  const uint32_t q = H / 4, half = H / 2;
  for (uint32_t k = 0; k < d.cascade; ++k) {
    const uint32_t c = aabb_scale == 1 ? 0u : k + 1;
    const float* own = ngp.data() + (size_t)c * H3;
    const float* fine = c > 0 ? ngp.data() + (size_t)(c - 1) * H3 : nullptr;
    float* out = m_density_grid.data() + (size_t)k * H3;
    for (uint32_t x = 0; x < H; ++x)
      for (uint32_t y = 0; y < H; ++y)
        for (uint32_t z = 0; z < H; ++z) {
          float v = own[morton3d(x, y, z)];
          if (fine && x >= q && x < q + half && y >= q && y < q + half && z >= q && z < q + half) {
            const uint32_t fx = 2 * (x - q), fy = 2 * (y - q), fz = 2 * (z - q);  // the eight children in the finer cascade
            for (uint32_t c8 = 0; c8 < 8; ++c8) v = std::max(v, fine[morton3d(fx + (c8 & 1), fy + ((c8 >> 1) & 1), fz + ((c8 >> 2) & 1))]);
          }
          out[((size_t)x * H + y) * H + z] = v;
        }
  }
  if (!snapshot.contains("params_binary")) throw std::runtime_error{"instant-ngp snapshot: params_binary is missing"};
  m_params = blob_to_floats(blob_of(snapshot.at("params_binary"), "snapshot.params_binary"), snapshot.value("params_type", "__half") == "float",
                            "snapshot.params_binary");
  // tcnn's Trainer::serialize writes `n_params` beside the blob (T/include/tiny-cuda-nn/trainer.h:267-279): a file whose two disagree is corrupt
  if (snapshot.contains("n_params") && (uint64_t)snapshot.value("n_params", (double)0) != (uint64_t)m_params.size())
    throw std::runtime_error{"snapshot.params_binary holds " + std::to_string(m_params.size()) + " values, snapshot.n_params says " +
                             std::to_string((uint64_t)snapshot.value("n_params", (double)0))};
  // what reset_network cannot derive from the reference's own rules
  static const mpk::Value no_enc = empty_map;
  const mpk::Value& enc = config.contains("encoding") ? config.at("encoding") : no_enc;
  if (!(enc.value("per_level_scale", 0.0f) > 0.0f) && enc.value("n_levels", 16u) > 1) {
    uint32_t base = enc.value("base_resolution", 0u);
    if (!base) base = 1u << (enc.value("log2_hashmap_size", 15u) / 3);
    check(nrf_default_per_level_scale((float)aabb_scale, base, enc.value("n_levels", 16u), &m_ngp_per_level_scale), "per_level_scale");
  }
  const std::string rgb_act = to_lower(nerf.value("rgb_activation", "Logistic"));
  m_ngp_rgb_sigmoid = rgb_act == "logistic" || rgb_act == "sigmoid";
}

// Camera path of a `transforms.json` (NeRF-synthetic / instant-ngp layout): explicit fl_x / fl_y / cx / cy / w / h when
// present, else camera_angle_x (camera_angle_y) over the given resolution; frames[i].transform_matrix = camera-to-world.
void load_camera_path(const std::string& transforms_json, int width, int height, std::vector<Camera>& cams, std::vector<Matrix4f>& poses) {
  std::ifstream f{transforms_json, std::ios::in | std::ios::binary};
  if (!f) throw std::runtime_error{"Camera path \"" + transforms_json + "\" does not exist."};
  const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const mpk::Value t = mpk::JsonReader(text.data(), text.size()).parse();
  if (t.type != mpk::Value::Map || !t.contains("frames") || t.at("frames").type != mpk::Value::Array)
    throw std::runtime_error{"transforms.json: no frames"};
  const double W = width > 0 ? width : t.value("w", 0.0), H = height > 0 ? height : t.value("h", 0.0);
  if (!(W > 0) || !(H > 0)) throw std::runtime_error{"transforms.json carries no resolution (w, h): pass width and height"};
  const double sx = W / t.value("w", W), sy = H / t.value("h", H);
  double fl_x, fl_y;
  if (t.contains("fl_x")) fl_x = t.value("fl_x", 0.0) * sx;
  else if (t.contains("camera_angle_x")) fl_x = 0.5 * W / std::tan(0.5 * t.value("camera_angle_x", 0.0));
  else throw std::runtime_error{"transforms.json: neither fl_x nor camera_angle_x"};
  if (t.contains("fl_y")) fl_y = t.value("fl_y", 0.0) * sy;
  else if (t.contains("camera_angle_y")) fl_y = 0.5 * H / std::tan(0.5 * t.value("camera_angle_y", 0.0));
  else fl_y = fl_x;
  const double cx = t.contains("cx") ? t.value("cx", 0.0) * sx : 0.5 * W, cy = t.contains("cy") ? t.value("cy", 0.0) * sy : 0.5 * H;
  cams.clear();
  poses.clear();
  for (const mpk::Value& fr : t.at("frames").arr) {
    if (!fr.contains("transform_matrix")) throw std::runtime_error{"transforms.json: frame without transform_matrix"};
    const mpk::Value& m = fr.at("transform_matrix");
    Matrix4f pose;
    if (m.type == mpk::Value::Array && m.arr.size() >= 3) {
      for (size_t r = 0; r < m.arr.size() && r < 4; ++r) {
        if (m.arr[r].type != mpk::Value::NumArray || m.arr[r].nums.size() != 4) throw std::runtime_error{"transforms.json: transform_matrix rows must hold 4 numbers"};
        for (int c = 0; c < 4; ++c) pose.m[4 * r + c] = m.arr[r].nums[c];
      }
    } else if (m.type == mpk::Value::NumArray && m.nums.size() == 16) {
      for (int i = 0; i < 16; ++i) pose.m[i] = m.nums[i];
    } else {
      throw std::runtime_error{"transforms.json: transform_matrix must be 4x4"};
    }
    poses.push_back(pose);
    cams.push_back(Camera{(float)fl_x, (float)fl_y, (float)cx, (float)cy});
  }
}

void NerfRender::reset_network() {
  if (!m_have_snapshot) throw std::runtime_error{"reset_network: no snapshot loaded"};
  const mpk::Value& config = m_network_config;
  static const mpk::Value empty_map = [] { mpk::Value v; v.type = mpk::Value::Map; return v; }();
  auto block = [&](const char* k) -> const mpk::Value& { return config.contains(k) ? config.at(k) : empty_map; };
  const mpk::Value& enc = block("encoding");
  const mpk::Value& net = block("network");
  const mpk::Value& dir = block("dir_encoding");
  const mpk::Value& rgb = block("rgb_network");
  nrf_model_desc& d = m_desc;

  // (hash)grid encoding: nerf_render.cu:125-171 and T/include/tiny-cuda-nn/encodings/grid.h:1355-1386
  const std::string otype = to_lower(enc.value("otype", "OneBlob"));
  if (otype.find("grid") == std::string::npos) throw std::runtime_error{"position encoding '" + otype + "' is outside the hot path"};
  const std::string default_type = otype == "tiledgrid" ? "tiled" : (otype == "densegrid" ? "dense" : "hash");
  const std::string gtype = to_lower(enc.value("type", default_type.c_str()));
  d.grid_type = gtype == "hash" ? NRF_GRID_HASH : (gtype == "dense" ? NRF_GRID_DENSE : NRF_GRID_TILED);
  d.n_features_per_level = enc.value("n_features_per_level", 2u);
  if (d.n_features_per_level != 1 && d.n_features_per_level != 2 && d.n_features_per_level != 4 && d.n_features_per_level != 8)
    throw std::runtime_error{"GridEncoding: n_features_per_level must be 1, 2, 4, or 8."};  // grid.h:1403-1411
  {
    const std::string interp = to_lower(enc.value("interpolation", "Linear"));  // grid.h:1383
    if (interp == "linear") d.interpolation = NRF_INTERP_LINEAR;
    else if (interp == "nearest") d.interpolation = NRF_INTERP_NEAREST;
    else if (interp == "smoothstep") d.interpolation = NRF_INTERP_SMOOTHSTEP;
    else throw std::runtime_error{"Invalid interpolation type: " + interp};
  }
  if (enc.contains("n_features") && enc.value("n_features", 0u) > 0) {
    if (enc.contains("n_levels")) throw std::runtime_error{"GridEncoding: may not specify n_features and n_levels simultaneously (one determines the other)"};
    d.n_levels = enc.value("n_features", 0u) / d.n_features_per_level;
  } else {
    d.n_levels = enc.value("n_levels", 16u);
  }
  d.log2_hashmap_size = enc.value("log2_hashmap_size", 15u);
  uint32_t base = enc.value("base_resolution", 0u);
  if (!base) base = 1u << (d.log2_hashmap_size / 3);
  d.base_resolution = base;
  float pls = enc.value("per_level_scale", 0.0f);
  if (pls <= 0.0f && m_ngp_per_level_scale > 0.0f) pls = m_ngp_per_level_scale;  // instant-ngp snapshot: derived from aabb_scale
  if (pls <= 0.0f && d.n_levels > 1) check(nrf_default_per_level_scale(d.bound, base, d.n_levels, &pls), "per_level_scale");
  if (pls <= 0.0f) pls = 2.0f;
  d.per_level_scale = pls;
  std::printf("GridEncoding:  Nmin=%u b=%g F=%u T=2^%u L=%u\n", base, pls, d.n_features_per_level, d.log2_hashmap_size, d.n_levels);

  auto mlp = [&](const mpk::Value& cfg, uint32_t& neurons, uint32_t& hidden, uint32_t& act, uint32_t& out_act) {
    neurons = cfg.value("n_neurons", 128u);  // T/src/network.cu:127-143
    hidden = cfg.value("n_hidden_layers", 5u);
    act = activation(cfg, "activation", "ReLU");
    out_act = activation(cfg, "output_activation", "None");
  };
  uint32_t n1, n2;
  mlp(net, n1, d.density_hidden_layers, d.density_activation, d.density_output_activation);
  mlp(rgb, n2, d.rgb_hidden_layers, d.rgb_activation, d.rgb_output_activation);
  if (n1 != n2) throw std::runtime_error{"density and rgb networks must share n_neurons"};
  if (m_ngp_rgb_sigmoid && d.rgb_output_activation == NRF_ACT_NONE) d.rgb_output_activation = NRF_ACT_SIGMOID;  // instant-ngp's logistic
  d.n_neurons = n1;
  d.density_n_output = net.value("n_output_dims", 16u);                // nerf_network.h:120-122
  d.sigma_activation = activation(net, "sigma_activation", "Exponential");  // nerf_network.h:125

  // dir_encoding: Composite{nested[i] with n_dims_to_encode == 3, ...} or a plain encoding
  const mpk::Value* node = &dir;
  if (to_lower(dir.value("otype", "Composite")) == "composite") {
    node = nullptr;
    if (dir.contains("nested") && dir.at("nested").type == mpk::Value::Array) {
      const auto& nested = dir.at("nested").arr;
      for (const mpk::Value& n : nested)
        if (n.value("n_dims_to_encode", 0u) == 3) { node = &n; break; }
      if (!node && nested.size() == 1 && !nested[0].contains("n_dims_to_encode")) node = &nested[0];
    }
    if (!node) throw std::runtime_error{"dir_encoding must encode all 3 direction dims with one nested encoding"};
  }
  const std::string dt = to_lower(node->value("otype", ""));
  if (dt == "sphericalharmonics") { d.dir_encoding = NRF_DIR_SH; d.sh_degree = node->value("degree", 4u); }
  else if (dt == "frequency") { d.dir_encoding = NRF_DIR_FREQUENCY; d.n_frequencies = node->value("n_frequencies", 12u); }
  else if (dt == "identity") { d.dir_encoding = NRF_DIR_IDENTITY; }
  else throw std::runtime_error{"dir encoding '" + dt + "' is outside the hot path"};

  d.params = m_params.data();
  d.n_params = m_params.size();
  d.density_grid = m_density_grid.empty() ? nullptr : m_density_grid.data();
  d.n_density_grid = m_density_grid.size();
  m_have_network = true;
}

void NerfRender::set_resolution(Vector2i res) {
  resolution = res;
  if (m_group) check(nrf_group_set_resolution(m_group, res[0], res[1]), "nrf_group_set_resolution");
  // (the reference allocates its host image here, nerf_render.cu:232-235; the pinned host planes of this mirror belong
  //  to the C ABI's host-frame slots and are sized by the first render)
}

// render_frame, nerf_render.cu:238-367.  The reference ends with a D2H copy of the float planes and a single-threaded
// quantise / de-interleave loop per GPU (:345-359); here the kernel writes the 8-bit Image itself and the copy engine
// moves it into pinned host memory (nrf_group_render_host_u8): the returned pointers are that memory.
Image NerfRender::render_frame(Camera cam, Matrix4f pos) {
  if (!m_have_network) throw std::runtime_error{"render_frame: no network loaded"};
  const float c4[4] = {cam.fl_x, cam.fl_y, cam.cx, cam.cy};
  nrf_host_frame f{};
  check(nrf_group_render_host_u8(m_group, 1, c4, pos.m, 0, &f), "nrf_group_render_host_u8");
  return Image(f.width, f.height, const_cast<unsigned char*>(f.rgb), const_cast<unsigned char*>(f.depth));
}

int NerfRender::submit_frames(const std::vector<Camera>& cams, const std::vector<Matrix4f>& poses, bool rgb_only) {
  if (!m_have_network) throw std::runtime_error{"render_frames: no network loaded"};
  if (cams.size() != poses.size()) throw std::runtime_error{"render_frames: cams and poses differ in length"};
  if (cams.empty()) throw std::runtime_error{"render_frames: no camera"};
  const int n = (int)cams.size();
  std::vector<float> c4((size_t)4 * n), p16((size_t)16 * n);
  for (int v = 0; v < n; ++v) {
    const float c[4] = {cams[v].fl_x, cams[v].fl_y, cams[v].cx, cams[v].cy};
    std::memcpy(&c4[4 * (size_t)v], c, sizeof(c));
    std::memcpy(&p16[16 * (size_t)v], poses[v].m, sizeof(poses[v].m));
  }
  int ticket = -1;
  check(nrf_group_submit_host_u8(m_group, n, c4.data(), p16.data(), rgb_only ? NRF_HOST_RGB_ONLY : 0, &ticket), "nrf_group_submit_host_u8");
  return ticket;
}

std::vector<Image> NerfRender::wait_frames(int ticket) {
  nrf_host_frame f{};
  check(nrf_group_wait_host_u8(m_group, ticket, &f), "nrf_group_wait_host_u8");
  m_last_wait_render_ms = f.render_ms;
  std::vector<Image> out;
  out.reserve((size_t)f.n_views);
  for (int v = 0; v < f.n_views; ++v)
    out.emplace_back(f.width, f.height, const_cast<unsigned char*>(f.rgb) + (size_t)v * f.view_stride_px * 3,
                     f.depth ? const_cast<unsigned char*>(f.depth) + (size_t)v * f.view_stride_px : nullptr);
  return out;
}

std::vector<Image> NerfRender::render_frames(const std::vector<Camera>& cams, const std::vector<Matrix4f>& poses) {
  if (cams.empty() && poses.empty()) {
    if (!m_have_network) throw std::runtime_error{"render_frames: no network loaded"};
    return {};
  }
  return wait_frames(submit_frames(cams, poses));
}

void NerfRender::generate_rays(Camera cam, Matrix4f pos, int threadid) {
  // reference nerf_render.cu:369-386 fills per-GPU device buffers rays_o / rays_d; the fused kernel
  // never materialises them, so this entry point produces host copies through the stage kernel.
  (void)threadid;
  if (!m_have_network) throw std::runtime_error{"generate_rays: no network loaded"};
  const float c4[4] = {cam.fl_x, cam.fl_y, cam.cx, cam.cy};
  const size_t n = (size_t)resolution[0] * resolution[1];
  m_rays_o.resize(3 * n);
  m_rays_d.resize(3 * n);
  check(nrf_generate_rays_host(m_ctx[0], c4, pos.m, m_rays_o.data(), m_rays_d.data(), nullptr, nullptr), "nrf_generate_rays_host");
}

void NerfRender::generate_density_grid() {
  // reference nerf_render.cu:388-429 (dead there: the density query is commented out at :415); completed behind
  // nrf_generate_density_grid with the reference's constants: decay 0.95 (:392), start value 1/64 (:393), scale
  // 0.001691 (:417).  16 passes: (1/64) * 0.95^16 < 0.01, so a cell the network finds empty falls below the march's
  // threshold min(0.01, mean_density) (render_utils.h:560) whenever the scene's mean density is at least 0.01.
  if (!m_have_network || !m_group) throw std::runtime_error{"generate_density_grid: no network loaded"};
  float mean = 0.0f;
  for (nrf_context* ctx : m_ctx) check(nrf_generate_density_grid(ctx, 16, 0.95f, &mean), "nrf_generate_density_grid");
  m_desc.mean_density = mean;
  std::printf("density grid generated from the network: mean_density %g\n", mean);
}

nrf_stats NerfRender::last_stats(int gpu) const {
  nrf_stats s{};
  check(nrf_get_stats(m_ctx.at(gpu), &s), "nrf_get_stats");
  return s;
}

}  // namespace ngp
