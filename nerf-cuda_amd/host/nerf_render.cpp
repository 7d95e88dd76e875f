// nerf_render.cpp -- ngp::NerfRender on top of include/nerfhip.h (no HIP in this file).
// Mirrors reference src/nerf_render.cu: load_network_config :66-91, reload_network_from_file
// :93-109, reset_network :111-184, set_resolution :186-236, render_frame :238-367,
// load_snapshot :431-473.
#include "nerf_render.h"

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
}  // namespace

void NerfRender::check(int rc, const char* what) const {
  if (rc != NRF_OK) throw std::runtime_error{std::string(what) + ": " + nrf_last_error()};
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
  const mpk::Value& aabb = snapshot.at("aabb");
  if (aabb.type != mpk::Value::NumArray || aabb.nums.size() != 6) throw std::runtime_error{"snapshot.aabb must hold 6 numbers"};
  nrf_model_desc& d = m_desc;
  d = nrf_model_desc{};
  d.abi_version = NRF_ABI_VERSION;
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
    const mpk::Value& b = snapshot.at(bkey.c_str());
    if (b.type != mpk::Value::Bin) throw std::runtime_error{"snapshot." + bkey + " must be a binary blob"};
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
  us_image.assign((size_t)res[0] * res[1] * 3, 0);  // nerf_render.cu:232-235
  us_depth.assign((size_t)res[0] * res[1], 0);
}

Image NerfRender::render_frame(Camera cam, Matrix4f pos) {
  if (!m_have_network) throw std::runtime_error{"render_frame: no network loaded"};
  const float c4[4] = {cam.fl_x, cam.fl_y, cam.cx, cam.cy};
  // every member renders its strips concurrently on its own stream; the shards meet on the first device
  // (the reference's threads + D2H + de-interleave loop, nerf_render.cu:252-362)
  check(nrf_group_render_views(m_group, 1, c4, pos.m, nullptr), "nrf_group_render_views");
  check(nrf_group_read_view_u8(m_group, 0, us_image.data(), us_depth.data()), "nrf_group_read_view_u8");
  return Image(resolution[0], resolution[1], us_image.data(), us_depth.data());
}

std::vector<Image> NerfRender::render_frames(const std::vector<Camera>& cams, const std::vector<Matrix4f>& poses) {
  if (!m_have_network) throw std::runtime_error{"render_frames: no network loaded"};
  if (cams.size() != poses.size()) throw std::runtime_error{"render_frames: cams and poses differ in length"};
  const int W = resolution[0], H = resolution[1], n = (int)cams.size();
  const size_t px = (size_t)W * H;
  m_batch_image.resize(px * 3 * (size_t)n);
  m_batch_depth.resize(px * (size_t)n);
  std::vector<Image> out;
  if (n > 0) {
    std::vector<float> c4((size_t)4 * n), p16((size_t)16 * n);
    for (int v = 0; v < n; ++v) {
      const float c[4] = {cams[v].fl_x, cams[v].fl_y, cams[v].cx, cams[v].cy};
      std::memcpy(&c4[4 * (size_t)v], c, sizeof(c));
      std::memcpy(&p16[16 * (size_t)v], poses[v].m, sizeof(poses[v].m));
    }
    check(nrf_group_render_views(m_group, n, c4.data(), p16.data(), nullptr), "nrf_group_render_views");
    for (int v = 0; v < n; ++v)
      check(nrf_group_read_view_u8(m_group, v, m_batch_image.data() + px * 3 * v, m_batch_depth.data() + px * v),
            "nrf_group_read_view_u8");
  }
  for (int v = 0; v < n; ++v) out.emplace_back(W, H, m_batch_image.data() + px * 3 * v, m_batch_depth.data() + px * v);
  return out;
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
