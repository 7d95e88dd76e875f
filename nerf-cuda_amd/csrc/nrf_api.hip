// nrf_api.hip -- the C ABI of include/nerfhip.h on top of the gfx950 kernels.
//
// Host-side mirror of what ngp::NerfRender does around its kernels
// (R/src/nerf_render.cu): model upload (load_snapshot + reset_network +
// NerfNetwork::deserialize), buffer allocation (set_resolution) and one
// kernel launch per frame (render_frame).  No CPU compute path exists here:
// without a gfx950 device nrf_create fails.

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "nrf_device.h"
#include "nrf_generic.h"
#include "nrf_launch.h"

using namespace nrf;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return NRF_E_HIP;
}
#define HIP_TRY(expr)                            \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return hip_fail(_e, #expr); \
  } while (0)

inline uint32_t next_multiple(uint32_t v, uint32_t d) { return (v + d - 1) / d * d; }

// T/include/tiny-cuda-nn/encodings/grid.h:899-931 (ctor) and :186-190 (kernel): the level geometry is
// computed once, on the host, with libm's exp2f/log2f, and handed to every kernel as constants.
int compute_level_table(const nrf_model_desc& d, nrf_level_table& t) {
  if (d.n_levels == 0 || d.n_levels > 16) return fail(NRF_E_UNSUPPORTED, "n_levels must be 1..16");
  if (d.log2_hashmap_size > 31) return fail(NRF_E_INVALID, "log2_hashmap_size must be <= 31");
  std::memset(&t, 0, sizeof(t));
  t.n_levels = d.n_levels;
  const float log2_pls = std::log2(d.per_level_scale);
  uint32_t offset = 0;
  for (uint32_t i = 0; i < d.n_levels; ++i) {
    const float scale = exp2f((float)i * log2_pls) * (float)d.base_resolution - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
    uint32_t params = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
    params = next_multiple(params, 8u);
    if (d.grid_type == NRF_GRID_TILED) {
      const uint32_t b3 = d.base_resolution * d.base_resolution * d.base_resolution;
      params = params < b3 ? params : b3;
    } else if (d.grid_type == NRF_GRID_HASH) {
      const uint32_t T = 1u << d.log2_hashmap_size;
      params = params < T ? params : T;
    } else if (d.grid_type != NRF_GRID_DENSE) {
      return fail(NRF_E_INVALID, "GridEncoding: invalid grid type");
    }
    t.offset[i] = offset;
    t.resolution[i] = res;
    t.scale[i] = scale;
    offset += params;
  }
  t.offset[d.n_levels] = offset;
  return NRF_OK;
}

uint32_t dir_raw_width(const nrf_model_desc& d) {
  switch (d.dir_encoding) {
    case NRF_DIR_SH: return d.sh_degree * d.sh_degree;
    case NRF_DIR_FREQUENCY: return 6 * d.n_frequencies;
    case NRF_DIR_IDENTITY: return 3;
    default: return 0;
  }
}

// n_params of NerfNetwork (nerf_network.h:273-291): density MLP | rgb MLP | grid | dir enc (0)
int expected_params(const nrf_model_desc& d, const nrf_level_table& t, uint64_t& n) {
  const uint32_t raw = dir_raw_width(d);
  if (raw == 0) return fail(NRF_E_UNSUPPORTED, "unknown dir encoding");
  const uint64_t Wn = d.n_neurons;
  const uint64_t feat = next_multiple(d.n_levels * d.n_features_per_level, 16u);
  const uint64_t rgb_in = next_multiple(next_multiple(raw, 16u) + 16u, 16u);
  if (d.density_hidden_layers < 1 || d.rgb_hidden_layers < 1)
    return fail(NRF_E_INVALID, "FullyFusedMLP requires at least 1 hidden layer (3 layers in total).");  // fully_fused_mlp.cu:653-655
  auto mlp = [&](uint64_t in, uint64_t hidden) { return in * Wn + (hidden - 1) * Wn * Wn + Wn * 16; };
  n = mlp(feat, d.density_hidden_layers) + mlp(rgb_in, d.rgb_hidden_layers) +
      (uint64_t)t.offset[d.n_levels] * d.n_features_per_level;
  return NRF_OK;
}

// fp16 weight fragments for v_mfma_f32_16x16x32_f16 (see nrf_device.h mlp_tiles):
// fragment f, lane l, element j  =  W[16m + (l&15)][kmap(s, l>>4, j)]
//   input layers   kmap(g,j) = 2(4(j>>1) + g) + (j&1)      (density: lane group g holds levels g, 4+g, 8+g, 12+g)
//                  kmap(g,j) = j<4 ? 4g+j : 16+4g+(j-4)    (rgb: [density out | dir enc])
//   hidden->next   kmap(s,g,j) = 16(2s + (j>>2)) + 4g + (j&3)   (a D fragment re-used in-lane as B)
void pack_fragments(const std::vector<_Float16>& w16, uint32_t rgb_in, std::vector<_Float16>& frags) {
  frags.assign((size_t)N_FRAGS_WIDE_ALL * 64 * 8, (_Float16)0.0f);
  const _Float16* D0 = w16.data();              // [64][32]
  const _Float16* D1 = D0 + 64 * 32;            // [16][64]
  const _Float16* R0 = D1 + 16 * 64;            // [64][rgb_in]: 32 for a 16-wide direction encoding, up to 96 (wide instance)
  const _Float16* R1 = R0 + 64 * (size_t)rgb_in;  // [64][64]
  const _Float16* R2 = R1 + 64 * 64;            // [16][64]
  auto khid = [](int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); };
  auto put = [&](int f, const _Float16* Wm, int in, int m, auto kmap) {
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int k = kmap(l >> 4, j);
        frags[((size_t)f * 64 + l) * 8 + j] = k < in ? Wm[(size_t)(16 * m + (l & 15)) * in + k] : (_Float16)0.0f;
      }
  };
  for (int m = 0; m < 4; ++m) put(FRAG_D0 + m, D0, 32, m, [](int g, int j) { return 2 * (4 * (j >> 1) + g) + (j & 1); });
  for (int m = 0; m < 4; ++m) put(FRAG_D0_NATURAL + m, D0, 32, m, [](int g, int j) { return 8 * g + j; });
  for (int s = 0; s < 2; ++s) put(FRAG_D1 + s, D1, 64, 0, [&](int g, int j) { return khid(s, g, j); });
  for (int m = 0; m < 4; ++m) put(FRAG_R0 + m, R0, (int)rgb_in, m, [](int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); });
  for (int s = 1; s < RK_WIDE; ++s)  // wide instance: the columns beyond the first 32, natural order; zero beyond rgb_in
    for (int m = 0; m < 4; ++m) put(FRAG_R0X + 4 * (s - 1) + m, R0, (int)rgb_in, m, [&](int g, int j) { return 32 * s + 8 * g + j; });
  for (int m = 0; m < 4; ++m)
    for (int s = 0; s < 2; ++s) put(FRAG_R1 + 2 * m + s, R1, 64, m, [&](int g, int j) { return khid(s, g, j); });
  for (int s = 0; s < 2; ++s) put(FRAG_R2 + s, R2, 64, 0, [&](int g, int j) { return khid(s, g, j); });
}

// GRID instances (nrf_render.h grid_features): base.json's MLPs behind a grid of F features per level, feat_w = its padded width
// (16 or 32).  The hot layout (fragments 0 .. N_FRAGS - 1) with the first density layer's K order of that grid: lane group g
// holds, in this order, the features of the levels {g, 4 + g, ...} it encodes --
//   F = 2: kmap(g, j) = 2 (4 (j >> 1) + g) + (j & 1)     F = 4: 4 (4 (j >> 2) + g) + (j & 3)     F = 8: 8 g + j     F = 1: 4 j + g (j < 4)
// -- and zero columns where the grid has no level (k >= feat_w, or a padded feature of feat_w itself).
void pack_fragments_grid(const std::vector<_Float16>& w16, uint32_t feat_w, uint32_t rgb_in, uint32_t F, std::vector<_Float16>& frags) {
  frags.assign((size_t)N_FRAGS * 64 * 8, (_Float16)0.0f);
  const _Float16* D0 = w16.data();                  // [64][feat_w]
  const _Float16* D1 = D0 + 64 * (size_t)feat_w;    // [16][64]
  const _Float16* R0 = D1 + 16 * 64;                // [64][rgb_in]
  const _Float16* R1 = R0 + 64 * (size_t)rgb_in;    // [64][64]
  const _Float16* R2 = R1 + 64 * 64;                // [16][64]
  auto khid = [](int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); };
  auto put = [&](int f, const _Float16* Wm, int in, int m, auto kmap) {
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int k = kmap(l >> 4, j);
        frags[((size_t)f * 64 + l) * 8 + j] = k < in ? Wm[(size_t)(16 * m + (l & 15)) * in + k] : (_Float16)0.0f;
      }
  };
  const int Fi = (int)F;
  for (int m = 0; m < 4; ++m)
    put(FRAG_D0 + m, D0, (int)feat_w, m, [Fi](int g, int j) {
      if (Fi == 1) return j < 4 ? 4 * j + g : (1 << 20);  // one feature per level: level 4 j + g; the lane's upper four are zero columns
      return Fi == 2 ? 2 * (4 * (j >> 1) + g) + (j & 1) : (Fi == 4 ? 4 * (4 * (j >> 2) + g) + (j & 3) : 8 * g + j);
    });
  for (int s = 0; s < 2; ++s) put(FRAG_D1 + s, D1, 64, 0, [&](int g, int j) { return khid(s, g, j); });
  for (int m = 0; m < 4; ++m) put(FRAG_R0 + m, R0, (int)rgb_in, m, [](int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); });
  for (int m = 0; m < 4; ++m)
    for (int s = 0; s < 2; ++s) put(FRAG_R1 + 2 * m + s, R1, 64, m, [&](int g, int j) { return khid(s, g, j); });
  for (int s = 0; s < 2; ++s) put(FRAG_R2 + s, R2, 64, 0, [&](int g, int j) { return khid(s, g, j); });
}

// The same fragment order for 16 / 32 / 128 neurons (nrf_device.h MlpShape<W>): D0 [W][32] | D1 [16][W] | R0 [W][32] | R1 [W][W] | R2 [16][W],
// MT = W / 16 row tiles, KS = ceil(W / 32) K steps; columns beyond a matrix's width are zero (W = 16: the upper half of the one step).
void pack_fragments_width(const std::vector<_Float16>& w16, int Wd, std::vector<_Float16>& frags) {
  const int MT = Wd / 16, KS = (Wd + 31) / 32;
  const int fD1 = MT, fR0 = MT + KS, fR1 = 2 * MT + KS, fR2 = 2 * MT + KS + MT * KS, n = 2 * MT + 2 * KS + MT * KS;
  frags.assign((size_t)n * 64 * 8, (_Float16)0.0f);
  const _Float16* D0 = w16.data();                 // [W][32]
  const _Float16* D1 = D0 + (size_t)Wd * 32;       // [16][W]
  const _Float16* R0 = D1 + (size_t)16 * Wd;       // [W][32]
  const _Float16* R1 = R0 + (size_t)Wd * 32;       // [W][W]
  const _Float16* R2 = R1 + (size_t)Wd * Wd;       // [16][W]
  auto khid = [](int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); };
  auto put = [&](int f, const _Float16* Wm, int in, int m, auto kmap) {
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int k = kmap(l >> 4, j);
        frags[((size_t)f * 64 + l) * 8 + j] = k < in ? Wm[(size_t)(16 * m + (l & 15)) * in + k] : (_Float16)0.0f;
      }
  };
  for (int m = 0; m < MT; ++m) put(m, D0, 32, m, [](int g, int j) { return 2 * (4 * (j >> 1) + g) + (j & 1); });
  for (int s = 0; s < KS; ++s) put(fD1 + s, D1, Wd, 0, [&](int g, int j) { return khid(s, g, j); });
  for (int m = 0; m < MT; ++m) put(fR0 + m, R0, 32, m, [](int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); });
  for (int m = 0; m < MT; ++m)
    for (int s = 0; s < KS; ++s) put(fR1 + KS * m + s, R1, Wd, m, [&](int g, int j) { return khid(s, g, j); });
  for (int s = 0; s < KS; ++s) put(fR2 + s, R2, Wd, 0, [&](int g, int j) { return khid(s, g, j); });
}

// DEPTH instance (nrf_device.h DF_*, mlp_tiles_depth): 64 neurons, nd / nr hidden layers in the density / rgb MLP.  Parameter order
// (tcnn): D0 [64][32] | (nd - 1) x [64][64] | D1 [16][64] | R0 [64][32] | (nr - 1) x [64][64] | R2 [16][64].  DEPTH_FRAGS fragments,
// unused ones zero.
void pack_fragments_depth(const std::vector<_Float16>& w16, int nd, int nr, std::vector<_Float16>& frags) {
  frags.assign((size_t)DEPTH_FRAGS * 64 * 8, (_Float16)0.0f);
  const int xd = nd - 1, xr = nr - 1;
  const _Float16* D0 = w16.data();
  const _Float16* DW = D0 + 64 * 32;
  const _Float16* D1 = DW + (size_t)xd * 64 * 64;
  const _Float16* R0 = D1 + 16 * 64;
  const _Float16* RW = R0 + 64 * 32;
  const _Float16* R2 = RW + (size_t)xr * 64 * 64;
  auto khid = [](int s, int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); };
  auto put = [&](int f, const _Float16* Wm, int in, int m, auto kmap) {
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int k = kmap(l >> 4, j);
        frags[((size_t)f * 64 + l) * 8 + j] = k < in ? Wm[(size_t)(16 * m + (l & 15)) * in + k] : (_Float16)0.0f;
      }
  };
  for (int m = 0; m < 4; ++m) put(DF_D0 + m, D0, 32, m, [](int g, int j) { return 2 * (4 * (j >> 1) + g) + (j & 1); });
  for (int s = 0; s < 2; ++s) put(DF_D1 + s, D1, 64, 0, [&](int g, int j) { return khid(s, g, j); });
  for (int m = 0; m < 4; ++m) put(DF_R0 + m, R0, 32, m, [](int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); });
  for (int s = 0; s < 2; ++s) put(DF_R2 + s, R2, 64, 0, [&](int g, int j) { return khid(s, g, j); });
  for (int e = 0; e < xd + xr; ++e) {
    const _Float16* Wm = e < xd ? DW + (size_t)e * 64 * 64 : RW + (size_t)(e - xd) * 64 * 64;
    for (int m = 0; m < 4; ++m)
      for (int s = 0; s < 2; ++s) put(DF_WW + 8 * e + 2 * m + s, Wm, 64, m, [&](int g, int j) { return khid(s, g, j); });
  }
}

// Generic instance (nrf_generic.h gen_layer): fragment (m, s) of a layer W[N][K], lane l, element j =
// W[16 m + (l & 15)][32 s + 8 (l >> 4) + j]  (natural K order), zero beyond K.
void pack_generic_layer(const _Float16* Wm, uint32_t N, uint32_t K, std::vector<_Float16>& frags) {
  const uint32_t n_tiles = N / 16, k_steps = (K + 31) / 32;
  for (uint32_t m = 0; m < n_tiles; ++m)
    for (uint32_t s = 0; s < k_steps; ++s)
      for (uint32_t l = 0; l < 64; ++l)
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t k = 32 * s + 8 * (l >> 4) + j;
          frags.push_back(k < K ? Wm[(size_t)(16 * m + (l & 15)) * K + k] : (_Float16)0.0f);
        }
}

// R/include/nerf-cuda/render_utils.h:68-77
void nerf_matrix_to_ngp(const float p[16], float s, float R[9], float org[3]) {
  const int rows[3] = {1, 2, 0};
  for (int r = 0; r < 3; ++r) {
    const float* src = p + 4 * rows[r];
    R[3 * r + 0] = src[0];
    R[3 * r + 1] = -src[1];
    R[3 * r + 2] = -src[2];
    org[r] = src[3] * s + 0.0f;
  }
}

}  // namespace

struct nrf_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool model_loaded = false;
  nrf_model_desc desc{};
  nrf_level_table lv{};
  DevModel dm{};
  void* d_grid = nullptr;
  void* d_occ = nullptr;
  void* d_wfrag = nullptr;
  void* d_lv = nullptr;
  void* d_coarse = nullptr;
  void* d_dilated = nullptr;
  void* d_ctab = nullptr;
  void* d_gen = nullptr;
  void* d_wfrag_gen = nullptr;  // wide models: generic-layout fragments for the stage entry points
  uint32_t model_hot_grid = 0;  // the loaded model's F if it has a register-resident GRID instance (else 0)
  uint32_t model_hot_width = 0; // the loaded model's width if it has a register-resident width instance (else 0)
  bool model_wide_sh = false;   // ... or the NET_WIDE_SH form (SH degree 5..8)
  void* d_wfrag_hot = nullptr;  // 16 / 32 / 128-neuron models of the base.json shape: fragments of their register-resident instance
  GenModel gen{};  // host copy of the generic instance's description (valid when dm.generic)
  std::vector<float> host_grid;  // the float density grid the march tables were built from
  bool grid_missing = false;     // loaded without a density grid and none generated yet
  bool allow_persistent = true;  // NRF_PERSISTENT=0 keeps the one-workgroup-per-strip render_kernel (A/B runs)
  bool centre_out = true;        // NRF_CENTRE_OUT=0: the persistent kernel's queue in row order
  bool allow_gen_wlds = true;    // NRF_GEN_WLDS=0: the generic instance's weight fragments are never staged in LDS
  bool allow_width_instances = true;  // NRF_WIDTH_INSTANCES=0: 16 / 32 / 128-neuron models render in the generic instance (A/B runs)
  int queue_classes = 0;         // NRF_QUEUE_CLASSES=1..8: work queues of the persistent kernel (default: one per XCD)
  int n_cus = 256;
  nrf_options opt{};
  int W = 0, H = 0;
  int n_local_tiles = 0;
  int max_views = 1;    // views the context's own frame buffers hold (nrf_set_max_views)
  int last_views = 1;
  size_t n_out_px = 0;  // pixels of ONE view in the frame buffers (= the view stride)
  size_t n_alloc_px = 0;
  void* d_rgba = nullptr;
  void* d_depth = nullptr;
  void* d_plan = nullptr;      // CALL_RING plan buffers (plan_price_kernel / plan_sort_kernel: the queue order of a launch), PLAN_BYTES each
  int plan_max_pos = 1 << 14;  // launches of up to this many strips are planned (NRF_PLAN_MAX_POS; 0: never) -- one or two 1080p views
  void* d_counters = nullptr;  // CALL_RING slots of statistics counters + work queues, one per render call (call_slot)
  int call_index = 0;          // ring position of the last render call
  int quad_levels = -1;        // NRF_QUAD_LEVELS=n: only the first n levels (whole steps of four) may get a cell-major quad copy (0: none; A/B runs)
  int quad_budget_mb = -1;     // NRF_QUAD_BUDGET_MB=m overrides nrf_model_desc.gather_copy_budget_mb (A/B runs, render_server deployments)
  uint64_t table_bytes = 0, table_ref_bytes = 0;  // device bytes of the grid table with / without the quad copies
  uint32_t gather_addresses = 0;  // lane addresses one sample sends into the texture path with the loaded model (nrf_stats)
  bool allow_gen_fast_grid = true;  // NRF_GEN_FAST_GRID=0: the generic instance always encodes with gen_level (A/B runs)
  int march_ff = 1;            // NRF_MARCH_FF=0: no barrier fast-forward ahead of t_skip (A/B runs, equality tests)
  int tail_split = 1;          // NRF_TAIL_SPLIT=0: no tail splitting in the persistent kernel (A/B runs)
  void* d_rgb8 = nullptr;
  void* d_depth8 = nullptr;
  void* bound_rgba = nullptr;  // caller-owned targets (nrf_bind_output)
  void* bound_depth = nullptr;
  void* bound_rgbd8 = nullptr;  // caller-owned packed 8-bit target (nrf_bind_output_rgbd8)
  void* bound_rgb8 = nullptr;   // caller-owned 8-bit planar targets (nrf_bind_output_u8)
  void* bound_depth8 = nullptr;
  // host frames (nrf_submit_host_u8): HOST_SLOTS x {device 8-bit planes the kernel writes, pinned host planes the copy
  // engine fills}, so that the copy of one call overlaps the render of the next
  struct HostSlot {
    void* d_buf = nullptr;     // device: rgb [views][px][3] | depth [views][px]
    uint8_t* h_buf = nullptr;  // pinned host, same layout
    size_t views = 0, px = 0;  // capacity
    hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;  // done: the call's last copy; t0 / t1: around its render launches
    hipEvent_t grp[2] = {nullptr, nullptr};                 // the two copy groups a progressive call keeps in flight (progressive_copies)
    std::vector<int> row_lo, row_hi;  // per view: the rows of the pinned planes that do not hold the background value ...
    std::vector<int> col_lo, col_hi;  // ... and, inside those rows, the columns
    int bg = -1;               // the 8-bit background value the other rows hold (-1: nothing filled yet)
    int n_views = 0, W = 0, H = 0;
    bool with_depth = true, pending = false;
    uint64_t copied = 0;
    std::vector<int> rows;     // this call: per view the rows [lo, hi) and the columns [x0, x1) of its region of interest (what is copied)
    // progress reporting (FrameParams::prog_*): the kernel flags finished strip rows, nrf_wait_host_u8 copies them meanwhile
    unsigned* d_done = nullptr;   // device [views][tiles_y]
    unsigned* h_flags = nullptr;  // pinned host [views][tiles_y]
    size_t prog_entries = 0;
    unsigned epoch = 0;
    bool progressive = false, copies_issued = false;
  } hs[2];
  int hs_next = 0;
  hipStream_t copy_stream = nullptr;
  bool host_merge = true;        // NRF_HOST_MERGE=0: every ready band is copied by itself, at once (A/B runs)
  bool host_progressive = true;  // NRF_HOST_PROGRESSIVE=0: every copy of a host frame waits for the end of its render (A/B runs)
  bool host_cols = true;         // NRF_HOST_COLS=0: a host frame that is copied after its render travels as whole rows (A/B runs)
  bool host_skip_outside = true; // NRF_HOST_SKIP_OUTSIDE=0: the kernel writes the background rows of a host frame as well (A/B runs)
  void* last_rgba = nullptr;
  void* last_depth = nullptr;
  int march_budget = 256;  // NRF_MARCH_BUDGET overrides (tuning only; the image does not depend on it)
  bool sample_cap_forced = false;  // NRF_SAMPLE_CAP was given: it holds for every launch (else launches of one or two views queue 8)
  int sample_cap = 2;      // per-round sample queue of a ray by its transmittance (FrameParams::sample_cap); NRF_SAMPLE_CAP=0 / 1: A/B runs
  bool rendered = false;
  hipStream_t last_stream = nullptr;
};

namespace {

int set_device(nrf_context* c) {
  HIP_TRY(hipSetDevice(c->device));
  return NRF_OK;
}

void free_model(nrf_context* c) {
  if (c->d_grid) (void)hipFree(c->d_grid);
  if (c->d_occ) (void)hipFree(c->d_occ);
  if (c->d_wfrag) (void)hipFree(c->d_wfrag);
  if (c->d_lv) (void)hipFree(c->d_lv);
  if (c->d_coarse) (void)hipFree(c->d_coarse);
  if (c->d_ctab) (void)hipFree(c->d_ctab);
  if (c->d_dilated) (void)hipFree(c->d_dilated);
  if (c->d_gen) (void)hipFree(c->d_gen);
  if (c->d_wfrag_gen) (void)hipFree(c->d_wfrag_gen);
  if (c->d_wfrag_hot) (void)hipFree(c->d_wfrag_hot);
  c->d_wfrag_gen = c->d_wfrag_hot = nullptr;
  c->d_grid = c->d_occ = c->d_wfrag = c->d_lv = c->d_coarse = c->d_ctab = c->d_dilated = c->d_gen = nullptr;
  c->model_loaded = false;
}

constexpr int CALL_RING = 16;  // render calls of one context that may be in flight on different streams
constexpr size_t CALL_SLOT_BYTES = COUNTER_BYTES + 128 + 65536;  // statistics | work queues | the diagnostic build's per-wave stamps
constexpr int HOST_SLOTS = 2;
// Cell-major quad copies of grid levels (nrf_load_model; nrf_device.h level_gather_quad): which levels get one by default.
constexpr uint32_t QUAD_BUDGET_MB_DEFAULT = 8192;  // base.json's grid: levels 0..7 take 95 MB, levels 8..11 4.5 GB (levels 12..15: 216 GB, never copied);
                                                   // an instant-ngp grid at aabb_scale 32: levels 0..3 2.4 MB, levels 4..7 9.2 GB
inline char* call_slot(const nrf_context* c, int index) { return (char*)c->d_counters + (size_t)(index % CALL_RING) * CALL_SLOT_BYTES; }

void free_host_slots(nrf_context* c) {
  for (auto& h : c->hs) {
    if (h.d_buf) (void)hipFree(h.d_buf);
    if (h.h_buf) (void)hipHostFree(h.h_buf);
    if (h.d_done) (void)hipFree(h.d_done);
    if (h.h_flags) (void)hipHostFree(h.h_flags);
    h.d_buf = nullptr;
    h.h_buf = nullptr;
    h.d_done = h.h_flags = nullptr;
    h.views = h.px = h.prog_entries = 0;
    h.bg = -1;
    h.pending = false;
  }
}

void free_frame(nrf_context* c) {
  if (c->d_rgba) (void)hipFree(c->d_rgba);
  if (c->d_depth) (void)hipFree(c->d_depth);
  if (c->d_rgb8) (void)hipFree(c->d_rgb8);
  if (c->d_depth8) (void)hipFree(c->d_depth8);
  c->d_rgba = c->d_depth = c->d_rgb8 = c->d_depth8 = nullptr;
  c->n_out_px = 0;
  c->rendered = false;
}

int total_strips(int W, int H) { return ((((W + 7) / 8) + 3) / 4) * ((H + 7) / 8); }

int local_tiles(int W, int H, int shard_index, int shard_count) {
  const int total = total_strips(W, H);
  if (shard_index >= total) return 0;
  return 4 * ((total - shard_index + shard_count - 1) / shard_count);
}

// the shard layout (tile-major [n_tiles][64]) instead of the row-major frame: every multi-shard render, and a single shard on request
bool tiled_layout(const nrf_context* c) { return c->opt.shard_count > 1 || c->opt.tile_major != 0; }

int alloc_frame(nrf_context* c) {
  if (c->W <= 0 || c->H <= 0) return NRF_OK;
  const bool tiled = tiled_layout(c);
  c->n_local_tiles = local_tiles(c->W, c->H, c->opt.shard_index, c->opt.shard_count);
  int tps = 0;
  nrf_tiles_per_shard(c->W, c->H, c->opt.shard_count, &tps);
  const size_t per_view = tiled ? (size_t)tps * 64 : (size_t)c->W * c->H;
  const size_t need = per_view * (size_t)c->max_views;
  if (per_view == c->n_out_px && need == c->n_alloc_px && c->d_rgba) return NRF_OK;
  HIP_TRY(hipDeviceSynchronize());
  free_frame(c);
  HIP_TRY(hipMalloc(&c->d_rgba, need * 16));
  HIP_TRY(hipMalloc(&c->d_depth, need * 4));
  HIP_TRY(hipMemsetAsync(c->d_rgba, 0, need * 16, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_depth, 0, need * 4, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_out_px = per_view;
  c->n_alloc_px = need;
  return NRF_OK;
}

int fill_frame_params(nrf_context* c, const float cam[4], const float pose[16], FrameParams& P) {
  std::memset(&P, 0, sizeof(P));
  nerf_matrix_to_ngp(pose, c->desc.scale, P.R, P.org);
  for (int i = 0; i < 4; ++i) P.cam[i] = cam[i];
  P.W = c->W;
  P.H = c->H;
  P.tiles_x = (c->W + 7) / 8;
  P.tiles_y = (c->H + 7) / 8;
  P.shard_index = c->opt.shard_index;
  P.shard_count = c->opt.shard_count;
  P.n_local_tiles = c->n_local_tiles;
  P.tile_major = tiled_layout(c);
  P.bg_color = c->opt.bg_color;
  P.min_near = c->opt.min_near;
  P.dt_gamma = c->opt.dt_gamma;
  P.density_scale = c->opt.density_scale;
  P.max_steps = c->opt.max_steps;
  P.march_budget = c->march_budget;
  P.sample_cap = c->sample_cap;
  P.centre_out = c->centre_out ? 1 : 0;
  P.out_mode = c->bound_rgbd8 ? OUT_RGBD8 : (c->bound_rgb8 ? OUT_U8 : OUT_F32);
  P.skip_outside = 0;
  P.prog_done = P.prog_flags = nullptr;
  P.prog_epoch = 0;
  P.tail_split = c->tail_split;
  P.perturb = c->opt.perturb;
  P.march_ff = c->opt.perturb ? 0 : c->march_ff;  // (the fast-forward replays a chain that starts at `near`: a shifted chain keeps its trips)
  P.fast_interp = c->opt.fast_interp ? 1 : 0;
  P.queue_classes = c->queue_classes;
  return NRF_OK;
}

// Pixel rectangle outside of which no ray of the view can enter `box` (the inflated box of occupied cells,
// NGP coordinates): the bounding rectangle of the projections of its 8 corners, 3 pixels wider on every side;
// the whole image when a corner is not safely in front of the camera; empty when the box is.  Conservative
// by construction: the box is convex, so a ray that enters it passes through the convex hull of the projected
// corners; rays inside the rectangle still take the exact per-ray slab test in the kernel.
void view_roi(const float R[9], const float org[3], const float cam[4], const float box[6], int W, int H, int roi[4]) {
  roi[0] = 0; roi[1] = 0; roi[2] = W - 1; roi[3] = H - 1;
  if (!(box[0] <= box[3])) {  // no occupied cell at all
    roi[2] = -1;
    roi[3] = -1;
    return;
  }
  // camera coordinates of a world offset p: v = R^-1 p (ray_dir applies R to the camera-space direction; poses
  // need not be orthonormal, so the inverse is computed, not assumed to be the transpose)
  const double a = R[0], b = R[1], cc = R[2], d = R[3], e = R[4], f = R[5], g = R[6], h = R[7], i = R[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + cc * (d * h - e * g);
  if (!(std::fabs(det) > 1e-12)) return;
  const double inv[9] = {(e * i - f * h) / det, (cc * h - b * i) / det, (b * f - cc * e) / det,
                         (f * g - d * i) / det, (a * i - cc * g) / det, (cc * d - a * f) / det,
                         (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int c = 0; c < 8; ++c) {
    const double p[3] = {(double)box[(c & 1) ? 3 : 0] - org[0], (double)box[(c & 2) ? 4 : 1] - org[1],
                         (double)box[(c & 4) ? 5 : 2] - org[2]};
    const double vx = inv[0] * p[0] + inv[1] * p[1] + inv[2] * p[2];
    const double vy = inv[3] * p[0] + inv[4] * p[1] + inv[5] * p[2];
    const double vz = inv[6] * p[0] + inv[7] * p[1] + inv[8] * p[2];
    if (!(vz > 1e-3)) return;  // corner beside / behind the camera (or NaN): keep the whole image
    const double u = cam[2] + cam[0] * vx / vz, v = cam[3] + cam[1] * vy / vz;
    if (!(u == u) || !(v == v)) return;
    lo[0] = u < lo[0] ? u : lo[0]; hi[0] = u > hi[0] ? u : hi[0];
    lo[1] = v < lo[1] ? v : lo[1]; hi[1] = v > hi[1] ? v : hi[1];
  }
  // pixel (i, j) looks through (i + 0.5, j + 0.5)
  const double m = 3.0;
  const double x0 = std::floor(lo[0] - 0.5 - m), y0 = std::floor(lo[1] - 0.5 - m);
  const double x1 = std::ceil(hi[0] - 0.5 + m), y1 = std::ceil(hi[1] - 0.5 + m);
  roi[0] = x0 < 0 ? 0 : (x0 > W ? W : (int)x0);
  roi[1] = y0 < 0 ? 0 : (y0 > H ? H : (int)y0);
  roi[2] = x1 < -1 ? -1 : (x1 > W - 1 ? W - 1 : (int)x1);
  roi[3] = y1 < -1 ? -1 : (y1 > H - 1 ? H - 1 : (int)y1);
}

// Everything the march needs from the density grid (reference: the float grid [C*H^3] of load_snapshot,
// nerf_render.cu:441-466, read by kernel_march_rays): occupancy bits, coarse occupancy, the box of occupied cells,
// the dilated coarse sets of the visibility walk and the cell-boundary table.  Called by nrf_load_model with the
// snapshot's grid and by nrf_generate_density_grid with the one evaluated from the network.  Needs c->desc, and
// c->dm.generic / gen_wave_bytes (LDS budget of the tables).
int set_density_grid(nrf_context* c, const float* density_grid, float mean_density) {
  const nrf_model_desc* d = &c->desc;
  const uint64_t Hh = d->density_grid_size;
  const uint64_t cells = Hh * Hh * Hh * d->cascade;
  HIP_TRY(hipDeviceSynchronize());  // nothing may still be marching on the old tables
  for (void** q : {&c->d_occ, &c->d_coarse, &c->d_ctab, &c->d_dilated}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  // occupancy bitfield: grid[cell] > min(0.01, mean_density) (render_utils.h:560,619), decided once
  const float thresh = fminf(0.01f, mean_density);
  std::vector<uint32_t> occ((cells + 31) / 32 + 1, 0u);
  for (uint64_t i = 0; i < cells; ++i)
    if (density_grid[i] > thresh) occ[i >> 5] |= 1u << (i & 31);

  // march tables (nrf_device.h march_next): coarse occupancy = OR over 4x4x4 cell blocks, and the
  // cell-boundary table ((v/(H-1))*2-1)*mip_bound in the reference's fp32 operation order
  const uint32_t Hs = d->density_grid_size, Cs = d->cascade;
  const uint32_t coarse_shift = (Hs % 4 == 0 && Hs >= 8) ? 2u : 0u;
  std::vector<uint32_t> coarse;
  if (coarse_shift) {
    const uint32_t Hc = Hs >> 2;
    coarse.assign(((uint64_t)Cs * Hc * Hc * Hc + 31) / 32 + 1, 0u);
    for (uint64_t i = 0; i < cells; ++i) {
      if (!((occ[i >> 5] >> (i & 31)) & 1u)) continue;
      const uint32_t level = (uint32_t)(i / (Hh * Hh * Hh));
      const uint64_t r = i % (Hh * Hh * Hh);
      const uint32_t nx = (uint32_t)(r / (Hh * Hh)), ny = (uint32_t)((r / Hh) % Hh), nz = (uint32_t)(r % Hh);
      const uint64_t cc = (((uint64_t)level * Hc + (nx >> 2)) * Hc + (ny >> 2)) * Hc + (nz >> 2);
      coarse[cc >> 5] |= 1u << (cc & 31);
    }
  }
  // A density cell of cascade k >= 1 is only ever looked up for positions of level k, i.e. with
  // max|p| >= 2^(k-1) (kernel_march_rays picks the level from frexp(max|p|), render_utils.h:603-607): cells
  // that lie, with one cell of slack, wholly inside the inner cube max|p| < 2^(k-1) cannot produce a sample
  // whatever their value, so they count neither for the box of occupied cells nor for the visibility sets.
  auto reachable = [&](uint32_t level, uint32_t nx, uint32_t ny, uint32_t nz) {
    if (level == 0 || Cs <= 1) return true;
    const double mb = fmin(ldexp(1.0, (int)level), (double)d->bound), cell = 2.0 * mb / (double)Hs;
    const uint32_t n3[3] = {nx, ny, nz};
    double r_max = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double lo = -mb + n3[a] * cell, hi = lo + cell;
      r_max = fmax(r_max, fmax(fabs(lo), fabs(hi)));
    }
    return !(r_max + cell < ldexp(1.0, (int)level - 1));
  };
  // world-space box around every occupied (and reachable) cell, inflated by 2 cells of its cascade level
  float occ_box[6] = {1.f, 1.f, 1.f, -1.f, -1.f, -1.f};  // empty
  bool boundary_occupied = false;  // an occupied (reachable) cell with index 0 or H-1 on some axis
  {
    bool any = false;
    for (uint32_t level = 0; level < Cs; ++level) {
      uint32_t lo[3] = {Hs, Hs, Hs}, hi[3] = {0, 0, 0};
      bool lvl_any = false;
      for (uint64_t r = 0; r < Hh * Hh * Hh; ++r) {
        const uint64_t i = (uint64_t)level * Hh * Hh * Hh + r;
        if (!((occ[i >> 5] >> (i & 31)) & 1u)) continue;
        const uint32_t n3[3] = {(uint32_t)(r / (Hh * Hh)), (uint32_t)((r / Hh) % Hh), (uint32_t)(r % Hh)};
        if (!reachable(level, n3[0], n3[1], n3[2])) continue;
        for (int a = 0; a < 3; ++a) { lo[a] = n3[a] < lo[a] ? n3[a] : lo[a]; hi[a] = n3[a] > hi[a] ? n3[a] : hi[a]; }
        lvl_any = true;
      }
      if (!lvl_any) continue;
      const double mip_bound = fmin(Cs > 1 ? ldexp(1.0, (int)level) : 1.0, (double)d->bound);
      const double cell = 2.0 * mip_bound / (double)Hs;
      for (int a = 0; a < 3; ++a) {
        float wlo = (float)(-mip_bound + ((double)lo[a] - 2.0) * cell);
        float whi = (float)(-mip_bound + ((double)hi[a] + 3.0) * cell);
        // The march clamps the position to +-bound and then the cell index to [0, H-1] (render_utils.h:595-611):
        // a position OUTSIDE this cascade's cube -- bound > 2^(C-1), or an aabb wider than +-bound -- lands in the
        // boundary layer of cells.  An occupied boundary cell therefore stands for everything beyond that face: the
        // box is extended to wherever a ray can be (its aabb range) on that side.
        if (lo[a] == 0) wlo = fminf(wlo, fminf(d->aabb[a], -d->bound) - (float)(2.0 * cell));
        if (hi[a] == Hs - 1) whi = fmaxf(whi, fmaxf(d->aabb[a + 3], d->bound) + (float)(2.0 * cell));
        if (!any || wlo < occ_box[a]) occ_box[a] = wlo;
        if (!any || whi > occ_box[a + 3]) occ_box[a + 3] = whi;
        boundary_occupied = boundary_occupied || lo[a] == 0 || hi[a] == Hs - 1;
      }
      any = true;
    }
  }
  // Positions outside the outermost cube exist when bound > its mip_bound or the aabb is wider than +-bound.  The
  // per-cascade visibility walk (coarse_visibility) only covers a ray's stretch INSIDE each cube, so with an
  // occupied boundary layer in such a model it would miss those samples: the walk is switched off then (the box
  // test above stays exact).
  bool exterior_positions = false;
  {
    const float outer = fminf(Cs > 1 ? ldexpf(1.0f, (int)Cs - 1) : 1.0f, d->bound);
    exterior_positions = d->bound > outer;
    for (int a = 0; a < 3; ++a) exterior_positions = exterior_positions || d->aabb[a] < -d->bound || d->aabb[a + 3] > d->bound;
  }
  const bool visibility_walk = !(exterior_positions && boundary_occupied);
  // Conservative coarse visibility set (single cascade): coarse cells that contain, or lie within one
  // density cell of, an occupied density cell (= the coarse image of the occupancy dilated by one
  // fine cell); used by the per-ray DDA of render_kernel (nrf_device.h coarse_visibility).
  std::vector<uint32_t> dilated;
  uint32_t dilated_level_words = 0;  // words per cascade level (whole words, so a level's bits start at bit 0)
  if (coarse_shift && visibility_walk) {
    const int Hc = (int)(Hs >> 2), Hf = (int)Hs;
    dilated_level_words = (uint32_t)(((uint64_t)Hc * Hc * Hc + 31) / 32);
    dilated.assign((size_t)dilated_level_words * Cs, 0u);
    for (uint32_t level = 0; level < Cs; ++level) {
      uint32_t* dl = dilated.data() + (size_t)level * dilated_level_words;
      for (int x = 0; x < Hf; ++x)
        for (int y = 0; y < Hf; ++y)
          for (int z = 0; z < Hf; ++z) {
            const uint64_t i = (((uint64_t)level * Hf + x) * Hf + y) * Hf + z;
            if (!((occ[i >> 5] >> (i & 31)) & 1u)) continue;
            if (!reachable(level, (uint32_t)x, (uint32_t)y, (uint32_t)z)) continue;
            for (int dx = -1; dx <= 1; ++dx)
              for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                  const int X = x + dx, Y = y + dy, Z = z + dz;
                  if (X < 0 || Y < 0 || Z < 0 || X >= Hf || Y >= Hf || Z >= Hf) continue;
                  const uint64_t nn = ((uint64_t)(X >> 2) * Hc + (Y >> 2)) * Hc + (Z >> 2);
                  dl[nn >> 5] |= 1u << (nn & 31);
                }
          }
    }
  }
  std::vector<float> ctab((size_t)Cs * (Hs + 1));
  for (uint32_t level = 0; level < Cs; ++level) {
    const float mip_bound = fminf(Cs > 1 ? ldexpf(1.0f, (int)level) : 1.0f, d->bound);
    const float Hm1 = (float)(Hs - 1);
    for (uint32_t v = 0; v <= Hs; ++v) ctab[(size_t)level * (Hs + 1) + v] = ((float)v / Hm1 * 2 - 1) * mip_bound;
  }

  auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  HIP_TRY(upload(&c->d_occ, occ.data(), occ.size() * 4));
  if (coarse_shift) HIP_TRY(upload(&c->d_coarse, coarse.data(), coarse.size() * 4));
  HIP_TRY(upload(&c->d_ctab, ctab.data(), ctab.size() * 4));
  if (!dilated.empty()) HIP_TRY(upload(&c->d_dilated, dilated.data(), dilated.size() * 4));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipDeviceSynchronize());
  DevModel& M = c->dm;
  M.occ_bits = (const uint32_t*)c->d_occ;
  for (int i = 0; i < 6; ++i) M.occ_box[i] = occ_box[i];
  M.occ_coarse = (const uint32_t*)c->d_coarse;
  M.cell_bound = (const float*)c->d_ctab;
  M.occ_dilated = (const uint32_t*)c->d_dilated;
  M.coarse_shift = coarse_shift;
  M.lds_coarse_words = M.lds_ctab_floats = M.lds_dilated_words = 0;
  {
    const uint64_t words = coarse_shift ? (uint64_t)coarse.size() : 0, fl = ctab.size();
    uint64_t budget = (uint64_t)render_lds_table_max_bytes();
    if (M.generic) {  // whatever the generic instance's rows leave of the CU's 160 KiB
      const uint64_t used = (uint64_t)render_gen_lds_fixed_bytes(M.gen_wave_bytes);
      budget = used + budget <= 160u * 1024u ? budget : 160u * 1024u - used;
    }
    if (M.wide) {  // three workgroups per CU: (160 KiB / 3 - fixed part) for the tables
      const uint64_t room = 160u * 1024u / 3u - (uint64_t)render_wide_lds_fixed_bytes();
      budget = budget < room ? budget : room;
    }
    if (coarse_shift && 4 * (words + fl) <= budget) {
      M.lds_coarse_words = (uint32_t)words;
      M.lds_ctab_floats = (uint32_t)fl;
    }
  }
  M.dilated_level_words = dilated_level_words;
  if (!dilated.empty() && dilated.size() * 4 <= (size_t)N_FRAGS * 64 * 16) M.lds_dilated_words = (uint32_t)dilated.size();
  // The persistent form of the render kernel (one workgroup per CU, waves pull strips from work queues) keeps every march
  // table in LDS for the whole launch: tables that fit beside the blocks of its waves (16 hot, 12 wide, 12 or 8 generic).
  M.persistent = 0;
  M.persist_waves = 0;
  M.n_cus = (uint32_t)c->n_cus;
  M.hot_width = c->model_hot_width;  // (decided again for every grid: nrf_generate_density_grid calls this too)
  M.wide_sh = c->model_wide_sh ? 1u : 0u;
  M.hot_grid = c->model_hot_grid;
  bool width_instance = false;  // the model's frames come from the register-resident instance of its width (persistent kernel only)
  if (c->allow_persistent && M.lds_coarse_words > 0 && (M.hot_width || M.wide_sh || M.hot_grid)) {
    const size_t tables = 4 * ((size_t)M.lds_coarse_words + M.lds_ctab_floats + dilated.size());
    const size_t fixed = M.wide_sh ? (size_t)render_persistent_lds_widesh_bytes()
                         : M.hot_grid ? (size_t)render_persistent_lds_fixed_bytes(0u, 0u, 0u, 16) : (size_t)render_persistent_lds_width_bytes((int)M.hot_width);
    if (fixed + tables <= 160u * 1024u) {
      M.persistent = 1;
      M.persist_waves = (uint32_t)render_persist_waves_for(M.generic, M.wide, M.wide_sh, M.hot_width, M.hot_grid, march_form(M.H, M.cascade, M.bound));
      M.gen_weights_lds = 0;
      M.lds_dilated_words = (uint32_t)dilated.size();
      width_instance = true;
    }
  }
  if (!width_instance) M.hot_width = M.wide_sh = M.hot_grid = 0;  // (the tables do not fit beside its workgroup: the generic instance renders)
  if (c->allow_persistent && M.lds_coarse_words > 0 && !width_instance) {
    const size_t tables = 4 * ((size_t)M.lds_coarse_words + M.lds_ctab_floats + dilated.size());
    // generic instance: 12 waves with the weight fragments in LDS, 12 waves without, 8 with, 8 without -- the first that fits
    // (NRF_GEN_WLDS=0 at nrf_create: never stage the fragments)
    const bool allow_wlds = c->allow_gen_wlds;
    M.gen_weights_lds = 0;
    // (the wide instance with the generic march -- a grid size or bound that is no power of two -- is compiled for 8 waves:
    //  launch_render's choice of the march form, nrf_kernels_wide.hip)
    const int first_waves = render_persist_waves_for(M.generic, M.wide, 0u, 0u, 0u, march_form(M.H, M.cascade, M.bound));
    for (int waves : {first_waves, M.generic && first_waves != 8 ? 8 : 0}) {  // (only the generic instance has a second workgroup size)
      if (waves == 0) break;
      const size_t fixed = (size_t)render_persistent_lds_fixed_bytes(M.generic, M.wide, M.gen_wave_bytes, waves) + tables;
      for (int wlds : {1, 0}) {
        if (wlds && (!M.generic || !allow_wlds)) continue;
        if (fixed + (wlds ? 16u + M.gen_frag_bytes : 0u) <= 160u * 1024u) {
          M.persistent = 1;
          M.persist_waves = (uint32_t)waves;
          M.gen_weights_lds = (uint32_t)wlds;
          M.lds_dilated_words = (uint32_t)dilated.size();
          break;
        }
      }
      if (M.persistent) break;
    }
  }
  c->desc.mean_density = mean_density;
  c->host_grid.assign(density_grid, density_grid + cells);
  return NRF_OK;
}

int need_model(nrf_context* c) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if (!c->model_loaded) return fail(NRF_E_STATE, "no model loaded (call nrf_load_model first)");
  return set_device(c);
}

}  // namespace

extern "C" {

const char* nrf_last_error(void) { return g_err.c_str(); }
void nrf_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }  // used by nrf_renderbuffer.hip
int nrf_abi_version(void) { return NRF_ABI_VERSION; }

void nrf_default_options(nrf_options* o) {
  if (!o) return;
  o->bg_color = 1.0f;
  o->min_near = 0.2f;
  o->dt_gamma = 1.0f / 128.0f;
  o->max_steps = 1024;
  o->density_scale = 1.0f;
  o->perturb = 0;
  o->shard_index = 0;
  o->shard_count = 1;
  o->fast_interp = 0;
  o->tile_major = 0;
}

int nrf_level_table_compute(const nrf_model_desc* d, nrf_level_table* t) {
  if (!d || !t) return fail(NRF_E_INVALID, "null argument");
  return compute_level_table(*d, *t);
}

int nrf_expected_n_params(const nrf_model_desc* d, uint64_t* n) {
  if (!d || !n) return fail(NRF_E_INVALID, "null argument");
  nrf_level_table t;
  int rc = compute_level_table(*d, t);
  if (rc) return rc;
  return expected_params(*d, t, *n);
}

int nrf_default_per_level_scale(float bound, uint32_t base_resolution, uint32_t n_levels, float* out) {
  if (!out || n_levels < 2 || base_resolution == 0) return fail(NRF_E_INVALID, "bad argument");
  const float desired_resolution = 2048.0f;  // R/src/nerf_render.cu:154-165
  *out = std::exp(std::log(desired_resolution * bound / (float)base_resolution) / (float)(n_levels - 1));
  return NRF_OK;
}

int nrf_tiles_per_shard(int width, int height, int shard_count, int* n) {
  if (!n || width <= 0 || height <= 0 || shard_count <= 0) return fail(NRF_E_INVALID, "bad argument");
  *n = 4 * ((total_strips(width, height) + shard_count - 1) / shard_count);
  return NRF_OK;
}

int nrf_create(int device, nrf_context** out) {
  if (!out) return fail(NRF_E_INVALID, "null argument");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(NRF_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= count) return fail(NRF_E_NODEVICE, "device index out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NRF_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  nrf_context* c = new nrf_context;
  c->device = device;
  nrf_default_options(&c->opt);
  if (const char* e = std::getenv("NRF_MARCH_BUDGET")) {
    const int b = std::atoi(e);
    if (b >= 1 && b <= 4096) c->march_budget = b;
  }
  if (const char* e = std::getenv("NRF_SAMPLE_CAP")) { c->sample_cap = std::atoi(e); c->sample_cap_forced = true; }
  if (const char* e = std::getenv("NRF_PERSISTENT")) c->allow_persistent = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_CENTRE_OUT")) c->centre_out = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_GEN_WLDS")) c->allow_gen_wlds = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_WIDTH_INSTANCES")) c->allow_width_instances = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_QUEUE_CLASSES")) c->queue_classes = std::atoi(e);
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  if (const char* e = std::getenv("NRF_HOST_PROGRESSIVE")) c->host_progressive = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_HOST_MERGE")) c->host_merge = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_HOST_SKIP_OUTSIDE")) c->host_skip_outside = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_HOST_COLS")) c->host_cols = std::atoi(e) != 0;
  // The copy stream gets a hardware queue of its own.  HIP maps streams onto a few hardware queues (4 by default) and a
  // device-to-host copy issued while a render is resident on a queue it shares does not start before that render has
  // ended (scripts/copy_overlap_probe.py: a 133 MB copy issued 2 ms into a 14 ms render ended 2.3 ms after the render's
  // END; on a queue of its own it ran beside the render and ended 4.5 ms after its start).  Streams of another priority
  // come from another pool of hardware queues, whatever GPU_MAX_HW_QUEUES says.
  {
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_hi));
  }
  for (auto& h : c->hs) {
    HIP_TRY(hipEventCreateWithFlags(&h.done, hipEventDisableTiming));
    for (hipEvent_t& g : h.grp) HIP_TRY(hipEventCreateWithFlags(&g, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&h.t0));
    HIP_TRY(hipEventCreate(&h.t1));
  }
  if (const char* e = std::getenv("NRF_TAIL_SPLIT")) c->tail_split = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("NRF_MARCH_FF")) c->march_ff = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("NRF_GEN_FAST_GRID")) c->allow_gen_fast_grid = std::atoi(e) != 0;
  if (const char* e = std::getenv("NRF_QUAD_LEVELS")) c->quad_levels = std::atoi(e);
  if (const char* e = std::getenv("NRF_QUAD_BUDGET_MB")) c->quad_budget_mb = std::max(0, std::atoi(e));
  HIP_TRY(hipMalloc(&c->d_counters, CALL_RING * CALL_SLOT_BYTES));
  HIP_TRY(hipMemset(c->d_counters, 0, CALL_RING * CALL_SLOT_BYTES));
  if (const char* e = std::getenv("NRF_PLAN_MAX_POS")) c->plan_max_pos = std::max(0, std::min(std::atoi(e), (int)PLAN_CAP));
  HIP_TRY(hipMalloc(&c->d_plan, CALL_RING * PLAN_BYTES));
  HIP_TRY(hipMemset(c->d_plan, 0, CALL_RING * PLAN_BYTES));
  *out = c;
  return NRF_OK;
}

int nrf_destroy(nrf_context* c) {
  if (!c) return NRF_OK;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  free_model(c);
  free_frame(c);
  free_host_slots(c);
  for (auto& h : c->hs) {
    if (h.done) (void)hipEventDestroy(h.done);
    for (hipEvent_t g : h.grp) if (g) (void)hipEventDestroy(g);
    if (h.t0) (void)hipEventDestroy(h.t0);
    if (h.t1) (void)hipEventDestroy(h.t1);
  }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_plan) (void)hipFree(c->d_plan);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return NRF_OK;
}

int nrf_load_model(nrf_context* c, const nrf_model_desc* d) {
  if (!c || !d || !d->params) return fail(NRF_E_INVALID, "null argument");
  if (d->abi_version != NRF_ABI_VERSION) return fail(NRF_E_INVALID, "abi_version mismatch");
  int rc = set_device(c);
  if (rc) return rc;
  // what the reference's vocabulary allows (T/.../grid.h:1403-1411, T/src/fully_fused_mlp.cu:700-725, 653-655;
  // spherical_harmonics.h:394-412); anything outside is refused loudly, never emulated on the CPU
  const uint32_t F = d->n_features_per_level;
  if (F != 1 && F != 2 && F != 4 && F != 8) return fail(NRF_E_INVALID, "GridEncoding: n_features_per_level must be 1, 2, 4, or 8.");
  if (d->interpolation > NRF_INTERP_SMOOTHSTEP) return fail(NRF_E_INVALID, "Invalid interpolation type");
  if (d->n_neurons != 16 && d->n_neurons != 32 && d->n_neurons != 64 && d->n_neurons != 128)
    return fail(NRF_E_INVALID, "FullyFusedMLP: n_neurons must be 16, 32, 64 or 128");
  if (d->density_hidden_layers < 1 || d->rgb_hidden_layers < 1)
    return fail(NRF_E_INVALID, "FullyFusedMLP requires at least 1 hidden layer (3 layers in total).");
  if (d->density_hidden_layers + d->rgb_hidden_layers + 2 > (uint32_t)GEN_MAX_LAYERS)
    return fail(NRF_E_UNSUPPORTED, "HIP path: more than 24 layers in the two MLPs together");
  if (d->density_n_output < 1 || d->density_n_output > 16)  // wider outputs take tcnn's CUTLASS last layer (out of scope)
    return fail(NRF_E_UNSUPPORTED, "HIP path: density n_output_dims must be 1..16");
  const uint32_t raw = dir_raw_width(*d);
  if (d->dir_encoding == NRF_DIR_SH && (d->sh_degree < 1 || d->sh_degree > 8))
    return fail(NRF_E_INVALID, "SphericalHarmonics: degree must be 1..8");
  if (raw == 0 || next_multiple(raw, 16u) > (uint32_t)GEN_MAX_DIR_W)
    return fail(NRF_E_UNSUPPORTED, "HIP path: direction encoding must have 1..112 outputs (after padding to 16)");
  if (d->density_grid_size < 2 || d->density_grid_size >= (1u << 24) || d->cascade < 1)
    return fail(NRF_E_INVALID, "bad density grid geometry");
  if (!(d->bound > 0.0f)) return fail(NRF_E_INVALID, "bound must be positive");

  nrf_level_table lv;
  rc = compute_level_table(*d, lv);
  if (rc) return rc;
  uint64_t expect = 0;
  rc = expected_params(*d, lv, expect);
  if (rc) return rc;
  if (d->n_params != expect)  // R/include/nerf-cuda/nerf_network.h:425-427
    return fail(NRF_E_PARAMS, "Can't set params because number of parameters and model size do not match with each other.");
  const uint64_t Hh = d->density_grid_size;
  const uint64_t cells = Hh * Hh * Hh * d->cascade;
  if (d->density_grid && d->n_density_grid != cells)  // R/src/nerf_render.cu:467-469
    return fail(NRF_E_PARAMS, "Incompatible number of grid cascades.");
  if (cells >= (1ull << 32)) return fail(NRF_E_UNSUPPORTED, "density grid too large");

  HIP_TRY(hipDeviceSynchronize());  // nothing may still be reading the old model
  free_model(c);
  // fp32 -> fp16 cast of every parameter (nerf_network.h:434-436), order: density MLP | rgb MLP | grid;
  // each MLP: first [W x in] | hidden [W x W] ... | last [16 x W] (fully_fused_mlp.cu:636-687)
  const uint32_t L = d->n_levels, Wn = d->n_neurons;
  const uint32_t feat_raw = L * F, feat_w = next_multiple(feat_raw, 16u);
  const uint32_t dir_w = next_multiple(raw, 16u), rgb_in = 16u + dir_w;
  struct LayerDim { uint32_t N, K, act; };
  std::vector<LayerDim> layers;
  auto add_mlp = [&](uint32_t in, uint32_t hidden, uint32_t act, uint32_t out_act) {
    layers.push_back({Wn, in, act});
    for (uint32_t i = 1; i < hidden; ++i) layers.push_back({Wn, Wn, act});
    layers.push_back({16u, Wn, out_act});
  };
  add_mlp(feat_w, d->density_hidden_layers, d->density_activation, d->density_output_activation);
  add_mlp(rgb_in, d->rgb_hidden_layers, d->rgb_activation, d->rgb_output_activation);
  size_t n_mlp = 0;
  for (const LayerDim& ly : layers) n_mlp += (size_t)ly.N * ly.K;
  std::vector<_Float16> w16(n_mlp);
  for (size_t i = 0; i < n_mlp; ++i) w16[i] = (_Float16)d->params[i];
  const size_t n_grid = (size_t)lv.offset[L] * F;
  const float* gp = d->params + n_mlp;
  std::vector<LevelParams> lp(16);
  bool generic_grid = false;
  for (uint32_t l = 0; l < 16; ++l) std::memset(&lp[l], 0, sizeof(LevelParams));
  for (uint32_t l = 0; l < L; ++l) {
    LevelParams& Lv = lp[l];
    Lv.scale = lv.scale[l];
    Lv.res = lv.resolution[l];
    Lv.size = lv.offset[l + 1] - lv.offset[l];
    Lv.hashed = d->grid_type == NRF_GRID_HASH;
    // replay grid_index's stride loop (grid.h:106-114) in uint32 to classify the level
    uint32_t stride = 1, mult[3] = {0, 0, 0};  // mult: what grid_index multiplies x, y, z with (0: the term is skipped)
    int dims = 0;
    for (; dims < 3 && stride <= Lv.size; ++dims) {
      mult[dims] = stride;
      stride *= Lv.res;  // uint32, as in the reference: wraps for res^3 >= 2^32
    }
    const bool uses_hash = Lv.hashed && Lv.size < stride;
    const bool pow2_size = Lv.size >= 2 && (Lv.size & (Lv.size - 1)) == 0;
    if (uses_hash && pow2_size) Lv.mode = LV_HASH_POW2;
    else if (!uses_hash && dims == 3 && Lv.res >= 2 && (uint64_t)Lv.res * Lv.res * Lv.res <= Lv.size) Lv.mode = LV_DENSE;
    else if (!uses_hash && pow2_size && dims >= 1) Lv.mode = LV_ADD_POW2;  // (x + y * mult[1] + z * mult[2]) & (size - 1), see nrf_device.h
    else Lv.mode = LV_GENERIC;
    Lv.my_b = mult[1] << 2;  // (the hashed levels' constants replace these below)
    Lv.mz_b = mult[2] << 2;
    generic_grid = generic_grid || Lv.mode == LV_GENERIC;
  }
  // The register-resident instance is the shape of the reference's base.json; everything else is the generic one.
  // (a Frequency direction encoding of 32..80 values keeps the register-resident instance: its `wide` form)
  const bool wide = dir_w > 16 && dir_w <= 16u * (2u * RK_WIDE - 1u) && d->dir_encoding == NRF_DIR_FREQUENCY;
  const bool generic = generic_grid || F != 2 || L != 16 || Wn != 64 || d->density_hidden_layers != 1 || d->rgb_hidden_layers != 2 ||
                       !(dir_w == 16 || wide) || d->interpolation != NRF_INTERP_LINEAR ||
                       !(d->density_activation == NRF_ACT_RELU && d->rgb_activation == NRF_ACT_RELU &&
                         d->density_output_activation == NRF_ACT_NONE &&
                         (d->rgb_output_activation == NRF_ACT_NONE || d->rgb_output_activation == NRF_ACT_SIGMOID) &&
                         d->sigma_activation == NRF_ACT_EXPONENTIAL);
  // ... except for its width: 16 / 32 / 128 neurons (tcnn's other FullyFusedMLP widths) keep the register-resident form in
  // the persistent kernel (NET_W16 / NET_W32 / NET_W128); everywhere else such a model is a generic one
  const bool base_shape_but_width = !generic_grid && F == 2 && L == 16 && d->density_hidden_layers == 1 && d->rgb_hidden_layers == 2 && dir_w == 16 &&
                                    d->interpolation == NRF_INTERP_LINEAR && d->density_activation == NRF_ACT_RELU && d->rgb_activation == NRF_ACT_RELU &&
                                    d->density_output_activation == NRF_ACT_NONE &&
                                    (d->rgb_output_activation == NRF_ACT_NONE || d->rgb_output_activation == NRF_ACT_SIGMOID) &&
                                    d->sigma_activation == NRF_ACT_EXPONENTIAL;
  // ... and for its depth: 64 neurons with other numbers of hidden layers (>= 1 each, at most DEPTH_MAX_WW 64 -> 64 layers in all)
  // keep the register-resident form as the DEPTH instance -- reported as hot_width 64
  // ... and (round 6) for its hidden activations: any activation of tcnn's vocabulary in either MLP keeps the register-resident form in
  // the same instance (the activation works on the fp32 accumulators, mlp_tiles_depth) -- base.json's own 1 + 2 layers included
  const bool relu_both = d->density_activation == NRF_ACT_RELU && d->rgb_activation == NRF_ACT_RELU;
  auto native_act = [](uint32_t a) {  // (nrf_device.h activate_native; Sine keeps the generic instance)
    return a == NRF_ACT_RELU || a == NRF_ACT_NONE || a == NRF_ACT_EXPONENTIAL || a == NRF_ACT_SIGMOID || a == NRF_ACT_SQUAREPLUS || a == NRF_ACT_SOFTPLUS;
  };
  const bool base_shape_but_depth = native_act(d->density_activation) && native_act(d->rgb_activation) && !generic_grid && F == 2 && L == 16 && Wn == 64 && dir_w == 16 && d->interpolation == NRF_INTERP_LINEAR &&
                                    d->density_output_activation == NRF_ACT_NONE &&
                                    (d->rgb_output_activation == NRF_ACT_NONE || d->rgb_output_activation == NRF_ACT_SIGMOID) &&
                                    d->sigma_activation == NRF_ACT_EXPONENTIAL && d->density_hidden_layers >= 1 && d->rgb_hidden_layers >= 1 &&
                                    !(d->density_hidden_layers == 1 && d->rgb_hidden_layers == 2 && relu_both) &&
                                    (d->density_hidden_layers - 1) + (d->rgb_hidden_layers - 1) <= (uint32_t)DEPTH_MAX_WW;
  const bool hot_depth = base_shape_but_depth && c->allow_width_instances;
  const uint32_t hot_width = hot_depth ? (relu_both ? 64u : HOT_WIDTH_ACT) : ((base_shape_but_width && (Wn == 16 || Wn == 32 || Wn == 128) && c->allow_width_instances) ? Wn : 0u);
  // ... and for its direction encoding: SphericalHarmonics of degree 5..8 (32..64 padded values) keeps the register-resident
  // MLPs in the persistent kernel's NET_WIDE_SH form (per-ray rows of coefficients in LDS)
  const bool wide_sh = !generic_grid && F == 2 && L == 16 && Wn == 64 && d->density_hidden_layers == 1 && d->rgb_hidden_layers == 2 &&
                       d->dir_encoding == NRF_DIR_SH && dir_w > 16 && dir_w <= 64 && d->interpolation == NRF_INTERP_LINEAR &&
                       d->density_activation == NRF_ACT_RELU && d->rgb_activation == NRF_ACT_RELU && d->density_output_activation == NRF_ACT_NONE &&
                       (d->rgb_output_activation == NRF_ACT_NONE || d->rgb_output_activation == NRF_ACT_SIGMOID) &&
                       d->sigma_activation == NRF_ACT_EXPONENTIAL && c->allow_width_instances;
  // ... and for its grid: F = 2 with fewer than 16 levels, F = 4 / 8 with at most 32 features in all, Linear, Smoothstep or -- any F = 2 /
  // 4 / 8 grid of at most 32 features, the 16 x 2 one included -- Nearest (round 5: one gather per level, grid.h:215-232), every
  // level dense / power-of-two hashed / LV_ADD_POW2 -- the GRID instances (NET_GRID2 / 4 / 8) keep base.json's MLPs in registers
  const bool hot_grid_ok = !generic_grid && (F == 1 || F == 2 || F == 4 || F == 8) && !(F == 2 && L == 16 && d->interpolation == NRF_INTERP_LINEAR) && L * F <= 32 &&
                           Wn == 64 && d->density_hidden_layers == 1 && d->rgb_hidden_layers == 2 && dir_w == 16 &&
                           (d->interpolation == NRF_INTERP_LINEAR || d->interpolation == NRF_INTERP_SMOOTHSTEP || d->interpolation == NRF_INTERP_NEAREST) &&
                           d->density_activation == NRF_ACT_RELU && d->rgb_activation == NRF_ACT_RELU && d->density_output_activation == NRF_ACT_NONE &&
                           (d->rgb_output_activation == NRF_ACT_NONE || d->rgb_output_activation == NRF_ACT_SIGMOID) &&
                           d->sigma_activation == NRF_ACT_EXPONENTIAL && c->allow_width_instances;
  const uint32_t hot_grid = hot_grid_ok ? F : 0u;
  std::vector<_Float16> frags, frags_gen, frags_hot;
  GenModel G;
  std::memset(&G, 0, sizeof(G));
  if (!generic) pack_fragments(w16, rgb_in, frags);
  if (hot_depth) pack_fragments_depth(w16, (int)d->density_hidden_layers, (int)d->rgb_hidden_layers, frags_hot);
  else if (hot_width) pack_fragments_width(w16, (int)hot_width, frags_hot);
  if (wide_sh) pack_fragments(w16, rgb_in, frags_hot);  // the wide layout: first rgb layer in RK_WIDE K steps
  if (hot_grid) pack_fragments_grid(w16, feat_w, rgb_in, F, frags_hot);
  // the generic description + fragments: the generic instance's model, and -- for a wide model -- what the stage
  // entry points nrf_encode_dir / nrf_mlp_forward run on (rows of the padded widths)
  if (generic || wide) {
    std::vector<_Float16>& fr = generic ? frags : frags_gen;
    G.F = F; G.interp = d->interpolation; G.n_levels = L; G.feat_raw = feat_raw; G.feat_w = feat_w;
    // 1: F = 2 (level_gather); 4 / 8: that F (level_gather_wide); 0: gen_level's literal index arithmetic (F = 1, Nearest, odd sizes)
    G.fast_grid = (!generic_grid && (F == 2 || F == 4 || F == 8) && (d->interpolation == NRF_INTERP_LINEAR || d->interpolation == NRF_INTERP_SMOOTHSTEP) &&
                   c->allow_gen_fast_grid) ? (F == 2 ? 1u : F) : 0u;
    G.feat_k = next_multiple(feat_w, 32u); G.width = Wn; G.dir_raw = raw; G.dir_w = dir_w; G.rgb_in = rgb_in;
    G.n_dens = d->density_hidden_layers + 1; G.n_rgb = d->rgb_hidden_layers + 1;
    const uint32_t max_k = G.feat_k > next_multiple(Wn, 32u) ? G.feat_k : next_multiple(Wn, 32u);
    G.act_stride = max_k + 8;   // +16 bytes: consecutive rows start 4 banks apart (ds_read_b128 of 16 rows: conflict-free)
    G.dir_stride = dir_w + 8;
    const _Float16* wp = w16.data();
    for (size_t i = 0; i < layers.size(); ++i) {
      G.layer[i].frag_off = (uint32_t)(fr.size() / (64 * 8));
      G.layer[i].k_steps = (layers[i].K + 31) / 32;
      G.layer[i].n_tiles = layers[i].N / 16;
      G.layer[i].act = layers[i].act;
      pack_generic_layer(wp, layers[i].N, layers[i].K, fr);
      wp += (size_t)layers[i].N * layers[i].K;
    }
  }
  const uint32_t gen_wave_bytes = (generic || wide) ? gen_dir_bytes(G) + gen_act_bytes(G) : 0u;
  if (generic && render_gen_lds_fixed_bytes(gen_wave_bytes) > 160 * 1024)
    return fail(NRF_E_UNSUPPORTED, "HIP path: this network shape needs more LDS than a CU has");
  // Device copy of the table: the reference's entries level by level; a dense level is followed by
  // res^2 + res + 1 copies of its first entries so that x + y*res + z*res^2 (at most
  // size + res^2 + res when a +1 corner sits on the x = 1 / y = 1 / z = 1 face) needs no modulo.
  std::vector<_Float16> grid16;  // F halves per entry
  grid16.reserve(n_grid + (size_t)F * (16 * 4096 + ((size_t)1 << d->log2_hashmap_size)));
  for (uint32_t l = 0; l < L; ++l) {
    LevelParams& Lv = lp[l];
    if (Lv.mode == LV_HASH_POW2 || Lv.mode == LV_ADD_POW2)  // aligned to its own (power-of-two) size: `index & mask | offset` (level_gather)
      while ((grid16.size() / F) % Lv.size != 0) grid16.push_back((_Float16)0.0f);
    Lv.offset = (uint32_t)(grid16.size() / F);
    const float* src = gp + (size_t)lv.offset[l] * F;
    for (size_t i = 0; i < (size_t)Lv.size * F; ++i) grid16.push_back((_Float16)src[i]);
    if (Lv.mode == LV_DENSE) {
      const size_t extra = (size_t)Lv.res * Lv.res + Lv.res + 1;
      for (size_t i = 0; i < extra * F; ++i) grid16.push_back((_Float16)src[i % ((size_t)Lv.size * F)]);
    }
  }
  if ((uint64_t)grid16.size() * 2 >= (1ull << 32))  // level_gather addresses the table by 32-bit byte offsets
    return fail(NRF_E_UNSUPPORTED, "hash tables of 4 GiB or more are not supported");
  // Cell-major quad copies (round 6; nrf_device.h level_gather_quad) behind the reference-order table, for the instances whose
  // network phase is network_from_lds with an F = 2 x 16 grid (the hot instance, its wide / width / depth forms): per cell
  // (x, y, z), x, y < res, z <= res, the four entries of the corners (x | x + 1, y | y + 1, z) as grid_index (grid.h:100-117)
  // names them.  The reference-order table stays: every other kernel (stage entry points, generic instance) reads it.
  // A step of the fused kernel (levels 4 jl .. 4 jl + 3, one per lane group) takes quads as a whole or not at all (steps that
  // mix the two forms run both instruction streams: measured no faster, profiles/r06/quad_sweep.txt); steps are granted in order
  // while their copies fit the budget (nrf_model_desc.gather_copy_budget_mb).  Copies that end beyond the 4 GiB a buffer
  // resource's byte offset reaches are FAR: addressed in 16-byte units from the table base (level_gather_quad_far).
  uint32_t quad_mask = 0, quad_far = 0;
  uint64_t table_bytes = (uint64_t)grid16.size() * 2;  // device bytes: the reference-order table + the quad copies (built on the device, below)
  const bool quad_shape = !generic_grid && F == 2 && L == 16 && d->interpolation == NRF_INTERP_LINEAR && !hot_grid &&
                          (!generic || hot_width || wide_sh);
  if (quad_shape) {
    uint64_t budget_mb = QUAD_BUDGET_MB_DEFAULT;  // (when the device does not say how much memory it has)
    {
      size_t mem_free = 0, mem_total = 0;  // default: a sixteenth of the device's memory (MI355X: 18 GB), and no more than half of what is free
      if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) budget_mb = std::min<uint64_t>((uint64_t)mem_total >> 24, (uint64_t)mem_free >> 21);
      else (void)hipGetLastError();
    }
    if (d->gather_copy_budget_mb) budget_mb = d->gather_copy_budget_mb;
    if (c->quad_budget_mb >= 0) budget_mb = (uint64_t)c->quad_budget_mb;
    uint64_t budget = budget_mb << 20;
    const int max_steps = c->quad_levels < 0 ? 4 : std::min(c->quad_levels, 16) / 4;
    uint64_t end_bytes = ((uint64_t)grid16.size() * 2 + 15) & ~15ull;
    for (int jl = 0; jl < max_steps; ++jl) {
      uint64_t step_bytes = 0;
      bool ok = true;
      for (int g = 0; g < 4; ++g) {
        const LevelParams& Lv = lp[4 * jl + g];
        ok = ok && (Lv.mode == LV_DENSE || Lv.mode == LV_HASH_POW2) && Lv.res >= 2 && Lv.res < 1024u;  // (res^2 << 4 < 2^24)
        step_bytes += (uint64_t)Lv.res * Lv.res * ((uint64_t)Lv.res + 1) * 16;
      }
      if (!ok || step_bytes > budget || end_bytes + step_bytes >= (1ull << 36)) continue;
      const bool far = end_bytes + step_bytes >= (1ull << 32);
      if (far && !generic && wide) continue;  // (NET_WIDE is compiled without the far form: nrf_render.h network_from_lds)
      budget -= step_bytes;
      quad_mask |= 15u << (4 * jl);
      if (far) quad_far |= 1u << jl;
      for (int g = 0; g < 4; ++g) {
        LevelParams& Lv = lp[4 * jl + g];
        const uint32_t res = Lv.res;
        Lv.q_off_b = far ? (uint32_t)(end_bytes >> 4) : (uint32_t)end_bytes;
        Lv.q_my_b = far ? res : res << 4;
        Lv.q_mz_b = far ? res * res : (res * res) << 4;
        Lv.q_max = res - 1;
        end_bytes += (uint64_t)res * res * (res + 1) * 16;
      }
    }
    table_bytes = end_bytes;
  }
  // Uploads go through the context's own stream and the device is drained afterwards: the
  // render stream is non-blocking, so a NULL-stream hipMemcpy gives no ordering against it
  // (seen on MI355X as a few stale table entries in the first frame after a reload).
  auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  // byte-offset constants of level_gather / level_gather_wide / level_gather_f1: an entry is 2 F bytes (the generic instance's
  // literal index arithmetic, gen_level, does not read them)
  const uint32_t sh_b = F == 8 ? 4u : (F == 4 ? 3u : (F == 1 ? 1u : 2u));
  for (LevelParams& L : lp) {
    const bool hashed_pow2 = L.mode == LV_HASH_POW2;
    L.off_b = L.offset << sh_b;
    if (hashed_pow2) {
      L.my_b = 2654435761u << sh_b;
      L.mz_b = 805459861u << sh_b;
    } else {  // the additive multipliers of the stride loop above (dense: res, res^2; LV_ADD_POW2: possibly wrapped / 0)
      L.my_b = (L.my_b >> 2) << sh_b;
      L.mz_b = (L.mz_b >> 2) << sh_b;
    }
    L.mask_b = (hashed_pow2 || L.mode == LV_ADD_POW2) ? ((L.size - 1) << sh_b) : 0xffffffffu;
  }
  if (hipMalloc(&c->d_grid, table_bytes) != hipSuccess) {  // no room for the copies: the reference-order table alone
    (void)hipGetLastError();
    c->d_grid = nullptr;
    for (LevelParams& Lv : lp) Lv.q_off_b = Lv.q_my_b = Lv.q_mz_b = Lv.q_max = 0;
    quad_mask = quad_far = 0;
    table_bytes = (uint64_t)grid16.size() * 2;
    HIP_TRY(hipMalloc(&c->d_grid, table_bytes));
  }
  HIP_TRY(hipMemcpyAsync(c->d_grid, grid16.data(), grid16.size() * 2, hipMemcpyHostToDevice, c->stream));
  for (uint32_t l = 0; l < L; ++l) {  // the quad copies, from the table just uploaded (same stream)
    if (!((quad_mask >> l) & 1u)) continue;
    const LevelParams& Lv = lp[l];
    const uint64_t q_bytes = ((quad_far >> (l >> 2)) & 1u) ? (uint64_t)Lv.q_off_b << 4 : (uint64_t)Lv.q_off_b;
    HIP_TRY(launch_build_quads((const char*)c->d_grid + (size_t)Lv.offset * 4, Lv.res, Lv.size, Lv.mode == LV_HASH_POW2,
                               (char*)c->d_grid + q_bytes, c->stream));
  }
  HIP_TRY(upload(&c->d_wfrag, frags.data(), frags.size() * 2));
  if (generic || wide) HIP_TRY(upload(&c->d_gen, &G, sizeof(G)));
  if (wide) HIP_TRY(upload(&c->d_wfrag_gen, frags_gen.data(), frags_gen.size() * 2));
  if (hot_width || wide_sh || hot_grid) HIP_TRY(upload(&c->d_wfrag_hot, frags_hot.data(), frags_hot.size() * 2));
  HIP_TRY(upload(&c->d_lv, lp.data(), lp.size() * sizeof(LevelParams)));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipDeviceSynchronize());

  c->desc = *d;
  c->desc.params = nullptr;
  c->desc.density_grid = nullptr;
  c->lv = lv;
  DevModel& M = c->dm;
  std::memset(&M, 0, sizeof(M));
  M.grid = (const uint32_t*)c->d_grid;
  M.grid_bytes = (uint32_t)std::min<uint64_t>(table_bytes, 0xffffffffull);  // (far quad copies lie beyond: no resource reads them)
  M.wfrag = (const uint4*)c->d_wfrag;
  M.lv = (const LevelParams*)c->d_lv;
  for (int i = 0; i < 6; ++i) M.aabb[i] = d->aabb[i];
  M.bound = d->bound;
  M.rbound = 1.0f / d->bound;
  M.pos_w = (float)(1.0 / (2 * (double)d->bound));
  {
    int e;
    M.pos_w_pow2 = std::frexp(M.pos_w, &e) == 0.5f ? 1u : 0u;
  }
  M.cascade = d->cascade;
  M.H = d->density_grid_size;
  M.n_levels = d->n_levels;
  M.dir_encoding = d->dir_encoding;
  M.sh_degree = d->sh_degree;
  M.n_frequencies = d->n_frequencies;
  M.density_activation = d->density_activation;
  M.density_output_activation = d->density_output_activation;
  M.sigma_activation = d->sigma_activation;
  M.rgb_activation = d->rgb_activation;
  M.rgb_output_activation = d->rgb_output_activation;
  M.uni_modes = 0;
  for (int jl = 0; jl < 4; ++jl) {
    bool all_dense = true, all_hash = true;
    for (int g = 0; g < 4; ++g) {
      if ((uint32_t)(4 * jl + g) >= L) continue;  // (a level the grid does not have: its lanes are masked, grid_features)
      all_dense = all_dense && lp[4 * jl + g].mode == LV_DENSE;
      all_hash = all_hash && lp[4 * jl + g].mode == LV_HASH_POW2;
    }
    M.uni_modes |= (all_dense ? 1u : (all_hash ? 2u : 0u)) << (2 * jl);
  }
  M.quad_mask = quad_mask;
  M.quad_far = quad_far;
  c->table_bytes = table_bytes;
  c->table_ref_bytes = (uint64_t)grid16.size() * 2;
  c->gather_addresses = 0;
  for (uint32_t l = 0; l < L; ++l) c->gather_addresses += ((quad_mask >> l) & 1u) ? 2u : (d->interpolation == NRF_INTERP_NEAREST ? 1u : 8u);
  M.generic = generic ? 1u : 0u;
  M.wide = (!generic && wide) ? 1u : 0u;
  M.gen = (const GenModel*)c->d_gen;
  M.gen_wave_bytes = gen_wave_bytes;
  M.gen_frag_bytes = generic ? (uint32_t)(frags.size() * 2) : 0u;
  M.hot_width = hot_width;
  M.depth_xd = hot_depth ? d->density_hidden_layers - 1 : 0u;
  M.depth_xr = hot_depth ? d->rgb_hidden_layers - 1 : 0u;
  M.wfrag_hot = (const uint4*)c->d_wfrag_hot;
  c->model_hot_width = hot_width;
  c->model_hot_grid = hot_grid;
  M.hot_grid = hot_grid;
  M.grid_smooth = d->interpolation == NRF_INTERP_SMOOTHSTEP ? 1u : 0u;
  M.grid_nearest = d->interpolation == NRF_INTERP_NEAREST ? 1u : 0u;
  c->model_wide_sh = wide_sh;
  M.wide_sh = wide_sh ? 1u : 0u;
  M.dir_w = dir_w;
  c->gen = G;
  // the density grid of the snapshot (nerf_render.cu:447-466) -- or none yet: nrf_generate_density_grid evaluates it
  // from the network (NerfRender::generate_density_grid); until then the model cannot be rendered
  if (d->density_grid) {
    rc = set_density_grid(c, d->density_grid, d->mean_density);
    if (rc) { free_model(c); return rc; }
    c->grid_missing = false;
  } else {
    std::vector<float> empty((size_t)cells, 0.0f);
    rc = set_density_grid(c, empty.data(), d->mean_density);
    if (rc) { free_model(c); return rc; }
    c->grid_missing = true;
  }
  c->model_loaded = true;
  // the code objects of the render kernel's families are loaded here, not by the first frame (once per process and device)
  // (group members and server workers load models concurrently: one flag per device, taken under a lock)
  static std::once_flag preloaded[64];
  if (c->device >= 0 && c->device < 64) std::call_once(preloaded[c->device], [] { preload_kernels(true); });
  return NRF_OK;
}

// NerfRender::generate_density_grid (R/src/nerf_render.cu:388-429; dead and incomplete in the reference: the density
// query is commented out at :415).  What it sets out to do (torch-ngp's update_extra_state, which it restates), made
// whole: for every cascade the density at every cell's position (init_xyzs + dd_scale, perturbation off), scaled by
// 0.001691, folded into a grid that starts at 1/64 with g = max(g * decay, value), n_iterations times; mean_density =
// mean of max(g, 0).  The march tables are rebuilt from the result.
int nrf_generate_density_grid(nrf_context* c, int n_iterations, float decay, float* mean_density_out) {
  int rc = need_model(c);
  if (rc) return rc;
  if (n_iterations < 1 || !(decay > 0.0f) || !(decay <= 1.0f)) return fail(NRF_E_INVALID, "n_iterations >= 1 and 0 < decay <= 1 required");
  const uint32_t H = c->desc.density_grid_size, C = c->desc.cascade;
  const uint64_t n = (uint64_t)H * H * H;
  if (n >= (1ull << 31)) return fail(NRF_E_UNSUPPORTED, "density grid too large");
  void* buf = nullptr;  // xyz [n][3] | dir [n][3] | rgb [n][3] | sigma [n] | grid [n]
  HIP_TRY(hipMalloc(&buf, n * 11 * sizeof(float)));
  float* d_xyz = (float*)buf;
  float* d_dir = d_xyz + 3 * n;
  float* d_rgb = d_dir + 3 * n;
  float* d_sigma = d_rgb + 3 * n;
  float* d_cell = d_sigma + n;
  std::vector<float> grid((size_t)n * C);
  hipError_t e = hipSuccess;
  for (uint32_t cas = 0; cas < C && e == hipSuccess; ++cas) {
    const float bound = (float)(1u << cas) < c->desc.bound ? (float)(1u << cas) : c->desc.bound;  // nerf_render.cu:409
    const float half_grid_size = bound / (float)H;
    e = launch_density_positions(H, bound - half_grid_size, d_xyz, d_dir, c->stream);
    if (e == hipSuccess) e = launch_network(c->dm, d_xyz, d_dir, (uint32_t)n, d_sigma, d_rgb, c->stream);
    if (e == hipSuccess) e = launch_density_update(d_sigma, (uint32_t)n, decay, n_iterations, d_cell, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(grid.data() + (size_t)cas * n, d_cell, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  (void)hipFree(buf);
  if (e != hipSuccess) return hip_fail(e, "nrf_generate_density_grid");
  double sum = 0.0;  // sequential, in cell order: the same number on every run (and in the oracle)
  for (float g : grid) sum += g > 0.0f ? (double)g : 0.0;
  const float mean = (float)(sum / (double)grid.size());
  rc = set_density_grid(c, grid.data(), mean);
  if (rc) return rc;
  c->grid_missing = false;
  if (mean_density_out) *mean_density_out = mean;
  return NRF_OK;
}

int nrf_read_density_grid(nrf_context* c, float* grid, uint64_t n, float* mean_density) {
  int rc = need_model(c);
  if (rc) return rc;
  if (grid) {
    if (n != c->host_grid.size()) return fail(NRF_E_INVALID, "n must be cascade * H^3");
    std::memcpy(grid, c->host_grid.data(), c->host_grid.size() * sizeof(float));
  }
  if (mean_density) *mean_density = c->desc.mean_density;
  return NRF_OK;
}

int nrf_set_resolution(nrf_context* c, int width, int height) {
  if (!c || width <= 0 || height <= 0) return fail(NRF_E_INVALID, "bad resolution");
  int rc = set_device(c);
  if (rc) return rc;
  c->W = width;
  c->H = height;
  return alloc_frame(c);
}

int nrf_set_options(nrf_context* c, const nrf_options* o) {
  if (!c || !o) return fail(NRF_E_INVALID, "null argument");
  if (o->shard_count < 1 || o->shard_index < 0 || o->shard_index >= o->shard_count)
    return fail(NRF_E_INVALID, "bad shard");
  if (o->perturb < 0) return fail(NRF_E_INVALID, "perturb must be >= 0 (0: off, the reference's m_perturb = false; > 0: the seed, render_utils.h:550)");
  if (o->max_steps < 1) return fail(NRF_E_INVALID, "max_steps must be >= 1");
  int rc = set_device(c);
  if (rc) return rc;
  c->opt = *o;
  return alloc_frame(c);
}

int nrf_set_max_views(nrf_context* c, int max_views) {
  if (!c || max_views < 1) return fail(NRF_E_INVALID, "max_views must be >= 1");
  int rc = set_device(c);
  if (rc) return rc;
  c->max_views = max_views;
  return alloc_frame(c);
}

}  // extern "C"

namespace {
// the rows [lo, hi) of a frame that the strip rows of a view's region of interest cover (what launch_render queues for the
// persistent kernel); every pixel outside them is the background
void roi_rows(const int roi[4], int H, int& lo, int& hi) {
  lo = hi = 0;
  if (roi[2] < roi[0] || roi[3] < roi[1]) return;
  const int tiles_y = (H + 7) / 8;
  const int ty0 = std::max(roi[1] >> 3, 0), ty1 = std::min(roi[3] >> 3, tiles_y - 1);
  if (ty1 < ty0) return;
  lo = 8 * ty0;
  hi = std::min(H, 8 * (ty1 + 1));
}

// the columns [x0, x1) of whole tiles a view's region of interest touches: the tiles beside them are background (the kernel tests
// every tile's 8x8 pixels against the region)
void roi_cols(const int roi[4], int W, int& x0, int& x1) {
  x0 = x1 = 0;
  if (roi[2] < roi[0] || roi[3] < roi[1]) return;
  const int tiles_x = (W + 7) / 8;
  const int tx0 = std::max(roi[0] >> 3, 0), tx1 = std::min(roi[2] >> 3, tiles_x - 1);
  if (tx1 < tx0) return;
  x0 = 8 * tx0;
  x1 = std::min(W, 8 * (tx1 + 1));
}

// One launch per NRF_MAX_VIEWS cameras; all launches of a call go to the same stream back to back.  Every call takes the
// next slot of the context's ring of statistics counters + work queues (cleared on the call's own stream), so calls of
// one context that overlap on different streams never share a queue.
// rows_out (optional): per view the rows [lo, hi) and the columns [x0, x1) (whole tiles) its region of interest covers.
constexpr float MAX_CAMERA_DISTANCE = 4096.0f;  // in the reference's ngp units (0.33 x the nerf pose's + 0.5), see render_views_impl
struct ProgressArgs {
  unsigned* done;   // device [n_views][tiles_y], zeroed on the stream before the launches
  unsigned* flags;  // pinned host [n_views][tiles_y]
  unsigned epoch;
};
int render_views_impl(nrf_context* c, int n_views, const float* cams, const float* poses, hipStream_t st, void* rgba, void* depth,
                      size_t stride_px, int out_mode, int skip_outside, int* rows_out, const ProgressArgs* prog = nullptr) {
  FrameParams P;
  fill_frame_params(c, cams, poses, P);
  P.out_mode = out_mode;
  P.skip_outside = skip_outside;
  // one or two views alone are latency-bound, not throughput-bound: their last tiles end sooner when every ray queues its full
  // eight samples per round, and the few samples evaluated for nothing cost nobody anything (0.904 against 0.914 ms per 1080p view)
  // ... unless the launch has fewer tiles than the chip has waves: it is ALL tail (idle waves take rays off the rendering ones from
  // the first round on, and every split group queues its own eight samples per ray behind a terminating one) -- small frames
  // keep the transmittance-dependent queue, which holds their evaluated samples within 15 % of the composited ones
  // (tests/test_parity_gpu.py, test_generic_gpu.py, test_golden.py; pixels cannot depend on it: tests/test_persistent_gpu.py)
  const bool all_tail = (long long)c->n_local_tiles * n_views < (long long)c->dm.n_cus * std::max(1u, c->dm.persist_waves);
  if (n_views <= 2 && !c->sample_cap_forced && !all_tail) P.sample_cap = 0;
  c->call_index = (c->call_index + 1) % CALL_RING;
  char* counters = call_slot(c, c->call_index);
  unsigned* plan = (c->plan_max_pos > 0 && prog == nullptr) ? (unsigned*)((char*)c->d_plan + (size_t)c->call_index * PLAN_BYTES) : nullptr;
  HIP_TRY(hipEventRecord(c->ev0, st));  // (render_ms covers the clearing of the call's counters and the planning of its queues)
  // views per launch: NRF_MAX_VIEWS, fewer when the frames are so large that the persistent kernel's 24-bit queue positions
  // (strip rows of all views x strips per row) would not hold the launch (8K frames: 64 views)
  int per_launch = MAX_VIEWS;
  {
    const long long per_view = (long long)P.tiles_y * ((P.tiles_x + 3) / 4);
    if (per_view * per_launch >= 0xffffff) per_launch = (int)std::max(1LL, 0xfffffeLL / std::max(per_view, 1LL));
  }
  const size_t px_bytes_a = out_mode == OUT_U8 ? 3 : 16, px_bytes_b = out_mode == OUT_U8 ? 1 : 4;
  for (int first = 0; first < n_views; first += per_launch) {
    ViewBatch VB;
    std::memset(&VB, 0, sizeof(VB));
    VB.n_views = n_views - first < per_launch ? n_views - first : per_launch;
    VB.view_stride_px = stride_px;
    for (int v = 0; v < VB.n_views; ++v) {
      nerf_matrix_to_ngp(poses + 16 * (size_t)(first + v), c->desc.scale, VB.v[v].R, VB.v[v].org);
      for (int i = 0; i < 4; ++i) VB.v[v].cam[i] = cams[4 * (size_t)(first + v) + i];
      view_roi(VB.v[v].R, VB.v[v].org, VB.v[v].cam, c->dm.occ_box, c->W, c->H, VB.v[v].roi);
      {
        // A camera thousands of scene sizes away: t + dt == t in fp32 once t passes ~2^24 dt (dt_min = 0.0034: t ~ 5.7e4), the
        // march of render_utils.h:593-653 stops advancing and never ends (the reference hangs there; so would this kernel).
        // Long before that the object is far below a pixel: such a view is background (an empty region of interest).
        const float* o = VB.v[v].org;
        const float far2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
        if (!(far2 <= MAX_CAMERA_DISTANCE * MAX_CAMERA_DISTANCE)) {
          VB.v[v].roi[0] = VB.v[v].roi[1] = 0;
          VB.v[v].roi[2] = VB.v[v].roi[3] = -1;
        }
      }
      if (rows_out) {  // {row lo, row hi, column lo, column hi} per view
        int* ro = rows_out + 4 * (size_t)(first + v);
        roi_rows(VB.v[v].roi, c->H, ro[0], ro[1]);
        roi_cols(VB.v[v].roi, c->W, ro[2], ro[3]);
      }
    }
    if (prog) {  // (the kernel indexes its progress arrays by the launch's own view numbers)
      P.prog_done = prog->done + (size_t)first * P.tiles_y;
      P.prog_flags = prog->flags + (size_t)first * P.tiles_y;
      P.prog_epoch = (int)prog->epoch;
    }
    HIP_TRY(launch_render(c->dm, P, VB, rgba ? (char*)rgba + (size_t)first * stride_px * px_bytes_a : nullptr,
                          (char*)depth + (size_t)first * stride_px * px_bytes_b, counters, st, first == 0, plan,
                          (unsigned)c->plan_max_pos));
  }
  c->last_views = n_views;
  HIP_TRY(hipEventRecord(c->ev1, st));
  c->last_stream = st;
  c->rendered = true;
  return NRF_OK;
}

int check_renderable(nrf_context* c, const float* cams, const float* poses, int n_views) {
  int rc = need_model(c);
  if (rc) return rc;
  if (!cams || !poses) return fail(NRF_E_INVALID, "null argument");
  if (n_views < 1) return fail(NRF_E_INVALID, "n_views must be >= 1");
  if (c->W <= 0 || !c->d_rgba) return fail(NRF_E_STATE, "set_resolution has not been called");
  if (c->grid_missing)
    return fail(NRF_E_STATE, "the model was loaded without a density grid: call nrf_generate_density_grid first");
  return NRF_OK;
}
}  // namespace

extern "C" {

int nrf_render_views(nrf_context* c, int n_views, const float* cams, const float* poses, void* stream, nrf_frame* out) {
  int rc = check_renderable(c, cams, poses, n_views);
  if (rc) return rc;
  const bool bound = c->bound_rgba || c->bound_rgbd8 || c->bound_rgb8;
  if (!bound && n_views > c->max_views)
    return fail(NRF_E_STATE, "more views than the context's buffers hold: call nrf_set_max_views or nrf_bind_output");
  if (c->bound_rgb8 && tiled_layout(c))
    return fail(NRF_E_STATE, "8-bit planar output (nrf_bind_output_u8) needs a single-shard (row-major) frame");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  void* rgba = c->bound_rgba ? c->bound_rgba : c->d_rgba;
  void* depth = c->bound_depth ? c->bound_depth : c->d_depth;
  int mode = OUT_F32;
  if (c->bound_rgbd8) {  // 4 bytes per pixel where the depth plane would be (store_pixel)
    rgba = nullptr;
    depth = c->bound_rgbd8;
    mode = OUT_RGBD8;
  } else if (c->bound_rgb8) {
    rgba = c->bound_rgb8;
    depth = c->bound_depth8;
    mode = OUT_U8;
  }
  rc = render_views_impl(c, n_views, cams, poses, st, rgba, depth, c->n_out_px, mode, 0, nullptr);
  if (rc) return rc;
  const bool floats = mode == OUT_F32;
  c->last_rgba = floats ? rgba : nullptr;  // (an 8-bit frame in a bound buffer is the caller's to read)
  c->last_depth = floats ? depth : nullptr;
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  if (out) {
    out->width = c->W;
    out->height = c->H;
    out->n_tiles = c->n_local_tiles;
    out->rgba = floats ? rgba : nullptr;
    out->depth = floats ? depth : nullptr;
    out->tile_major = tiled_layout(c);
    out->n_views = n_views;
    out->view_stride_px = (int64_t)c->n_out_px;
  }
  return NRF_OK;
}

// ---- host frames: the reference's render_frame ends in HOST memory (R/src/nerf_render.cu:345-359: D2H of the float
// planes, then a single-threaded quantise / de-interleave loop per GPU).  Here the kernel writes the 8-bit Image itself
// (OUT_U8), the rows of the view's region of interest travel by one asynchronous copy per plane into pinned memory the
// context owns, and the rows outside it -- background by construction -- are filled by the calling thread while the GPU
// renders (only those that held something else: a camera that moves a little costs a few rows).  Two slots: the copy of
// one call overlaps the render of the next.
static uint8_t host_quant_u8(float v) {  // quant_u8 of nrf_render.h
  const double s = 255.0 * (double)v;
  if (!(s > 0.0)) return 0;
  if (s >= 255.0) return 255;
  return (uint8_t)s;
}

namespace {
// copies the rows [lo, hi) of view v of a host-frame slot (both planes) on the context's copy stream; x0 < x1: only the
// columns [x0, x1) of those rows (a pitched copy: scripts/copy2d_probe.py measured 46.5 GB/s for 80 % of a 1080p frame's width
// against 49.1 GB/s for whole rows -- 0.85 of the time)
int copy_rows(nrf_context* c, nrf_context::HostSlot& h, int v, int lo, int hi, int x0 = 0, int x1 = 0) {
  if (hi <= lo) return NRF_OK;
  const size_t Wb = (size_t)h.W, px = h.px, depth_off = h.views * px * 3;
  const bool cols = x1 > x0 && (x0 > 0 || x1 < h.W);
  const size_t ro = ((size_t)v * px + (size_t)lo * Wb) * 3, rn = (size_t)(hi - lo) * Wb * 3;
  const size_t dofs = depth_off + (size_t)v * px + (size_t)lo * Wb, dn = (size_t)(hi - lo) * Wb;
  if (cols) {
    const size_t wpx = (size_t)(x1 - x0), n_rows = (size_t)(hi - lo);
    HIP_TRY(hipMemcpy2DAsync(h.h_buf + ro + (size_t)x0 * 3, Wb * 3, (const uint8_t*)h.d_buf + ro + (size_t)x0 * 3, Wb * 3, wpx * 3, n_rows,
                             hipMemcpyDeviceToHost, c->copy_stream));
    h.copied += wpx * 3 * n_rows;
    if (h.with_depth) {
      HIP_TRY(hipMemcpy2DAsync(h.h_buf + dofs + x0, Wb, (const uint8_t*)h.d_buf + dofs + x0, Wb, wpx, n_rows, hipMemcpyDeviceToHost, c->copy_stream));
      h.copied += wpx * n_rows;
    }
    return NRF_OK;
  }
  HIP_TRY(hipMemcpyAsync(h.h_buf + ro, (const uint8_t*)h.d_buf + ro, rn, hipMemcpyDeviceToHost, c->copy_stream));
  h.copied += rn;
  if (h.with_depth) {
    HIP_TRY(hipMemcpyAsync(h.h_buf + dofs, (const uint8_t*)h.d_buf + dofs, dn, hipMemcpyDeviceToHost, c->copy_stream));
    h.copied += dn;
  }
  return NRF_OK;
}

// The copies of a progressive call, issued by the waiting thread while the render is still running: a band of strip rows
// goes to the copy engine as soon as the kernel has flagged all of its rows (the bytes are in memory by then: write-through
// stores, acknowledged before the row was counted -- nrf_render.h tile_written); whatever is left when the kernel's end
// event fires is copied then.  The loop ends with the kernel at the latest: it cannot wait for a flag that never comes.
int progressive_copies(nrf_context* c, nrf_context::HostSlot& h) {
  struct Band { int view, lo, hi, s0, s1; };  // pixel rows [lo, hi) = strip rows [s0, s1) of the view
  std::vector<Band> bands;
  const int tiles_y = (h.H + 7) / 8;
  // ~16 bands per call (one view alone: bands of ~70 rows at 1080p; a batch of 16 views: one band per view), but no copy
  // below 64 KiB: the runtime moves smaller ones with a blit KERNEL, which gets no compute unit while the persistent render
  // is resident -- it, and every copy queued behind it, would wait for the render's end (scripts/copy_overlap_probe2.py:
  // 16 KiB copies issued during a 13 ms render all ended with it, 64 KiB ones ran beside it).  The smallest copy of a band
  // is its depth plane (W bytes per row; rgb-only frames: 3 W).
  long total = 0;
  for (int v = 0; v < h.n_views; ++v) total += (h.rows[4 * v + 1] + 7) / 8 - h.rows[4 * v] / 8;
  const long want_rows = std::max(4L, (total + 15) / 16);  // strip rows per band
  const long row_bytes = (long)h.W * (h.with_depth ? 1 : 3);
  const long min_rows = (65536 + 8 * row_bytes - 1) / (8 * row_bytes);  // strip rows whose smallest plane is 64 KiB
  for (int v = 0; v < h.n_views; ++v) {
    const int lo = h.rows[4 * v], hi = h.rows[4 * v + 1];
    if (hi <= lo) continue;
    const long s_lo = lo / 8, s_hi = (hi + 7) / 8, n = s_hi - s_lo;
    const long per = std::max(want_rows, min_rows);
    const long n_bands = std::max(1L, n / per);  // (the remainder is spread over the bands: none is smaller than `per`)
    for (long b = 0; b < n_bands; ++b) {
      const int s0 = (int)(s_lo + n * b / n_bands), s1 = (int)(s_lo + n * (b + 1) / n_bands);
      bands.push_back({v, std::max(lo, 8 * s0), std::min(hi, 8 * s1), s0, s1});
    }
  }
  // A copy costs the engine ~12 us before its first byte moves (a 550 KB band: 34 us for 10 us of link time), and the rows of
  // a frame rendered alone complete within the last tenth of its render: band by band the copies ran for 0.3 ms after the
  // kernel's end.  So ready bands are issued in GROUPS: adjacent ones merged into one copy per plane, at most two groups in
  // flight -- while the engine is busy the ready bands collect, and what completes together leaves as one copy.
  size_t remaining = bands.size();
  bool kernel_done = false;
  unsigned spins = 0;
  static const bool debug = std::getenv("NRF_HOST_DEBUG") != nullptr;
  std::vector<hipEvent_t> dbg_events;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  std::vector<char> ready(bands.size(), 0);
  size_t n_ready = 0;     // ready, not yet issued
  int in_flight = 0;      // groups whose event has not been seen done (oldest: grp[(grp_next + 2 - in_flight) % 2])
  int grp_next = 0;
  const int max_groups = c->host_merge ? 2 : 1 << 30;
  while (remaining) {
    if (!kernel_done) {
      for (size_t i = 0; i < bands.size(); ++i) {
        const Band& b = bands[i];
        if (b.view < 0 || ready[i]) continue;
        bool ok = true;
        const unsigned* f = h.h_flags + (size_t)b.view * tiles_y;
        for (int r = b.s0; r < b.s1 && ok; ++r) ok = __atomic_load_n(f + r, __ATOMIC_ACQUIRE) == h.epoch;
        if (ok) { ready[i] = 1; ++n_ready; }
      }
    } else if (n_ready < remaining) {
      for (size_t i = 0; i < bands.size(); ++i) if (bands[i].view >= 0 && !ready[i]) { ready[i] = 1; ++n_ready; }
    }
    while (in_flight > 0 && c->host_merge) {
      const hipError_t e = hipEventQuery(h.grp[(grp_next + 2 - in_flight) % 2]);
      if (e == hipSuccess) --in_flight;
      else if (e == hipErrorNotReady) break;
      else return hip_fail(e, "hipEventQuery");
    }
    if (n_ready && in_flight < max_groups) {
      for (size_t i = 0; i < bands.size();) {
        if (bands[i].view < 0 || !ready[i]) { ++i; continue; }
        size_t j = i;  // [i, j]: ready bands of one view whose rows follow each other
        while (c->host_merge && j + 1 < bands.size() && bands[j + 1].view == bands[i].view && ready[j + 1] && bands[j + 1].s0 == bands[j].s1) ++j;
        if (debug) std::fprintf(stderr, "[host frame] +%.3f ms: view %d rows %d..%d (%zu band(s))%s\n", since(), bands[i].view, bands[i].lo, bands[j].hi, j - i + 1, kernel_done ? " (kernel done)" : "");
        hipEvent_t d0 = nullptr, d1 = nullptr;
        if (debug) { (void)hipEventCreate(&d0); (void)hipEventCreate(&d1); (void)hipEventRecord(d0, c->copy_stream); }
        int rc = copy_rows(c, h, bands[i].view, bands[i].lo, bands[j].hi);
        if (rc) return rc;
        if (debug) { (void)hipEventRecord(d1, c->copy_stream); dbg_events.push_back(d0); dbg_events.push_back(d1); }
        for (size_t k = i; k <= j; ++k) { bands[k].view = -1; ready[k] = 0; --remaining; --n_ready; }
        i = j + 1;
      }
      if (c->host_merge && remaining) {
        HIP_TRY(hipEventRecord(h.grp[grp_next], c->copy_stream));
        grp_next ^= 1;
        ++in_flight;
      }
      continue;
    }
    if (!remaining) break;
    if (!kernel_done && (++spins & 31u) == 0u) {
      const hipError_t e = hipEventQuery(h.t1);
      if (e == hipSuccess) kernel_done = true;
      else if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery");
    } else {
      __builtin_ia32_pause();
    }
  }
  HIP_TRY(hipEventRecord(h.done, c->copy_stream));
  h.copies_issued = true;
  if (debug) {  // when the copy engine really moved each band, relative to the start of the render
    (void)hipEventSynchronize(h.done);
    for (size_t i = 0; i + 1 < dbg_events.size(); i += 2) {
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, h.t0, dbg_events[i]);
      (void)hipEventElapsedTime(&b, h.t0, dbg_events[i + 1]);
      std::fprintf(stderr, "[host frame] copy %zu ran %.3f .. %.3f ms after the render's start\n", i / 2, a, b);
      (void)hipEventDestroy(dbg_events[i]);
      (void)hipEventDestroy(dbg_events[i + 1]);
    }
  }
  return NRF_OK;
}
}  // namespace

int nrf_submit_host_u8(nrf_context* c, int n_views, const float* cams, const float* poses, int flags, int* ticket) {
  int rc = check_renderable(c, cams, poses, n_views);
  if (rc) return rc;
  if (!ticket) return fail(NRF_E_INVALID, "null argument");
  if (tiled_layout(c)) return fail(NRF_E_STATE, "host frames need a single-shard (row-major) frame");
  const int si = c->hs_next;
  nrf_context::HostSlot& h = c->hs[si];
  // the slot's previous frames are overwritten (nerfhip.h: valid until the second next submit); a call nobody waited for
  // has no copy in flight -- its render precedes this one on the context's stream
  if (h.pending && h.copies_issued) HIP_TRY(hipEventSynchronize(h.done));
  h.pending = false;
  const size_t px = (size_t)c->W * c->H;
  const int tiles_y = (c->H + 7) / 8;
  // keyed on the frame's GEOMETRY, not its pixel count: 1920x1080 -> 1080x1920 keeps W * H but changes the row bookkeeping
  // (row_lo / row_hi / bg describe byte ranges of the old width) and the number of strip rows (d_done / h_flags entries)
  if (h.px != px || h.W != c->W || h.H != c->H || h.views < (size_t)n_views) {
    HIP_TRY(hipDeviceSynchronize());
    for (void* q : {h.d_buf, (void*)h.d_done}) if (q) (void)hipFree(q);
    for (void* q : {(void*)h.h_buf, (void*)h.h_flags}) if (q) (void)hipHostFree(q);
    h.d_buf = nullptr;
    h.h_buf = nullptr;
    h.d_done = h.h_flags = nullptr;
    h.views = std::max((size_t)n_views, (size_t)c->max_views);
    h.px = px;
    h.prog_entries = h.views * (size_t)tiles_y;
    void* hp = nullptr;
    HIP_TRY(hipHostMalloc(&hp, h.views * px * 4, hipHostMallocPortable));
    h.h_buf = (uint8_t*)hp;
    HIP_TRY(hipMalloc(&h.d_buf, h.views * px * 4));
    HIP_TRY(hipMalloc((void**)&h.d_done, h.prog_entries * 4));
    HIP_TRY(hipHostMalloc(&hp, h.prog_entries * 4, hipHostMallocPortable | hipHostMallocMapped));
    h.h_flags = (unsigned*)hp;
    std::memset(h.h_flags, 0, h.prog_entries * 4);
    h.epoch = 0;
    h.row_lo.assign(h.views, 0);
    h.row_hi.assign(h.views, c->H);  // unknown content: everything counts as "not background"
    h.col_lo.assign(h.views, 0);
    h.col_hi.assign(h.views, c->W);
    h.bg = -1;
  }
  h.W = c->W;
  h.H = c->H;
  h.n_views = n_views;
  h.with_depth = !(flags & NRF_HOST_RGB_ONLY);
  h.copied = 0;
  h.copies_issued = false;
  // progress reporting needs the persistent form of the kernel (launch_render's choice for this model)
  h.progressive = c->host_progressive && c->dm.persistent && c->dm.lds_coarse_words > 0;
  {
    // A launch whose queue order is PLANNED (one or two views: dearest strips first, nrf_kernels.hip "queue planning") completes
    // its rows within the last tenth of the render -- nothing to copy meanwhile, and copies that are already queued behind the
    // render's event start sooner than ones the waiting thread issues when it sees the flags (one 1080p view: 1.09 against
    // 1.14 ms per call; three views and more, which are not planned: 2.54 against 2.98 the other way round)
    // ... and two views gain next to nothing from the plan on the device (1.64 against 1.67 ms) while it costs their host end
    // the progressive copies (2.00 against 1.82 ms per call): only a frame rendered alone is planned here, every larger call
    // keeps the queue order of its views and copies them as they complete (render_views_impl: no plan with progress reporting)
    const long strips = (long)n_views * tiles_y * ((c->W + 31) / 32);
    if (n_views == 1 && c->plan_max_pos > 0 && strips <= (long)c->plan_max_pos) h.progressive = false;
  }
  const size_t depth_off = h.views * px * 3;  // depth planes follow the rgb planes of ALL views the slot holds
  h.rows.assign((size_t)4 * n_views, 0);
  ProgressArgs prog{h.d_done, h.h_flags, 0u};
  if (h.progressive) {
    prog.epoch = ++h.epoch;
    HIP_TRY(hipMemsetAsync(h.d_done, 0, (size_t)n_views * tiles_y * 4, c->stream));
  }
  HIP_TRY(hipEventRecord(h.t0, c->stream));
  rc = render_views_impl(c, n_views, cams, poses, c->stream, h.d_buf, (uint8_t*)h.d_buf + depth_off, px, OUT_U8, c->host_skip_outside ? 1 : 0, h.rows.data(),
                         h.progressive ? &prog : nullptr);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h.t1, c->stream));
  c->last_rgba = c->last_depth = nullptr;
  if (!h.progressive) {  // every copy after the render's end
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, h.t1, 0));
    for (int v = 0; v < n_views; ++v) {
      // (the copies of a call that is not progressive start after the render: only the region's columns travel.  Progressive
      //  copies run BESIDE the render, where a pitched copy the runtime chose to move with a blit kernel would find no compute
      //  unit free -- they keep whole rows)
      rc = copy_rows(c, h, v, h.rows[4 * v], h.rows[4 * v + 1], c->host_cols ? h.rows[4 * v + 2] : 0, c->host_cols ? h.rows[4 * v + 3] : 0);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(h.done, c->copy_stream));
    h.copies_issued = true;
  }
  // while the GPU works: the pixels outside the regions of interest.  The pinned planes hold the background value except in
  // the rectangle the copies of earlier calls have written (row_lo / row_hi x col_lo / col_hi), so only that rectangle's
  // difference to the new one is filled.
  {
    const int bg = host_quant_u8(c->opt.bg_color);
    const bool all = h.bg != bg;
    const size_t Wb = (size_t)c->W;
    auto fill = [&](int v, int r0, int r1, int c0, int c1) {
      if (r1 <= r0 || c1 <= c0) return;
      if (c0 == 0 && c1 == c->W) {
        std::memset(h.h_buf + ((size_t)v * px + (size_t)r0 * Wb) * 3, bg, (size_t)(r1 - r0) * Wb * 3);
        std::memset(h.h_buf + depth_off + (size_t)v * px + (size_t)r0 * Wb, 0, (size_t)(r1 - r0) * Wb);  // depth of a missed ray: 0
        return;
      }
      for (int r = r0; r < r1; ++r) {
        std::memset(h.h_buf + ((size_t)v * px + (size_t)r * Wb + (size_t)c0) * 3, bg, (size_t)(c1 - c0) * 3);
        std::memset(h.h_buf + depth_off + (size_t)v * px + (size_t)r * Wb + (size_t)c0, 0, (size_t)(c1 - c0));
      }
    };
    const bool use_cols = c->host_cols && !h.progressive;  // (what this call's copies write: the region's columns, or whole rows)
    for (int v = 0; v < (int)h.views; ++v) {
      if (v < n_views) {
        const int lo = h.rows[4 * v], hi = h.rows[4 * v + 1];
        int x0 = use_cols ? h.rows[4 * v + 2] : 0, x1 = use_cols ? h.rows[4 * v + 3] : c->W;
        if (x1 <= x0) { x0 = 0; x1 = c->W; }
        const int plo = all ? 0 : h.row_lo[v], phi = all ? c->H : h.row_hi[v];
        const int pc0 = all ? 0 : h.col_lo[v], pc1 = all ? c->W : h.col_hi[v];
        if (hi <= lo) fill(v, plo, phi, pc0, pc1);
        else {
          fill(v, plo, std::min(phi, lo), pc0, pc1);
          fill(v, std::max(plo, hi), phi, pc0, pc1);
          const int m0 = std::max(plo, lo), m1 = std::min(phi, hi);  // the rows both rectangles share: the columns beside the new one
          fill(v, m0, m1, pc0, std::min(pc1, x0));
          fill(v, m0, m1, std::max(pc0, x1), pc1);
        }
        h.row_lo[v] = lo;
        h.row_hi[v] = hi;
        h.col_lo[v] = x0;
        h.col_hi[v] = x1;
      } else if (all) {  // (not part of this call: stays as it is, counted as unknown)
        h.row_lo[v] = 0;
        h.row_hi[v] = c->H;
        h.col_lo[v] = 0;
        h.col_hi[v] = c->W;
      }
    }
    h.bg = bg;
  }
  h.pending = true;
  c->hs_next = (si + 1) % HOST_SLOTS;
  *ticket = si;
  return NRF_OK;
}

int nrf_wait_host_u8(nrf_context* c, int ticket, nrf_host_frame* out) {
  if (!c || ticket < 0 || ticket >= HOST_SLOTS) return fail(NRF_E_INVALID, "bad ticket");
  nrf_context::HostSlot& h = c->hs[ticket];
  if (!h.h_buf || h.n_views < 1) return fail(NRF_E_STATE, "nothing was submitted with this ticket");
  int rc = set_device(c);
  if (rc) return rc;
  if (h.pending) {
    if (!h.copies_issued) {
      rc = progressive_copies(c, h);
      if (rc) return rc;
    }
    // the last copy is a few tens of microseconds away: poll before blocking
    hipError_t e = hipErrorNotReady;
    for (int i = 0; i < 4000 && e == hipErrorNotReady; ++i) e = hipEventQuery(h.done);
    if (e == hipErrorNotReady) e = hipEventSynchronize(h.done);
    if (e != hipSuccess) return hip_fail(e, "waiting for the host frame's copies");
  }
  h.pending = false;
  if (out) {
    out->width = h.W;
    out->height = h.H;
    out->n_views = h.n_views;
    out->rgb = h.h_buf;
    out->depth = h.with_depth ? h.h_buf + h.views * h.px * 3 : nullptr;
    out->view_stride_px = (int64_t)h.px;
    out->render_ms = 0.f;
    out->copied_bytes = h.copied;
    (void)hipEventElapsedTime(&out->render_ms, h.t0, h.t1);
  }
  return NRF_OK;
}

int nrf_render_host_u8(nrf_context* c, int n_views, const float* cams, const float* poses, int flags, nrf_host_frame* out) {
  int ticket = -1;
  int rc = nrf_submit_host_u8(c, n_views, cams, poses, flags, &ticket);
  if (rc) return rc;
  return nrf_wait_host_u8(c, ticket, out);
}

int nrf_render_batch(nrf_context* c, int n_views, const float* cams, const float* poses, void* stream, nrf_frame* out) {
  return nrf_render_views(c, n_views, cams, poses, stream, out);
}

int nrf_render(nrf_context* c, const float cam[4], const float pose[16], void* stream, nrf_frame* out) {
  if (!cam || !pose) return fail(NRF_E_INVALID, "null argument");
  return nrf_render_views(c, 1, cam, pose, stream, out);
}

// Diagnostic (not part of include/nerfhip.h): raw counters of the last render; slots 2..6 are
// only filled by the NRF_PHASE_TIMING build (make prof).
int nrf_debug_counters(nrf_context* c, unsigned long long out[16]) {
  if (!c || !out) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(hipEventSynchronize(c->ev1));
  unsigned long long raw[COUNTER_SLOTS * 16];
  HIP_TRY(hipMemcpy(raw, call_slot(c, c->call_index), COUNTER_BYTES, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) {
    out[i] = 0;
    for (int sl = 0; sl < COUNTER_SLOTS; ++sl) out[i] = i == 14 ? std::max(out[i], raw[sl * 16 + i]) : out[i] + raw[sl * 16 + i];  // 14: a maximum
  }
  return NRF_OK;
}

// Diagnostic (not part of include/nerfhip.h): which kernel instance renders the loaded model -- 0 register-resident,
// 1 generic, 2 wide, 3 the register-resident instance of another width (16 / 32 / 128 neurons) or depth, 4 its wide form for SH
// degree 5..8, 5 a GRID instance (base.json's MLPs behind another grid: F = 2 with fewer than 16 levels, F = 4 / 8, Smoothstep);
// + 16 when the persistent form is used (tests assert that a model runs where it is meant to).
extern "C" int nrf_debug_instance(nrf_context* c) {
  if (!c || !c->model_loaded) return -1;
  return (c->dm.wide_sh ? 4 : (c->dm.hot_grid ? 5 : (c->dm.hot_width ? 3 : (c->dm.generic ? 1 : (c->dm.wide ? 2 : 0))))) + (c->dm.persistent ? 16 : 0);
}

// Diagnostic build: entry / exit stamps (s_memtime) of the persistent kernel's waves, 2 x n values.
extern "C" int nrf_debug_wave_times(nrf_context* c, unsigned long long* out, int n_waves) {
  if (!c || !out || n_waves < 1 || n_waves > 4096) return fail(NRF_E_INVALID, "bad argument");
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipMemcpy(out, call_slot(c, c->call_index) + COUNTER_BYTES + 128, (size_t)n_waves * 16, hipMemcpyDeviceToHost));
  return NRF_OK;
}

int nrf_render_async(nrf_context* c, const float cam[4], const float pose[16], nrf_frame* out) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  return nrf_render(c, cam, pose, (void*)c->stream, out);
}

int nrf_sync(nrf_context* c) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  int rc = set_device(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->last_stream && c->last_stream != c->stream) HIP_TRY(hipStreamSynchronize(c->last_stream));
  return NRF_OK;
}

int nrf_bind_output(nrf_context* c, void* rgba, void* depth) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if ((rgba == nullptr) != (depth == nullptr)) return fail(NRF_E_INVALID, "bind both planes or neither");
  c->bound_rgba = rgba;
  c->bound_depth = depth;
  c->bound_rgbd8 = nullptr;
  c->bound_rgb8 = c->bound_depth8 = nullptr;
  return NRF_OK;
}

int nrf_bind_output_rgbd8(nrf_context* c, void* rgbd8) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  c->bound_rgbd8 = rgbd8;
  if (rgbd8) c->bound_rgba = c->bound_depth = c->bound_rgb8 = c->bound_depth8 = nullptr;
  return NRF_OK;
}

int nrf_bind_output_u8(nrf_context* c, void* rgb8, void* depth8) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if ((rgb8 == nullptr) != (depth8 == nullptr)) return fail(NRF_E_INVALID, "bind both planes or neither");
  if (((uintptr_t)rgb8 | (uintptr_t)depth8) & 3u) return fail(NRF_E_INVALID, "8-bit planes must be 4-byte aligned");
  c->bound_rgb8 = rgb8;
  c->bound_depth8 = depth8;
  if (rgb8) c->bound_rgba = c->bound_depth = c->bound_rgbd8 = nullptr;
  return NRF_OK;
}

int nrf_get_stats(nrf_context* c, nrf_stats* s) {
  if (!c || !s) return fail(NRF_E_INVALID, "null argument");
  if (!c->rendered) return fail(NRF_E_STATE, "nothing rendered yet");
  int rc = set_device(c);
  if (rc) return rc;
  HIP_TRY(hipEventSynchronize(c->ev1));
  unsigned long long raw[COUNTER_SLOTS * 16], cnt[6] = {0, 0, 0, 0, 0, 0};
  HIP_TRY(hipMemcpy(raw, call_slot(c, c->call_index), COUNTER_BYTES, hipMemcpyDeviceToHost));
  for (int sl = 0; sl < COUNTER_SLOTS; ++sl) {
    cnt[0] += raw[sl * 16];
    cnt[1] += raw[sl * 16 + 1];
    cnt[2] += raw[sl * 16 + 11];
    cnt[3] += raw[sl * 16 + 7];
    cnt[4] += raw[sl * 16 + 12];
    cnt[5] += raw[sl * 16 + 13];
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  s->n_rays = (uint64_t)c->n_local_tiles * 64;
  s->n_samples = cnt[0];
  s->n_rounds = cnt[1];
  s->n_network_evals = cnt[2];
  s->n_composited = cnt[3];
  s->render_ms = ms;
  s->shader_clock_mhz = 0.f;
  s->gather_addresses_per_sample = c->gather_addresses;
  s->grid_device_bytes = c->table_bytes;
#ifndef NRF_PHASE_TIMING  // (a diagnostic build keeps other quantities in these two counters)
  if (cnt[5] > 0) s->shader_clock_mhz = (float)((double)cnt[4] / (double)cnt[5] * 100.0);
#endif
  return NRF_OK;
}

int nrf_read_view_f32(nrf_context* c, int view, float* rgba, float* depth) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if (!c->rendered) return fail(NRF_E_STATE, "nothing rendered yet");
  if (!c->last_rgba) return fail(NRF_E_STATE, "the last render went to a packed 8-bit buffer (nrf_bind_output_rgbd8): it is the caller's to read");
  if (tiled_layout(c)) return fail(NRF_E_STATE, "nrf_read_f32 needs a single-shard (row-major) frame");
  if (view < 0 || view >= c->last_views) return fail(NRF_E_INVALID, "view index out of range");
  int rc = set_device(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  const size_t n = (size_t)c->W * c->H;
  if (rgba) HIP_TRY(hipMemcpy(rgba, (const char*)c->last_rgba + (size_t)view * c->n_out_px * 16, n * 16, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(hipMemcpy(depth, (const char*)c->last_depth + (size_t)view * c->n_out_px * 4, n * 4, hipMemcpyDeviceToHost));
  return NRF_OK;
}

int nrf_read_f32(nrf_context* c, float* rgba, float* depth) { return nrf_read_view_f32(c, 0, rgba, depth); }

int nrf_read_shard_f32(nrf_context* c, float* rgba, float* depth) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if (!c->rendered) return fail(NRF_E_STATE, "nothing rendered yet");
  if (!c->last_rgba) return fail(NRF_E_STATE, "the last render went to a packed 8-bit buffer (nrf_bind_output_rgbd8): it is the caller's to read");
  int rc = set_device(c);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->last_stream));
  const size_t n = tiled_layout(c) ? (size_t)c->n_local_tiles * 64 : (size_t)c->W * c->H;
  if (rgba) HIP_TRY(hipMemcpy(rgba, c->last_rgba, n * 16, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(hipMemcpy(depth, c->last_depth, n * 4, hipMemcpyDeviceToHost));
  return NRF_OK;
}

int nrf_read_view_u8(nrf_context* c, int view, uint8_t* rgb, uint8_t* depth) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  if (!c->rendered) return fail(NRF_E_STATE, "nothing rendered yet");
  if (!c->last_rgba) return fail(NRF_E_STATE, "the last render went to a packed 8-bit buffer (nrf_bind_output_rgbd8): it is the caller's to read");
  if (tiled_layout(c)) return fail(NRF_E_STATE, "nrf_read_u8 needs a single-shard (row-major) frame");
  if (view < 0 || view >= c->last_views) return fail(NRF_E_INVALID, "view index out of range");
  int rc = set_device(c);
  if (rc) return rc;
  const size_t n = (size_t)c->W * c->H;
  if (!c->d_rgb8) {
    HIP_TRY(hipMalloc(&c->d_rgb8, n * 3));
    HIP_TRY(hipMalloc(&c->d_depth8, n));
  }
  hipStream_t st = c->last_stream;
  HIP_TRY(launch_quantize((const char*)c->last_rgba + (size_t)view * c->n_out_px * 16,
                          (const char*)c->last_depth + (size_t)view * c->n_out_px * 4, (int)n, c->d_rgb8, c->d_depth8, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (rgb) HIP_TRY(hipMemcpy(rgb, c->d_rgb8, n * 3, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(hipMemcpy(depth, c->d_depth8, n, hipMemcpyDeviceToHost));
  return NRF_OK;
}

int nrf_read_u8(nrf_context* c, uint8_t* rgb, uint8_t* depth) { return nrf_read_view_u8(c, 0, rgb, depth); }

int nrf_quantize_rgbd8(nrf_context* c, const void* rgba, const void* depth, uint64_t n_px, void* out_u32, void* stream) {
  if (!c || !rgba || !depth || !out_u32) return fail(NRF_E_INVALID, "null argument");
  int rc = set_device(c);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIP_TRY(launch_quantize_rgbd8(rgba, depth, n_px, out_u32, st));
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_quantize_u8(nrf_context* c, const void* rgba, const void* depth, uint64_t n_px, void* rgb8, void* depth8, void* stream) {
  if (!c || !rgba || !depth || !rgb8 || !depth8) return fail(NRF_E_INVALID, "null argument");
  if (n_px >= (1ull << 31)) return fail(NRF_E_INVALID, "n_px must be below 2^31");
  int rc = set_device(c);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIP_TRY(launch_quantize(rgba, depth, (int)n_px, rgb8, depth8, st));
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_untile_views_u8(nrf_context* c, const void* gathered_rgbd8, int shard_count, int tiles_per_shard, int n_views, void* rgb8,
                        void* depth8, void* stream) {
  if (!c || !gathered_rgbd8 || !rgb8 || !depth8 || shard_count < 1 || tiles_per_shard < 1 || n_views < 1)
    return fail(NRF_E_INVALID, "bad argument");
  if (c->W <= 0) return fail(NRF_E_STATE, "set_resolution has not been called");
  if (((uintptr_t)gathered_rgbd8 & 15u) || ((c->W & 3) == 0 && (((uintptr_t)rgb8 | (uintptr_t)depth8) & 3u)))
    return fail(NRF_E_INVALID, "the gathered shards must be 16-byte aligned, the 8-bit planes 4-byte aligned (widths that are multiples of 4)");
  int rc = set_device(c);
  if (rc) return rc;
  int tps = 0;
  nrf_tiles_per_shard(c->W, c->H, shard_count, &tps);
  if (tps != tiles_per_shard) return fail(NRF_E_INVALID, "tiles_per_shard does not match the resolution");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIP_TRY(launch_untile_rgbd8_u8(gathered_rgbd8, shard_count, tiles_per_shard, c->W, c->H, n_views, rgb8, depth8, st));
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_untile_views(nrf_context* c, const void* gathered, int shard_count, int tiles_per_shard, int channels, int n_views,
                     void* out, void* stream) {
  if (!c || !gathered || !out || shard_count < 1 || tiles_per_shard < 1 || channels < 1 || n_views < 1)
    return fail(NRF_E_INVALID, "bad argument");
  if (c->W <= 0) return fail(NRF_E_STATE, "set_resolution has not been called");
  int rc = set_device(c);
  if (rc) return rc;
  int tps = 0;
  nrf_tiles_per_shard(c->W, c->H, shard_count, &tps);
  if (tps != tiles_per_shard) return fail(NRF_E_INVALID, "tiles_per_shard does not match the resolution");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  HIP_TRY(launch_untile(gathered, shard_count, tiles_per_shard, channels, c->W, c->H, n_views, out, st));
  if (!stream) HIP_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_untile(nrf_context* c, const void* gathered, int shard_count, int tiles_per_shard, int channels, void* out,
               void* stream) {
  return nrf_untile_views(c, gathered, shard_count, tiles_per_shard, channels, 1, out, stream);
}

// ---- stage entry points ----
#define STAGE_PROLOGUE()                                         \
  int rc = need_model(c);                                        \
  if (rc) return rc;                                             \
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;

#define STAGE_EPILOGUE() \
  if (!stream) HIP_TRY(hipStreamSynchronize(st)); \
  return NRF_OK;

int nrf_encode_grid(nrf_context* c, const void* pos01, uint32_t n, void* out, void* stream) {
  STAGE_PROLOGUE();
  if (n && (!pos01 || !out)) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(launch_encode_grid(c->dm, pos01, n, out, st, c->opt.fast_interp != 0));
  STAGE_EPILOGUE();
}

// A wide model's rows (direction encodings of 32..80 values) go through the generic stage kernels: same arithmetic,
// generic-layout fragments; the fused kernel and nrf_network run the wide instance itself.
static DevModel stage_model(const nrf_context* c) {
  DevModel m = c->dm;
  if (m.wide) {
    m.generic = 1u;
    m.wfrag = (const uint4*)c->d_wfrag_gen;
  }
  return m;
}

int nrf_encode_dir(nrf_context* c, const void* dir01, uint32_t n, void* out, void* stream) {
  STAGE_PROLOGUE();
  if (n && (!dir01 || !out)) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(launch_encode_dir(stage_model(c), dir01, n, out, st));
  STAGE_EPILOGUE();
}

int nrf_mlp_forward_repeat(nrf_context* c, const void* feat, const void* dirfeat, uint32_t n, void* out, uint32_t repeat,
                           void* stream) {
  STAGE_PROLOGUE();
  if (n && (!feat || !dirfeat || !out)) return fail(NRF_E_INVALID, "null argument");
  if (repeat < 1) return fail(NRF_E_INVALID, "repeat must be >= 1");
  HIP_TRY(launch_mlp_forward(stage_model(c), feat, dirfeat, n, out, repeat, st));
  STAGE_EPILOGUE();
}

int nrf_mlp_forward(nrf_context* c, const void* feat, const void* dirfeat, uint32_t n, void* out, void* stream) {
  return nrf_mlp_forward_repeat(c, feat, dirfeat, n, out, 1, stream);
}

int nrf_network(nrf_context* c, const void* xyz, const void* dir, uint32_t n, void* sigma, void* rgb, void* stream) {
  STAGE_PROLOGUE();
  if (n && (!xyz || !dir || !sigma || !rgb)) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(launch_network(c->dm, xyz, dir, n, sigma, rgb, st));
  STAGE_EPILOGUE();
}

int nrf_generate_rays(nrf_context* c, const float cam[4], const float pose[16], void* rays_o, void* rays_d, void* nears,
                      void* fars, void* stream) {
  STAGE_PROLOGUE();
  if (!cam || !pose) return fail(NRF_E_INVALID, "null argument");
  if (c->W <= 0) return fail(NRF_E_STATE, "set_resolution has not been called");
  FrameParams P;
  fill_frame_params(c, cam, pose, P);
  HIP_TRY(launch_generate_rays(c->dm, P, rays_o, rays_d, nears, fars, st));
  STAGE_EPILOGUE();
}

int nrf_generate_rays_host(nrf_context* c, const float cam[4], const float pose[16], float* rays_o, float* rays_d,
                           float* nears, float* fars) {
  int rc = need_model(c);
  if (rc) return rc;
  if (c->W <= 0) return fail(NRF_E_STATE, "set_resolution has not been called");
  const size_t n = (size_t)c->W * c->H;
  void* buf = nullptr;
  HIP_TRY(hipMalloc(&buf, n * 8 * sizeof(float)));
  float* d_o = (float*)buf;
  float* d_d = d_o + 3 * n;
  float* d_n = d_d + 3 * n;
  float* d_f = d_n + n;
  rc = nrf_generate_rays(c, cam, pose, d_o, d_d, d_n, d_f, nullptr);
  hipError_t e = hipSuccess;
  if (!rc && rays_o) e = hipMemcpy(rays_o, d_o, n * 12, hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && rays_d) e = hipMemcpy(rays_d, d_d, n * 12, hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && nears) e = hipMemcpy(nears, d_n, n * 4, hipMemcpyDeviceToHost);
  if (!rc && e == hipSuccess && fars) e = hipMemcpy(fars, d_f, n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy");
  return NRF_OK;
}

int nrf_march(nrf_context* c, const void* rays_o, const void* rays_d, const void* rays_t, const void* fars, uint32_t n,
              uint32_t n_step, void* xyzs, void* dirs, void* deltas, void* stream) {
  STAGE_PROLOGUE();
  if (n_step < 1 || n_step > 8) return fail(NRF_E_INVALID, "n_step must be 1..8");
  if (n && (!rays_o || !rays_d || !rays_t || !fars || !xyzs || !dirs || !deltas)) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(launch_march(c->dm, c->opt.dt_gamma, rays_o, rays_d, rays_t, fars, n, n_step, xyzs, dirs, deltas, st, (uint32_t)c->opt.perturb));
  STAGE_EPILOGUE();
}

int nrf_composite(nrf_context* c, const void* sigmas, const void* rgbs, const void* deltas, uint32_t n, uint32_t n_step,
                  void* rays_t, void* state, void* stream) {
  if (!c) return fail(NRF_E_INVALID, "null context");
  int rc = set_device(c);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (n_step < 1 || n_step > 8) return fail(NRF_E_INVALID, "n_step must be 1..8");
  if (n && (!sigmas || !rgbs || !deltas || !rays_t || !state)) return fail(NRF_E_INVALID, "null argument");
  HIP_TRY(launch_composite(sigmas, rgbs, deltas, n, n_step, rays_t, state, st));
  STAGE_EPILOGUE();
}

}  // extern "C"
