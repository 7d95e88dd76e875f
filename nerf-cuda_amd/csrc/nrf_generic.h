// nrf_generic.h -- the GENERIC network instance of the gfx950 render path.
//
// The register-resident instance (nrf_device.h: level_gather / mlp_tiles) is specialised for the shape of the
// reference's base.json.  Everything else the reference's JSON vocabulary can describe runs through the functions
// below, with every model parameter a runtime value:
//   * hash grid: n_levels <= 16, n_features_per_level in {1, 2, 4, 8}, Hash / Dense / Tiled tables of any size,
//     Linear / Nearest / Smoothstep interpolation        (T/include/tiny-cuda-nn/encodings/grid.h:100-117,139-268,1365-1411)
//   * direction encoding: SphericalHarmonics degree 1..8, Frequency with any n_frequencies (padded width <= 112),
//     Identity                                           (T/.../spherical_harmonics.h:46-152, frequency.h:46-93, identity.h:46-67)
//   * FullyFusedMLP of width 16 / 32 / 64 / 128 with any number of hidden layers and any tcnn activation
//                                                        (T/src/fully_fused_mlp.cu:500-558,636-725; T/.../common_device.h:68-114)
// The arithmetic is the hot instance's (and the oracle's): fp16 storage, fp16 interpolation accumulator in corner
// order, fp32 MFMA accumulation, activation on the fp32 sum, fp16 between layers.
//
// Data flow of one pass over <= 32 queued samples of a wave (two 16-sample MFMA tiles):
//   features -> LDS rows X[sample][k] (fp16) -> for every layer: Y[sample][n] = act(W X) on v_mfma_f32_16x16x32_f16
//   with A = weight fragments streamed from global memory / L2 (they are the same for every wave of the chip) and
//   B = one ds_read_b128 per lane from the activation rows; X and Y swap roles from layer to layer.  The first rgb
//   layer reads its input columns from two places: columns 0..15 = the density MLP's output, the rest = the
//   sample's direction encoding, evaluated per pass from the ray's direction into 32 LDS rows (a per-ray buffer
//   of the widest encodings -- 64 rays x 88 halves -- would cost a workgroup per CU).
#pragma once

#include "nrf_device.h"

namespace nrf {

constexpr int GEN_MAX_LAYERS = 24;        // density + rgb matmuls
constexpr int GEN_TILES = 2;              // 16-sample tiles per pass
constexpr int GEN_SAMPLES = 16 * GEN_TILES;
constexpr int GEN_MAX_DIR_W = 112;        // padded direction-encoding width (rgb input <= 128)

struct GenLayer {
  uint32_t frag_off;  // first fragment of the layer: fragment (m, s) = frag_off + m * k_steps + s, 64 uint4 each
  uint32_t k_steps;   // ceil(K / 32)
  uint32_t n_tiles;   // N / 16
  uint32_t act;       // activation of this layer's output (NRF_ACT_*)
};

// Lives in device memory (DevModel::gen); all fields are wave-uniform, so the kernels read them with scalar loads.
struct GenModel {
  uint32_t F;          // features per level
  uint32_t interp;     // NRF_INTERP_*
  uint32_t n_levels;
  uint32_t feat_raw;   // n_levels * F
  uint32_t feat_w;     // feat_raw padded to 16: rows of nrf_encode_grid, n_input_dims of the density MLP
  uint32_t feat_k;     // feat_w padded to 32 (the MFMA K step); the extra columns are zero
  uint32_t width;      // n_neurons
  uint32_t dir_raw, dir_w;  // direction encoding: raw and padded (16) width
  uint32_t rgb_in;     // 16 + dir_w
  uint32_t n_dens, n_rgb;   // matmuls of the density / rgb MLP (hidden layers + 1)
  uint32_t act_stride; // halves per row of the activation buffers X, Y
  uint32_t dir_stride; // halves per row of the per-pass direction rows
  uint32_t fast_grid;  // 1: F == 2, Linear or Smoothstep interpolation, every level dense / power-of-two hashed / LV_ADD_POW2 -- the feature rows come
                       // from the register-resident instance's level_gather / level_interp (gen_encode_rows)
  uint32_t pad1;
  GenLayer layer[GEN_MAX_LAYERS];  // density layers, then rgb layers
};

// LDS bytes per wave of the two generic regions (see render_kernel's LDS map)
__host__ __device__ inline uint32_t gen_dir_bytes(const GenModel& G) { return (uint32_t)GEN_SAMPLES * G.dir_stride * 2u; }
__host__ __device__ inline uint32_t gen_act_bytes(const GenModel& G) { return 2u * GEN_SAMPLES * G.act_stride * 2u; }

struct GenLds {
  half_t* dens;  // [GEN_SAMPLES][16]: output of the density MLP = columns 0..15 of the rgb MLP's input
  half_t* dir;   // [GEN_SAMPLES][dir_stride]: direction encoding of the pass's samples
  float* rayd;   // [64 rays][3]: 0.5 d + 0.5 of every ray (nerf_render.cu:313-314)
  half_t* X;     // [GEN_SAMPLES][act_stride]
  half_t* Y;
  const uint4* wfrag;  // the layers' weight fragments: DevModel::wfrag (global memory) or the persistent kernel's LDS copy
};

// ---------------------------------------------------------------- hash grid ----
// grid_index, T/.../grid.h:100-117, on the device copy of the table (level offsets L.offset are the device ones)
__device__ __forceinline__ uint32_t gen_grid_index(const LevelParams& L, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t stride = 1, index = 0;
  if (stride <= L.size) { index += x * stride; stride *= L.res; }
  if (stride <= L.size) { index += y * stride; stride *= L.res; }
  if (stride <= L.size) { index += z * stride; stride *= L.res; }
  if (L.hashed && L.size < stride) index = x ^ (y * 2654435761u) ^ (z * 805459861u);
  return ((L.size & (L.size - 1u)) == 0u) ? (index & (L.size - 1u)) : (index % L.size);
}

template <int F>
__device__ __forceinline__ void gen_entry_load(const half_t* __restrict__ grid, const LevelParams& L, uint32_t entry, half_t (&v)[F]) {
  const half_t* p = grid + ((size_t)L.offset + entry) * F;
  if constexpr (F == 1) {
    v[0] = p[0];
  } else if constexpr (F == 2) {
    const half2_t t = *reinterpret_cast<const half2_t*>(p);
    v[0] = t.x; v[1] = t.y;
  } else if constexpr (F == 4) {
    const half4_t t = *reinterpret_cast<const half4_t*>(p);
#pragma unroll
    for (int f = 0; f < 4; ++f) v[f] = t[f];
  } else {
    const half8_t t = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
    for (int f = 0; f < 8; ++f) v[f] = t[f];
  }
}

// One (sample, level) of kernel_grid<half,3,F> (grid.h:186-267): F fp16 features.
template <int F>
__device__ __forceinline__ void gen_level(const half_t* __restrict__ grid, const LevelParams L, uint32_t interp, float px, float py,
                                          float pz, half_t (&res)[F]) {
  const float in[3] = {px, py, pz};
  float fr[3];
  uint32_t g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {  // pos_fract, common_device.h:414-422
    float v = in[d] * L.scale;
    v = v + 0.5f;
    const float fl = floorf(v);
    g[d] = (uint32_t)(int)fl;
    float f = v - fl;
    if (interp == NRF_INTERP_SMOOTHSTEP) {  // val*val*(3.0f - 2.0f*val), common_device.h:379-381
      const float sq = f * f;
      const float b = 2.0f * f;
      f = sq * (3.0f - b);
    }
    fr[d] = f;
  }
  if (interp == NRF_INTERP_NEAREST) {  // grid.h:215-232
    gen_entry_load<F>(grid, L, gen_grid_index(L, g[0], g[1], g[2]), res);
    return;
  }
  half_t v[8][F];
#pragma unroll
  for (int c = 0; c < 8; ++c)  // all eight gathers in flight before the first is used
    gen_entry_load<F>(grid, L, gen_grid_index(L, g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1)), v[c]);
#pragma unroll
  for (int f = 0; f < F; ++f) res[f] = (half_t)0.0f;
  const float wx[2] = {1 - fr[0], fr[0]}, wy[2] = {1 - fr[1], fr[1]}, wz[2] = {1 - fr[2], fr[2]};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];  // ((1 * wx) * wy) * wz, grid.h:240-252
#pragma unroll
    for (int f = 0; f < F; ++f) res[f] = res[f] + f2h_rne(w * (float)v[c][f]);  // result += (T)(weight * data), grid.h:260
  }
}

// --------------------------------------------------------- direction encoding ----
// dir_w fp16 values of one direction into `row` (LDS or global).  d01 = 0.5 d + 0.5 (nerf_render.cu:313-314).
__device__ __forceinline__ void gen_encode_dir(const DevModel& M, const GenModel& G, float d01x, float d01y, float d01z, half_t* row) {
  const uint32_t raw = G.dir_raw, width = G.dir_w;
  if (M.dir_encoding == NRF_DIR_SH) {
    const uint32_t pad = width - raw;  // SH pads with LEADING ones (spherical_harmonics.h:57-64)
    for (uint32_t j = 0; j < pad; ++j) row[j] = (half_t)1.0f;
    half_t* o = row + pad;
    const uint32_t degree = M.sh_degree;
    const float x = d01x * 2.f - 1.f, y = d01y * 2.f - 1.f, z = d01z * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
    // the polynomial table of kernel_sh (spherical_harmonics.h:78-152), C++ operator order, one rounding per operation
#define NRF_SH(k, expr) o[k] = f2h_rne(expr)
    NRF_SH(0, 0.28209479177387814f);
    if (degree <= 1) return;
    NRF_SH(1, -0.48860251190291987f * y);
    NRF_SH(2, 0.48860251190291987f * z);
    NRF_SH(3, -0.48860251190291987f * x);
    if (degree <= 2) return;
    NRF_SH(4, 1.0925484305920792f * xy);
    NRF_SH(5, -1.0925484305920792f * yz);
    NRF_SH(6, 0.94617469575755997f * z2 - 0.31539156525251999f);
    NRF_SH(7, -1.0925484305920792f * xz);
    NRF_SH(8, 0.54627421529603959f * x2 - 0.54627421529603959f * y2);
    if (degree <= 3) return;
    NRF_SH(9, 0.59004358992664352f * y * (-3.0f * x2 + y2));
    NRF_SH(10, 2.8906114426405538f * xy * z);
    NRF_SH(11, 0.45704579946446572f * y * (1.0f - 5.0f * z2));
    NRF_SH(12, 0.3731763325901154f * z * (5.0f * z2 - 3.0f));
    NRF_SH(13, 0.45704579946446572f * x * (1.0f - 5.0f * z2));
    NRF_SH(14, 1.4453057213202769f * z * (x2 - y2));
    NRF_SH(15, 0.59004358992664352f * x * (-x2 + 3.0f * y2));
    if (degree <= 4) return;
    NRF_SH(16, 2.5033429417967046f * xy * (x2 - y2));
    NRF_SH(17, 1.7701307697799304f * yz * (-3.0f * x2 + y2));
    NRF_SH(18, 0.94617469575756008f * xy * (7.0f * z2 - 1.0f));
    NRF_SH(19, 0.66904654355728921f * yz * (3.0f - 7.0f * z2));
    NRF_SH(20, -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f);
    NRF_SH(21, 0.66904654355728921f * xz * (3.0f - 7.0f * z2));
    NRF_SH(22, 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f));
    NRF_SH(23, 1.7701307697799304f * xz * (-x2 + 3.0f * y2));
    NRF_SH(24, -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4);
    if (degree <= 5) return;
    NRF_SH(25, 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4));
    NRF_SH(26, 8.3026492595241645f * xy * z * (x2 - y2));
    NRF_SH(27, -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f));
    NRF_SH(28, 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f));
    NRF_SH(29, 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f));
    NRF_SH(30, 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f));
    NRF_SH(31, 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f));
    NRF_SH(32, 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f));
    NRF_SH(33, -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f));
    NRF_SH(34, 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4));
    NRF_SH(35, 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4));
    if (degree <= 6) return;
    NRF_SH(36, 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4));
    NRF_SH(37, 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4));
    NRF_SH(38, 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f));
    NRF_SH(39, -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f));
    NRF_SH(40, 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f));
    NRF_SH(41, 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f));
    NRF_SH(42, 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f);
    NRF_SH(43, 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f));
    NRF_SH(44, 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f));
    NRF_SH(45, -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f));
    NRF_SH(46, 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4));
    NRF_SH(47, 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4));
    NRF_SH(48, 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6);
    if (degree <= 7) return;
    NRF_SH(49, 0.70716273252459627f * y * (-21.0f * x2 * y4 + 35.0f * x4 * y2 - 7.0f * x6 + y6));
    NRF_SH(50, 5.2919213236038001f * xy * z * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4));
    NRF_SH(51, -0.51891557872026028f * y * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + 5.0f * x4 + y4));
    NRF_SH(52, 4.1513246297620823f * xy * z * (x2 - y2) * (13.0f * z2 - 3.0f));
    NRF_SH(53, -0.15645893386229404f * y * (3.0f * x2 - y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f));
    NRF_SH(54, 0.44253269244498261f * xy * z * (-110.0f * z2 + 143.0f * z4 + 15.0f));
    NRF_SH(55, 0.090331607582517306f * y * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f));
    NRF_SH(56, 0.068284276912004949f * z * (315.0f * z2 - 693.0f * z4 + 429.0f * z6 - 35.0f));
    NRF_SH(57, 0.090331607582517306f * x * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f));
    NRF_SH(58, 0.07375544874083044f * z * (x2 - y2) * (143.0f * z2 * (3.0f * z2 - 1.0f) - 187.0f * z2 + 45.0f));
    NRF_SH(59, -0.15645893386229404f * x * (x2 - 3.0f * y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f));
    NRF_SH(60, 1.0378311574405206f * z * (13.0f * z2 - 3.0f) * (-6.0f * x2 * y2 + x4 + y4));
    NRF_SH(61, -0.51891557872026028f * x * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + x4 + 5.0f * y4));
    NRF_SH(62, 2.6459606618019f * z * (15.0f * x2 * y4 - 15.0f * x4 * y2 + x6 - y6));
    NRF_SH(63, 0.70716273252459627f * x * (-35.0f * x2 * y4 + 21.0f * x4 * y2 - x6 + 7.0f * y6));
#undef NRF_SH
  } else if (M.dir_encoding == NRF_DIR_FREQUENCY) {  // frequency.h:72-89
    const float PI = 3.14159265358979323846f;
    const uint32_t nf = M.n_frequencies;
    const float in[3] = {d01x, d01y, d01z};
    uint32_t j = 0;
#pragma unroll
    for (int feat = 0; feat < 3; ++feat)                     // j / (2 nf)
      for (uint32_t k = 0; k < nf; ++k) {                    // (j / 2) % nf
        const float xs = ldexpf(in[feat], (int)k) * PI;      // scalbnf(x, log2_frequency) * PI
        row[j++] = f2h_rne(__sinf(xs + 0.0f));               // phase (j % 2) * (PI / 2)
        row[j++] = f2h_rne(__sinf(xs + PI / 2));
      }
    for (; j < width; ++j) row[j] = (half_t)1.0f;            // trailing ones
  } else {  // Identity: scale 1, offset 0, trailing ones (identity.h:60-66)
    row[0] = f2h_rne(d01x);
    row[1] = f2h_rne(d01y);
    row[2] = f2h_rne(d01z);
    for (uint32_t j = 3; j < width; ++j) row[j] = (half_t)1.0f;
  }
}

// ------------------------------------------------------------------- layers ----
__device__ __forceinline__ half8_t gen_frag(const uint4* __restrict__ frags, uint32_t f, int lane) {
  const uint4 v = frags[(size_t)f * 64 + lane];
  return __builtin_bit_cast(half8_t, v);
}

__device__ __forceinline__ uint2 gen_pack4(uint32_t act, float4_t a) {
  return make_uint2(pack_h2(activate(act, a[0]), activate(act, a[1])), pack_h2(activate(act, a[2]), activate(act, a[3])));
}

// One layer for GEN_TILES tiles of 16 samples: out[sample][n] = act(sum_k W[n][k] in[sample][k]).
// bfetch(tile, s) returns lane (g, c)'s B fragment = input columns 32 s + 8 g .. + 7 of sample 16 tile + c.
// Hidden layers (out != nullptr) store fp16 rows; the caller synchronises the wave before reading them.
// Output layers (16 rows) leave lane (g, c) with rows 4g .. 4g + 3 of sample c in `last`.
template <typename BFetch>
__device__ __forceinline__ void gen_layer(const GenLayer ly, const uint4* __restrict__ frags, int lane, BFetch bfetch, half_t* out,
                                          uint32_t out_stride, float4_t (&last)[GEN_TILES]) {
  const int g = lane >> 4, c = lane & 15;
  const float4_t zero = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t m = 0; m < ly.n_tiles; ++m) {
    float4_t acc[GEN_TILES];
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n) acc[n] = zero;
    for (uint32_t s = 0; s < ly.k_steps; ++s) {
      const half8_t a = gen_frag(frags, ly.frag_off + m * ly.k_steps + s, lane);
#pragma unroll
      for (int n = 0; n < GEN_TILES; ++n) acc[n] = mfma16(a, bfetch(n, s), acc[n]);
    }
    if (out) {
#pragma unroll
      for (int n = 0; n < GEN_TILES; ++n)
        *reinterpret_cast<uint2*>(out + (size_t)(16 * n + c) * out_stride + 16 * m + 4 * g) = gen_pack4(ly.act, acc[n]);
    } else {
#pragma unroll
      for (int n = 0; n < GEN_TILES; ++n) last[n] = acc[n];
    }
  }
  if (out && (ly.n_tiles & 1u)) {  // a 16-wide layer feeds a 32-wide K step: the upper half of that step is zero
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n)
      *reinterpret_cast<uint2*>(out + (size_t)(16 * n + c) * out_stride + 16 * ly.n_tiles + 4 * g) = make_uint2(0u, 0u);
  }
}

__device__ __forceinline__ void gen_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Both MLPs (nerf_network.h:148-196) on the rows X[0..31][0..feat_k) of the wave; ray[n] = row of the direction
// rows that belongs to sample 16 n + c.  Results: lanes g == 0 hold (r, g, b, sigma) of sample c of tile n.
// DENSITY_ONLY: stop after the density MLP (sigma only; density-grid generation).
template <bool DENSITY_ONLY>
__device__ __forceinline__ void gen_mlps(const DevModel& M, const GenModel& G, const GenLds& Lw, int lane, const int (&ray)[GEN_TILES],
                                         float4_t (&out)[GEN_TILES]) {
  const int g = lane >> 4, c = lane & 15;
  const uint4* __restrict__ frags = Lw.wfrag;
  const uint32_t stride = G.act_stride;
  half_t* cur = Lw.X;
  half_t* nxt = Lw.Y;
  float4_t last[GEN_TILES];
  // ---- density MLP
  for (uint32_t i = 0; i < G.n_dens; ++i) {
    const GenLayer ly = G.layer[i];
    const bool is_last = i + 1 == G.n_dens;
    const half_t* in = cur;
    gen_layer(ly, frags, lane,
              [&](int n, uint32_t s) { return *reinterpret_cast<const half8_t*>(in + (size_t)(16 * n + c) * stride + 32 * s + 8 * g); },
              is_last ? nullptr : nxt, stride, last);
    gen_wave_sync();
    half_t* t = cur; cur = nxt; nxt = t;
  }
  float sig_pre[GEN_TILES];
#pragma unroll
  for (int n = 0; n < GEN_TILES; ++n) {
    // density output rows 4g .. 4g+3 (fp16) = columns 0..15 of the rgb MLP's input (nerf_network.h:162-164)
    const uint2 r = gen_pack4(M.density_output_activation, last[n]);
    *reinterpret_cast<uint2*>(Lw.dens + (16 * n + c) * 16 + 4 * g) = r;
    sig_pre[n] = (float)bits_h2(r.x).x;  // row 0 lives in lanes g == 0
  }
  if constexpr (!DENSITY_ONLY) {
    gen_wave_sync();
    // ---- rgb MLP; its first layer gathers [density out | direction encoding | zero padding]
    cur = Lw.X;
    nxt = Lw.Y;
    for (uint32_t i = 0; i < G.n_rgb; ++i) {
      const GenLayer ly = G.layer[G.n_dens + i];
      const bool is_last = i + 1 == G.n_rgb;
      if (i == 0) {
        gen_layer(ly, frags, lane,
                  [&](int n, uint32_t s) {
                    const uint32_t col = 32u * s + 8u * (uint32_t)g;
                    const half8_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                    if (col < 16u) return *reinterpret_cast<const half8_t*>(Lw.dens + (16 * n + c) * 16 + col);
                    if (col < G.rgb_in) return *reinterpret_cast<const half8_t*>(Lw.dir + (size_t)ray[n] * G.dir_stride + (col - 16u));
                    return z;
                  },
                  is_last ? nullptr : cur, stride, last);  // X's feature rows are dead by now
      } else {
        const half_t* in = cur;
        gen_layer(ly, frags, lane,
                  [&](int n, uint32_t s) { return *reinterpret_cast<const half8_t*>(in + (size_t)(16 * n + c) * stride + 32 * s + 8 * g); },
                  is_last ? nullptr : nxt, stride, last);
        half_t* t = cur; cur = nxt; nxt = t;
      }
      gen_wave_sync();
    }
  }
#pragma unroll
  for (int n = 0; n < GEN_TILES; ++n) {
    float s = sig_pre[n];
    switch (M.sigma_activation) {  // wrap_a_activation handles exactly these (nerf_network.h:32-47); fp32 math, fp16 store
      case NRF_ACT_RELU: s = s > 0.0f ? s : 0.0f; break;
      case NRF_ACT_EXPONENTIAL: s = expf(s); break;
      case NRF_ACT_SIGMOID: s = 1.0f / (1.0f + expf(-s)); break;
      default: break;
    }
    float4_t o = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!DENSITY_ONLY) {
      o[0] = (float)(half_t)activate(M.rgb_output_activation, last[n][0]);
      o[1] = (float)(half_t)activate(M.rgb_output_activation, last[n][1]);
      o[2] = (float)(half_t)activate(M.rgb_output_activation, last[n][2]);
    }
    o[3] = (float)(half_t)s;
    out[n] = o;
  }
}

// Feature rows of <= 32 samples: lane (g, c) encodes levels g, 4 + g, ... of sample 16 n + c into X, and the
// padding columns [feat_raw, feat_k) are zeroed (grid.h:959-969 pads with 0; the rest is the MFMA K padding).
__device__ __forceinline__ void gen_encode_rows(const DevModel& M, const GenModel& G, const LevelParams* lvs, const GenLds& Lw, int lane,
                                                const float (&p01)[GEN_TILES][3], const bool (&valid)[GEN_TILES]) {
  const int g = lane >> 4, c = lane & 15;
  const half_t* __restrict__ grid = reinterpret_cast<const half_t*>(M.grid);
  if (G.fast_grid == 4u || G.fast_grid == 8u) {
    // F = 4 / 8 on a standard grid: the GRID instances' aligned 8- / 16-byte gathers (level_gather_wide), one level of the lane in
    // flight at a time (two F = 4 levels at once made the 12-wave instance spill 40 VGPRs)
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n) {
      half_t* row = Lw.X + (size_t)(16 * n + c) * G.act_stride;
      for (uint32_t j = G.feat_raw + (uint32_t)g; j < G.feat_k; j += 4u) row[j] = (half_t)0.0f;
      if (!valid[n]) continue;
      if (G.fast_grid == 4u) {
        for (uint32_t lv = (uint32_t)g; lv < G.n_levels; lv += 4u) {
          uint32_t v[16], o[2];
          float fr[3];
          level_gather_wide<0, 2>(M.grid, M.grid_bytes, lvs[lv], p01[n][0], p01[n][1], p01[n][2], v, fr);
          if (G.interp == NRF_INTERP_SMOOTHSTEP) smoothstep_fractions(fr);
          level_interp_wide<2>(v, fr, o);
          *reinterpret_cast<uint2*>(row + 4u * lv) = make_uint2(o[0], o[1]);
        }
      } else {
        for (uint32_t lv = (uint32_t)g; lv < G.n_levels; lv += 4u) {
          uint32_t v[32], o[4];
          float fr[3];
          level_gather_wide<0, 4>(M.grid, M.grid_bytes, lvs[lv], p01[n][0], p01[n][1], p01[n][2], v, fr);
          if (G.interp == NRF_INTERP_SMOOTHSTEP) smoothstep_fractions(fr);
          level_interp_wide<4>(v, fr, o);
          *reinterpret_cast<uint4*>(row + 8u * lv) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    }
    return;
  }
  if (G.fast_grid != 0u) {
    // The grid is the register-resident instance's kind (only the networks behind it are not): its gathers -- byte-offset MUBUF
    // loads with the per-level constants of LevelParams, every load of the lane's (up to four) levels in flight before the
    // first is consumed -- instead of gen_level's literal index arithmetic, one level at a time.  Same values: both follow
    // grid.h:100-117 / :186-267 (tests/test_generic_gpu.py compares the rows with the oracle's bit for bit).
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n) {
      half_t* row = Lw.X + (size_t)(16 * n + c) * G.act_stride;
      for (uint32_t j = G.feat_raw + (uint32_t)g; j < G.feat_k; j += 4u) row[j] = (half_t)0.0f;
      if (!valid[n]) continue;
      uint32_t v[4][8];
      float fr[4][3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lv = (uint32_t)g + 4u * (uint32_t)j;
        if (lv < G.n_levels) {
          level_gather<0>(M.grid, M.grid_bytes, lvs[lv], p01[n][0], p01[n][1], p01[n][2], v[j], fr[j]);
          if (G.interp == NRF_INTERP_SMOOTHSTEP) {  // (wave-uniform) val*val*(3.0f - 2.0f*val) on the fractions, as gen_level does
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const float f = fr[j][d], sq = f * f, b = 2.0f * f;
              fr[j][d] = sq * (3.0f - b);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lv = (uint32_t)g + 4u * (uint32_t)j;
        if (lv < G.n_levels) *reinterpret_cast<uint32_t*>(row + 2u * lv) = level_interp<false>(v[j], fr[j]);
      }
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < GEN_TILES; ++n) {
    half_t* row = Lw.X + (size_t)(16 * n + c) * G.act_stride;
    for (uint32_t j = G.feat_raw + (uint32_t)g; j < G.feat_k; j += 4u) row[j] = (half_t)0.0f;
    if (!valid[n]) continue;
    for (uint32_t lv = (uint32_t)g; lv < G.n_levels; lv += 4u) {
      const LevelParams L = lvs[lv];
      switch (G.F) {
        case 1: {
          half_t r[1];
          gen_level<1>(grid, L, G.interp, p01[n][0], p01[n][1], p01[n][2], r);
          row[lv] = r[0];
        } break;
        case 2: {
          half_t r[2];
          gen_level<2>(grid, L, G.interp, p01[n][0], p01[n][1], p01[n][2], r);
          half2_t t; t.x = r[0]; t.y = r[1];
          *reinterpret_cast<half2_t*>(row + 2 * lv) = t;
        } break;
        case 4: {
          half_t r[4];
          gen_level<4>(grid, L, G.interp, p01[n][0], p01[n][1], p01[n][2], r);
          half4_t t;
#pragma unroll
          for (int f = 0; f < 4; ++f) t[f] = r[f];
          *reinterpret_cast<half4_t*>(row + 4 * lv) = t;
        } break;
        default: {
          half_t r[8];
          gen_level<8>(grid, L, G.interp, p01[n][0], p01[n][1], p01[n][2], r);
          half8_t t;
#pragma unroll
          for (int f = 0; f < 8; ++f) t[f] = r[f];
          *reinterpret_cast<half8_t*>(row + 8 * lv) = t;
        } break;
      }
    }
  }
}

}  // namespace nrf
