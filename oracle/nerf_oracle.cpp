// nerf_oracle.cpp -- CPU restatement of the reference hot path (see nerf_oracle.h).
// TEST INFRASTRUCTURE ONLY; parity unpinned (no reference fixtures exist).
//
// Every function cites the reference file:line it follows.  R/ = the
// reference repo, T/ = R/dependencies/tiny-cuda-nn.
// Build: oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).

#include "nerf_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __F16C__
#include <immintrin.h>  // vcvtps2ph / vcvtph2ps: the same RNE bits as the software conversions below (tests/test_oracle_kat.py)
#endif

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// ---------------------------------------------------------------- fp16 ----
// IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept.  The software forms are the definition (and what a
// build without F16C runs); with -mf16c (oracle/Makefile, when the build host has it) f2h / h2f are one instruction each --
// the CPU baseline then measures the path, not a software-float emulator.
inline uint16_t f2h_soft(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) {  // inf / nan
    return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? (0x0200u | ((x >> 13) & 0x3ffu)) : 0u));
  }
  if (x >= 0x477ff000u) {  // >= 65520 rounds to inf
    return (uint16_t)(sign | 0x7c00u);
  }
  if (x < 0x38800000u) {  // below the smallest normal half (2^-14): subnormal or zero
    if (x < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
    const int e = (int)(x >> 23);                // biased exponent, 102..112
    uint32_t mant = (x & 0x7fffffu) | 0x800000u; // 24-bit significand
    const int shift = 126 - e;                   // 14..24: value = mant * 2^(e-150); half sub unit 2^-24
    const uint32_t q = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  // normal
  uint32_t mant = x & 0x7fffffu;
  uint32_t e = (x >> 23) - 112u;  // half biased exponent
  uint32_t r = (e << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;  // carries into exponent correctly
  return (uint16_t)(sign | r);
}

inline float h2f_soft(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1fu;
  const uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {  // subnormal: m * 2^-24
      float f = (float)m * 5.9604644775390625e-08f;
      std::memcpy(&x, &f, 4);
      x |= sign;
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

#ifdef __F16C__
inline uint16_t f2h(float f) { return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
#else
inline uint16_t f2h(float f) { return f2h_soft(f); }
inline float h2f(uint16_t h) { return h2f_soft(h); }
#endif

// The contraction switch (nrfo_set_contract): `a * b + c` with every operation rounded (the arithmetic contract shared with
// the HIP path) or as ONE fused multiply-add, which is what nvcc makes of it by default (-fmad=true; R/CMakeLists.txt:71-79
// sets no -fmad=false).  Rule applied at every site below: a product that feeds an add / subtract of the same expression
// is fused; of two products in one sum the LEFT one is fused and the right one rounded first (LLVM's combine order:
// fadd(fmul(a, b), z) -> fma(a, b, z) is tried before fadd(z, fmul(a, b))); a product whose value is cast in between
// ((T)(weight * data), grid.h:260) is not.
inline float mad(int contract, float a, float b, float c) { return contract ? fmaf(a, b, c) : a * b + c; }
// `a * b + c * d`: which product is fused is the compiler's choice.  Mode 1: the left one (above); mode 2 (the sensitivity run of
// nrfo_set_contract): the right one.
inline float mad2(int contract, float a, float b, float c, float d) {
  return contract == 2 ? fmaf(c, d, a * b) : (contract ? fmaf(a, b, c * d) : a * b + c * d);
}
static thread_local bool tl_fuse_right = false;  // operator+(Prod, Prod) below: set by the caller of a formula table (mode 2)

// The same rule for whole formula tables (kernel_sh): a float that remembers being a product until it is used.  With
// S = float a table is plain individually-rounded arithmetic; with S = Fx every `a*b + c`, `a*b - c`, `c + a*b`, `c - a*b`
// becomes one fmaf.
struct Fx;
struct Prod {
  float a, b;
  float value() const { return a * b; }
};
struct Fx {
  float v;
  Fx() : v(0.0f) {}
  Fx(float f) : v(f) {}
  Fx(Prod p) : v(p.value()) {}
};
inline Prod operator*(Fx a, Fx b) { return {a.v, b.v}; }
inline Prod operator*(float a, Fx b) { return {a, b.v}; }
inline Prod operator*(Fx a, float b) { return {a.v, b}; }
inline Prod operator*(Prod a, Fx b) { return {a.value(), b.v}; }
inline Prod operator*(Prod a, float b) { return {a.value(), b}; }
inline Prod operator*(Fx a, Prod b) { return {a.v, b.value()}; }
inline Prod operator*(float a, Prod b) { return {a, b.value()}; }
inline Prod operator*(Prod a, Prod b) { return {a.value(), b.value()}; }
inline Fx operator+(Fx a, Fx b) { return a.v + b.v; }
inline Fx operator-(Fx a, Fx b) { return a.v - b.v; }
inline Fx operator+(float a, Fx b) { return a + b.v; }
inline Fx operator-(float a, Fx b) { return a - b.v; }
inline Fx operator+(Fx a, float b) { return a.v + b; }
inline Fx operator-(Fx a, float b) { return a.v - b; }
inline Fx operator-(Fx a) { return -a.v; }
inline Fx operator+(Prod p, Fx c) { return fmaf(p.a, p.b, c.v); }
inline Fx operator-(Prod p, Fx c) { return fmaf(p.a, p.b, -c.v); }
inline Fx operator+(Prod p, float c) { return fmaf(p.a, p.b, c); }
inline Fx operator-(Prod p, float c) { return fmaf(p.a, p.b, -c); }
inline Fx operator+(Fx c, Prod p) { return fmaf(p.a, p.b, c.v); }
inline Fx operator-(Fx c, Prod p) { return fmaf(-p.a, p.b, c.v); }
inline Fx operator+(float c, Prod p) { return fmaf(p.a, p.b, c); }
inline Fx operator-(float c, Prod p) { return fmaf(-p.a, p.b, c); }
inline Fx operator+(Prod p, Prod q) { return tl_fuse_right ? fmaf(q.a, q.b, p.value()) : fmaf(p.a, p.b, q.value()); }
inline Fx operator-(Prod p, Prod q) { return tl_fuse_right ? fmaf(-q.a, q.b, p.value()) : fmaf(p.a, p.b, -q.value()); }
struct ShOut {  // one coefficient of the table: takes whatever the formula's last operation produced
  float v;
  ShOut& operator=(float f) { v = f; return *this; }
  ShOut& operator=(Fx f) { v = f.v; return *this; }
  ShOut& operator=(Prod p) { v = p.value(); return *this; }
};

// fp16 addition: exact via fp32 (24 >= 2*11+2 bits, so double rounding is innocuous).
inline uint16_t hadd(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); }

// pcg32(initstate, initseq).next_float() (T/dependencies/pcg32/pcg32.h:48-62 seed, :65-71 next_uint, :108-117 next_float): the
// first float of the generator the march seeds per ray and call (render_utils.h:586-589).
inline float pcg32_first_float(uint64_t initstate, uint64_t initseq) {
  const uint64_t MULT = 0x5851f42d4c957f2dULL;  // PCG32_MULT
  const uint64_t inc = (initseq << 1u) | 1u;
  uint64_t state = 0U;
  state = state * MULT + inc;  // seed(): next_uint()
  state += initstate;
  state = state * MULT + inc;  // seed(): next_uint()
  const uint64_t oldstate = state;  // next_float(): next_uint()
  const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  const uint32_t rot = (uint32_t)(oldstate >> 59u);
  const uint32_t r = (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
  const uint32_t u = (r >> 9) | 0x3f800000u;
  float f;
  std::memcpy(&f, &u, 4);
  return f - 1.0f;
}

// ---------------------------------------------------------- activations ----
// T/include/tiny-cuda-nn/common_device.h:68-114 (warp_activation), applied to
// an fp32 pre-activation; the caller rounds the result to fp16.
inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }
// tcnn's ReLU is a PRODUCT in the network's precision: `frag.x[t] * (T)((T)frag.x[t] > (T)0.0f)` with T = __half
// (T/include/tiny-cuda-nn/common_device.h:71-76; R/include/nerf-cuda/nerf_network.h:36-37 for the sigma activation): a negative
// value gives -0, NaN stays NaN and -inf -- an fp16 accumulator below -65504 -- becomes NaN (-inf * 0); max(x, 0) would give +0
// for all three.  v is rounded to fp16 first (the reference's accumulator IS an fp16 value); the product of an fp16 value with
// 0 or 1 is exact.  The HIP path clamps with v_pk_max_f16 instead: DESIGN.md deviation D-10, tests/test_parity_gpu.py.
inline float relu_of_half(float h) {  // h: an fp16 value held in a float
  return h > 0.0f ? h : h * 0.0f;  // == h * (half)(h > 0): negative -> -0, NaN and -inf -> NaN
}
inline float relu_half(float v) { return relu_of_half(h2f(f2h(v))); }
inline float activate(uint32_t act, float v) {
  switch (act) {
    case NRF_ACT_RELU: return relu_half(v);
    case NRF_ACT_EXPONENTIAL: return expf(v);
    case NRF_ACT_SIGMOID: return logistic(v);
    case NRF_ACT_SQUAREPLUS: {
      const float x = v * 10.0f;
      return 0.5f * (x + sqrtf(x * x + 4)) / 10.0f;
    }
    case NRF_ACT_SOFTPLUS: return logf(expf(v * 10.0f) + 1.0f) / 10.0f;
    case NRF_ACT_SINE: return sinf(v);
    default: return v;
  }
}

inline uint32_t next_multiple(uint32_t v, uint32_t d) { return (v + d - 1) / d * d; }

}  // namespace

// ------------------------------------------------------------ the model ----
struct nrfo_model {
  nrf_model_desc d;
  nrf_level_table lv;
  uint32_t feat_width;     // L*F padded to 16 (nerf_network.h:103-111)
  uint32_t dir_width;      // padded dir encoding width (alignment 16)
  uint32_t dir_raw;        // unpadded
  uint32_t rgb_in;         // nerf_network.h:127-130
  uint32_t W;              // n_neurons
  std::vector<std::vector<float>> dens_w;  // per layer, fp16-rounded values as float, [out][in]
  std::vector<std::vector<float>> rgb_w;
  std::vector<uint32_t> dens_dims, rgb_dims;  // layer widths: in, W, ..., 16
  std::vector<uint16_t> grid;                 // fp16 table
  std::vector<float> density_grid;
  uint32_t mlp_acc_block = 0;  // nrfo_set_mlp_accumulate: 0 = fp32 sums; n = fp16 accumulator updated every n products
  int contract = 0;            // nrfo_set_contract: a * b + c as one fmaf wherever the reference's source has it in one expression
                               // (1: the rule of `mad` above; 2: the OTHER choice at every site where the compiler has one -- the sensitivity run)
  float scale_dev[16] = {};    // the level scale as the KERNEL computes it under contraction (grid.h:189): fmaf(exp2f(..), base, -1)
};

extern "C" {

const char* nrfo_last_error(void) { return g_err.c_str(); }
// The entry points that need no model convert halves as well: on a CPU without the F16C instructions a -mf16c build answers
// them with the software forms (the same bits) instead of an illegal instruction; everything that takes a model is behind
// nrfo_create's refusal.
static bool f16c_usable() {
#ifdef __F16C__
  static const bool ok = __builtin_cpu_supports("f16c");
  return ok;
#else
  return true;  // (f2h / h2f ARE the software forms)
#endif
}
uint16_t nrfo_f32_to_f16(float f) { return f16c_usable() ? f2h(f) : f2h_soft(f); }
float nrfo_f16_to_f32(uint16_t h) { return f16c_usable() ? h2f(h) : h2f_soft(h); }
float nrfo_pcg32_first_float(uint64_t initstate, uint64_t initseq) { return pcg32_first_float(initstate, initseq); }
float nrfo_activation(uint32_t act, float v) {
  if (!f16c_usable() && act == NRF_ACT_RELU) {
    const float h = h2f_soft(f2h_soft(v));
    return h * (h > 0.0f ? 1.0f : 0.0f);
  }
  return activate(act, v);
}
uint16_t nrfo_f32_to_f16_soft(float f) { return f2h_soft(f); }
float nrfo_f16_to_f32_soft(uint16_t h) { return h2f_soft(h); }
// the conversions in use against their bit-level definition: every 16-bit pattern one way, every `stride`-th 32-bit pattern
// the other (stride 1: all 2^32); returns the number of disagreements (NaNs must agree in their bits as well)
uint64_t nrfo_fp16_selfcheck(uint32_t stride) {
  if (!f16c_usable()) return 0;  // nothing but the software forms can run here (nrfo_create refuses this build on this CPU)
  uint64_t bad = 0;
  for (uint32_t h = 0; h < 65536u; ++h) {
    const float a = h2f((uint16_t)h), b = h2f_soft((uint16_t)h);
    uint32_t ab, bb;
    std::memcpy(&ab, &a, 4);
    std::memcpy(&bb, &b, 4);
    // a SIGNALLING half NaN (never produced by f2h) comes back quieted from the instruction, untouched from the software form
    if (a != a && b != b) { ab |= 0x00400000u; bb |= 0x00400000u; }
    if (ab != bb) ++bad;
  }
  if (stride == 0) stride = 1;
  const int64_t n = (int64_t)((0xffffffffull + stride) / stride);
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t x = (uint32_t)((uint64_t)i * stride);
    float f;
    std::memcpy(&f, &x, 4);
    if (f2h(f) != f2h_soft(f)) ++bad;
  }
  return bad;
}
const char* nrfo_fp16_backend(void) {
#ifdef __F16C__
  return "f16c";
#else
  return "software";
#endif
}
int nrfo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// R/include/nerf-cuda/render_utils.h:68-77
void nrfo_nerf_matrix_to_ngp(const float p[16], float s, float o[16]) {
  // rows (1,2,0) of the input, columns 1 and 2 negated, translation scaled (+ offset 0)
  const int rows[3] = {1, 2, 0};
  for (int r = 0; r < 3; ++r) {
    const float* src = p + 4 * rows[r];
    o[4 * r + 0] = src[0];
    o[4 * r + 1] = -src[1];
    o[4 * r + 2] = -src[2];
    o[4 * r + 3] = src[3] * s + 0.0f;
  }
  o[12] = 0;
  o[13] = 0;
  o[14] = 0;
  o[15] = 1;
}

// T/include/tiny-cuda-nn/encodings/grid.h:81-98 for N_DIMS = 3
uint32_t nrfo_fast_hash3(uint32_t x, uint32_t y, uint32_t z) {
  return (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
}

// R/src/render_buffer.cu:413-429 (colormap_turbo) and :431-477 (overlay_depth_kernel); surface [H][W][4] in place
static float dot4_(const float a[4], float b0, float b1, float b2, float b3) { return ((a[0] * b0 + a[1] * b1) + a[2] * b2) + a[3] * b3; }
void nrfo_rb_overlay_depth(float* surface, int W, int H, float alpha, const float* depth, float depth_scale, int img_w, int img_h,
                           int fov_axis, float zoom, float center_x, float center_y) {
  const float kR4[4] = {0.13572138f, 4.61539260f, -42.66032258f, 132.13108234f};
  const float kG4[4] = {0.09140261f, 2.19418839f, 4.84296658f, -14.18503333f};
  const float kB4[4] = {0.10667330f, 12.64194608f, -60.58204836f, 110.36276771f};
  const float kR2[2] = {-152.94239396f, 59.28637943f}, kG2[2] = {4.27729857f, 2.82956604f}, kB2[2] = {-89.90310912f, 27.34824973f};
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float scale = (float)(fov_axis == 0 ? img_w : img_h) / (float)(fov_axis == 0 ? W : H);
      float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
      fx -= (float)W * 0.5f; fx /= zoom; fx += center_x * (float)W;
      fy -= (float)H * 0.5f; fy /= zoom; fy += center_y * (float)H;
      const float u = (fx - (float)W * 0.5f) * scale + (float)img_w * 0.5f;
      const float v = (fy - (float)H * 0.5f) * scale + (float)img_h * 0.5f;
      const int srcx = (int)std::floor(u), srcy = (int)std::floor(v);
      float color[4] = {0.f, 0.f, 0.f, 0.f};
      if (!(srcx >= img_w || srcy >= img_h || srcx < 0 || srcy < 0)) {
        float t = depth[(size_t)srcx + (size_t)img_w * srcy] * depth_scale;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        if (!(t == t)) t = 0.0f;
        const float x2 = t * t, x3 = x2 * t, v20 = x3 * t, v21 = x3 * x2;
        color[0] = dot4_(kR4, 1.0f, t, x2, x3) + (v20 * kR2[0] + v21 * kR2[1]);
        color[1] = dot4_(kG4, 1.0f, t, x2, x3) + (v20 * kG2[0] + v21 * kG2[1]);
        color[2] = dot4_(kB4, 1.0f, t, x2, x3) + (v20 * kB2[0] + v21 * kB2[1]);
        color[3] = 1.0f;
      }
      float* px = surface + 4 * ((size_t)y * W + x);
      const float ia = 1.f - alpha;
      for (int k = 0; k < 4; ++k) px[k] = color[k] * alpha + px[k] * ia;
    }
}

}  // extern "C"

namespace {

// T/include/tiny-cuda-nn/encodings/grid.h:899-931 (ctor) and :186-190 (kernel)
int level_table(const nrf_model_desc& d, nrf_level_table& t) {
  if (d.n_levels == 0 || d.n_levels > 16) return fail(NRF_E_UNSUPPORTED, "n_levels must be 1..16");
  t.n_levels = d.n_levels;
  const float log2_pls = std::log2(d.per_level_scale);  // float overload, as std::log2(float)
  uint32_t offset = 0;
  for (uint32_t i = 0; i < d.n_levels; ++i) {
    const float scale = exp2f((float)i * log2_pls) * (float)d.base_resolution - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
    uint32_t params = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
    params = next_multiple(params, 8u);
    if (d.grid_type == NRF_GRID_TILED) {
      params = std::min(params, d.base_resolution * d.base_resolution * d.base_resolution);
    } else if (d.grid_type == NRF_GRID_HASH) {
      params = std::min(params, 1u << d.log2_hashmap_size);
    }
    t.offset[i] = offset;
    t.resolution[i] = res;
    t.scale[i] = scale;
    offset += params;
  }
  t.offset[d.n_levels] = offset;
  return NRF_OK;
}

// T/src/fully_fused_mlp.cu:636-687: first [W x in], (h-1) x [W x W], last [padded_out x W]
void mlp_dims(uint32_t in, uint32_t W, uint32_t hidden, std::vector<uint32_t>& dims) {
  dims.clear();
  dims.push_back(in);
  for (uint32_t i = 0; i < hidden; ++i) dims.push_back(W);
  dims.push_back(16);
}
uint64_t mlp_params(const std::vector<uint32_t>& dims) {
  uint64_t n = 0;
  for (size_t i = 0; i + 1 < dims.size(); ++i) n += (uint64_t)dims[i] * dims[i + 1];
  return n;
}

int dir_widths(const nrf_model_desc& d, uint32_t& raw, uint32_t& padded) {
  switch (d.dir_encoding) {
    case NRF_DIR_SH:  // spherical_harmonics.h:394-412: degree 1..8
      if (d.sh_degree < 1 || d.sh_degree > 8) return fail(NRF_E_INVALID, "SphericalHarmonics: degree must be 1..8");
      raw = d.sh_degree * d.sh_degree;
      break;
    case NRF_DIR_FREQUENCY: raw = 3 * d.n_frequencies * 2; break;
    case NRF_DIR_IDENTITY: raw = 3; break;
    default: return fail(NRF_E_UNSUPPORTED, "unknown dir encoding");
  }
  padded = next_multiple(raw, 16u);  // T/src/encoding.cu:97-117 with alignment 16
  if (raw == 0 || padded > 112) return fail(NRF_E_UNSUPPORTED, "oracle: direction encoding wider than 112");
  return NRF_OK;
}

}  // namespace

extern "C" {

int nrfo_create(const nrf_model_desc* d, nrfo_model** out) {
  if (!d || !out || !d->params) return fail(NRF_E_INVALID, "null argument");
  const uint32_t F = d->n_features_per_level;
  if (F != 1 && F != 2 && F != 4 && F != 8)  // grid.h:1403-1411
    return fail(NRF_E_INVALID, "GridEncoding: n_features_per_level must be 1, 2, 4, or 8.");
  if (d->interpolation > NRF_INTERP_SMOOTHSTEP) return fail(NRF_E_INVALID, "Invalid interpolation type");
  if (d->n_neurons != 16 && d->n_neurons != 32 && d->n_neurons != 64 && d->n_neurons != 128)  // fully_fused_mlp.cu:700-725
    return fail(NRF_E_INVALID, "FullyFusedMLP: n_neurons must be 16, 32, 64 or 128");
  if (d->density_hidden_layers < 1 || d->rgb_hidden_layers < 1)  // fully_fused_mlp.cu:653-655
    return fail(NRF_E_INVALID, "FullyFusedMLP requires at least 1 hidden layer (3 layers in total).");
  if (d->density_n_output < 1 || d->density_n_output > 16)  // wider outputs go through CUTLASS in tcnn (out of scope)
    return fail(NRF_E_UNSUPPORTED, "oracle: density n_output_dims must be 1..16");
#ifdef __F16C__
  if (!__builtin_cpu_supports("f16c")) return fail(NRF_E_UNSUPPORTED, "this oracle was built with -mf16c and the CPU has no F16C: rebuild oracle/ here");
#endif
  nrfo_model* m = new nrfo_model;
  m->d = *d;
  int rc = level_table(*d, m->lv);
  if (rc) { delete m; return rc; }
  {
    const float log2_pls = std::log2(d->per_level_scale);
    for (uint32_t i = 0; i < d->n_levels; ++i) m->scale_dev[i] = fmaf(exp2f((float)i * log2_pls), (float)d->base_resolution, -1.0f);
  }
  rc = dir_widths(*d, m->dir_raw, m->dir_width);
  if (rc) { delete m; return rc; }
  m->W = d->n_neurons;
  m->feat_width = next_multiple(d->n_levels * F, 16u);
  m->rgb_in = next_multiple(m->dir_width + 16u, 16u);
  mlp_dims(m->feat_width, m->W, d->density_hidden_layers, m->dens_dims);
  mlp_dims(m->rgb_in, m->W, d->rgb_hidden_layers, m->rgb_dims);
  const uint64_t n_grid = (uint64_t)m->lv.offset[d->n_levels] * F;  // grid.h:927
  const uint64_t expect = mlp_params(m->dens_dims) + mlp_params(m->rgb_dims) + n_grid;
  if (d->n_params != expect) {  // R/include/nerf-cuda/nerf_network.h:425-427
    delete m;
    return fail(NRF_E_PARAMS, "Can't set params because number of parameters and model size do not match");
  }
  const uint64_t H = d->density_grid_size;
  if (d->density_grid && d->n_density_grid != H * H * H * d->cascade) {  // R/src/nerf_render.cu:467-469
    delete m;
    return fail(NRF_E_PARAMS, "Incompatible number of grid cascades.");
  }
  // deserialize: fp32 -> fp16 cast of every parameter (nerf_network.h:434-436),
  // order density MLP | rgb MLP | grid (nerf_network.h:273-291)
  const float* p = d->params;
  auto take = [&](const std::vector<uint32_t>& dims, std::vector<std::vector<float>>& w) {
    w.resize(dims.size() - 1);
    for (size_t l = 0; l + 1 < dims.size(); ++l) {
      const size_t n = (size_t)dims[l] * dims[l + 1];
      w[l].resize(n);
      for (size_t i = 0; i < n; ++i) w[l][i] = h2f(f2h(p[i]));
      p += n;
    }
  };
  take(m->dens_dims, m->dens_w);
  take(m->rgb_dims, m->rgb_w);
  m->grid.resize(n_grid);
  for (uint64_t i = 0; i < n_grid; ++i) m->grid[i] = f2h(p[i]);
  if (d->density_grid) m->density_grid.assign(d->density_grid, d->density_grid + d->n_density_grid);
  else m->density_grid.assign((size_t)(H * H * H * d->cascade), 0.0f);  // none in the snapshot: see nrfo_density_grid
  m->d.params = nullptr;
  m->d.density_grid = nullptr;
  *out = m;
  return NRF_OK;
}

void nrfo_destroy(nrfo_model* m) { delete m; }

// padded encoding widths = the input widths of the two MLPs (nerf_network.h:103-130)
void nrfo_widths(const nrfo_model* m, uint32_t* feat_width, uint32_t* dir_width) {
  if (feat_width) *feat_width = m->feat_width;
  if (dir_width) *dir_width = m->dir_width;
}

// the accumulator arithmetic of the two MLPs (see mlp_one); the default is NRFO_ACC_FP32
int nrfo_set_mlp_accumulate(nrfo_model* m, int mode) {
  if (!m) return fail(NRF_E_INVALID, "null model");
  if (mode != NRFO_ACC_FP32 && mode != NRFO_ACC_FP16_STEP && mode != NRFO_ACC_FP16_K4 && mode != NRFO_ACC_FP16_K8 &&
      mode != NRFO_ACC_FP16_K16)
    return fail(NRF_E_INVALID, "mlp accumulate mode must be 0 (fp32), 1, 4, 8 or 16 (fp16 accumulator per K block)");
  m->mlp_acc_block = (uint32_t)mode;
  return NRF_OK;
}

// see `mad` above; the default (0) is the arithmetic contract shared with the HIP path
int nrfo_set_contract(nrfo_model* m, int on) {
  if (!m) return fail(NRF_E_INVALID, "null model");
  if (on < 0 || on > 2) return fail(NRF_E_INVALID, "contract mode must be 0 (off), 1 (left product fused, aggressive fusion) or 2 (the other choices)");
  m->contract = on;
  return NRF_OK;
}

// T/include/tiny-cuda-nn/encodings/grid.h:100-117
uint32_t nrfo_grid_index(const nrfo_model* m, uint32_t level, uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t hashmap_size = m->lv.offset[level + 1] - m->lv.offset[level];
  const uint32_t res = m->lv.resolution[level];
  const uint32_t pg[3] = {x, y, z};
  uint32_t stride = 1, index = 0;
  for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
    index += pg[dim] * stride;
    stride *= res;
  }
  if (m->d.grid_type == NRF_GRID_HASH && hashmap_size < stride) index = nrfo_fast_hash3(x, y, z);
  return index % hashmap_size;
}

}  // extern "C"

namespace {

// One sample of kernel_grid<half,3,F>: T/include/tiny-cuda-nn/encodings/grid.h:186-267,
// pos_fract: T/include/tiny-cuda-nn/common_device.h:414-422, smoothstep :379-381.
void encode_grid_one(const nrfo_model* m, const float p01[3], uint16_t* out) {
  const uint32_t L = m->d.n_levels, F = m->d.n_features_per_level;
  for (uint32_t level = 0; level < L; ++level) {
    // (the table geometry -- resolution, offsets -- is host arithmetic either way, grid.h:899-931; the kernel recomputes `scale`)
    const float scale = m->contract ? m->scale_dev[level] : m->lv.scale[level];
    float pos[3];
    uint32_t pg[3];
    for (int dim = 0; dim < 3; ++dim) {
      float v = mad(m->contract, p01[dim], scale, 0.5f);  // pos_fract, common_device.h:416: input * scale + 0.5f
      const int tmp = (int)floorf(v);
      pg[dim] = (uint32_t)tmp;
      float fr = v - (float)tmp;
      if (m->d.interpolation == NRF_INTERP_SMOOTHSTEP) {  // val*val*(3.0f - 2.0f*val)
        const float sq = fr * fr;
        const float b = 2.0f * fr;
        fr = sq * (3.0f - b);
      }
      pos[dim] = fr;
    }
    const uint16_t* table = m->grid.data() + (size_t)m->lv.offset[level] * F;
    uint16_t* res = out + (size_t)level * F;
    if (m->d.interpolation == NRF_INTERP_NEAREST) {  // grid.h:215-232: the entry at floor(pos), no weights
      const uint32_t e = nrfo_grid_index(m, level, pg[0], pg[1], pg[2]);
      for (uint32_t f = 0; f < F; ++f) res[f] = table[(size_t)e * F + f];
      continue;
    }
    for (uint32_t f = 0; f < F; ++f) res[f] = 0;  // fp16 accumulators, grid.h:236
    for (uint32_t idx = 0; idx < 8; ++idx) {
      float weight = 1;
      uint32_t pl[3];
      for (int dim = 0; dim < 3; ++dim) {
        if ((idx & (1u << dim)) == 0) {
          weight *= 1 - pos[dim];
          pl[dim] = pg[dim];
        } else {
          weight *= pos[dim];
          pl[dim] = pg[dim] + 1;
        }
      }
      const uint32_t e = nrfo_grid_index(m, level, pl[0], pl[1], pl[2]);
      for (uint32_t f = 0; f < F; ++f)  // grid.h:260: result += (T)(weight * data)
        res[f] = hadd(res[f], f2h(weight * h2f(table[(size_t)e * F + f])));
    }
  }
  for (uint32_t j = F * L; j < m->feat_width; ++j) out[j] = 0;  // alignment padding is ZERO for grids, grid.h:959-969
}

// kernel_sh's polynomial table, T/include/tiny-cuda-nn/encodings/spherical_harmonics.h:66-152 (degree <= 8): the
// published real-SH formulae as that file states them, each evaluated in C++ operator order with every fp32
// operation individually rounded.
template <typename S>
void sh_coefficients(uint32_t degree, S x, S y, S z, ShOut* c) {
  const S xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  const S x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
  const S x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
  c[0] = 0.28209479177387814f;
  if (degree <= 1) return;
  c[1] = -0.48860251190291987f * y;
  c[2] = 0.48860251190291987f * z;
  c[3] = -0.48860251190291987f * x;
  if (degree <= 2) return;
  c[4] = 1.0925484305920792f * xy;
  c[5] = -1.0925484305920792f * yz;
  c[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  c[7] = -1.0925484305920792f * xz;
  c[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  if (degree <= 3) return;
  c[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  c[10] = 2.8906114426405538f * xy * z;
  c[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  c[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  c[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  c[14] = 1.4453057213202769f * z * (x2 - y2);
  c[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  if (degree <= 4) return;
  c[16] = 2.5033429417967046f * xy * (x2 - y2);
  c[17] = 1.7701307697799304f * yz * (-3.0f * x2 + y2);
  c[18] = 0.94617469575756008f * xy * (7.0f * z2 - 1.0f);
  c[19] = 0.66904654355728921f * yz * (3.0f - 7.0f * z2);
  c[20] = -3.1735664074561294f * z2 + 3.7024941420321507f * z4 + 0.31735664074561293f;
  c[21] = 0.66904654355728921f * xz * (3.0f - 7.0f * z2);
  c[22] = 0.47308734787878004f * (x2 - y2) * (7.0f * z2 - 1.0f);
  c[23] = 1.7701307697799304f * xz * (-x2 + 3.0f * y2);
  c[24] = -3.7550144126950569f * x2 * y2 + 0.62583573544917614f * x4 + 0.62583573544917614f * y4;
  if (degree <= 5) return;
  c[25] = 0.65638205684017015f * y * (10.0f * x2 * y2 - 5.0f * x4 - y4);
  c[26] = 8.3026492595241645f * xy * z * (x2 - y2);
  c[27] = -0.48923829943525038f * y * (3.0f * x2 - y2) * (9.0f * z2 - 1.0f);
  c[28] = 4.7935367849733241f * xy * z * (3.0f * z2 - 1.0f);
  c[29] = 0.45294665119569694f * y * (14.0f * z2 - 21.0f * z4 - 1.0f);
  c[30] = 0.1169503224534236f * z * (-70.0f * z2 + 63.0f * z4 + 15.0f);
  c[31] = 0.45294665119569694f * x * (14.0f * z2 - 21.0f * z4 - 1.0f);
  c[32] = 2.3967683924866621f * z * (x2 - y2) * (3.0f * z2 - 1.0f);
  c[33] = -0.48923829943525038f * x * (x2 - 3.0f * y2) * (9.0f * z2 - 1.0f);
  c[34] = 2.0756623148810411f * z * (-6.0f * x2 * y2 + x4 + y4);
  c[35] = 0.65638205684017015f * x * (10.0f * x2 * y2 - x4 - 5.0f * y4);
  if (degree <= 6) return;
  c[36] = 1.3663682103838286f * xy * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
  c[37] = 2.3666191622317521f * yz * (10.0f * x2 * y2 - 5.0f * x4 - y4);
  c[38] = 2.0182596029148963f * xy * (x2 - y2) * (11.0f * z2 - 1.0f);
  c[39] = -0.92120525951492349f * yz * (3.0f * x2 - y2) * (11.0f * z2 - 3.0f);
  c[40] = 0.92120525951492349f * xy * (-18.0f * z2 + 33.0f * z4 + 1.0f);
  c[41] = 0.58262136251873131f * yz * (30.0f * z2 - 33.0f * z4 - 5.0f);
  c[42] = 6.6747662381009842f * z2 - 20.024298714302954f * z4 + 14.684485723822165f * z6 - 0.31784601133814211f;
  c[43] = 0.58262136251873131f * xz * (30.0f * z2 - 33.0f * z4 - 5.0f);
  c[44] = 0.46060262975746175f * (x2 - y2) * (11.0f * z2 * (3.0f * z2 - 1.0f) - 7.0f * z2 + 1.0f);
  c[45] = -0.92120525951492349f * xz * (x2 - 3.0f * y2) * (11.0f * z2 - 3.0f);
  c[46] = 0.50456490072872406f * (11.0f * z2 - 1.0f) * (-6.0f * x2 * y2 + x4 + y4);
  c[47] = 2.3666191622317521f * xz * (10.0f * x2 * y2 - x4 - 5.0f * y4);
  c[48] = 10.247761577878714f * x2 * y4 - 10.247761577878714f * x4 * y2 + 0.6831841051919143f * x6 - 0.6831841051919143f * y6;
  if (degree <= 7) return;
  c[49] = 0.70716273252459627f * y * (-21.0f * x2 * y4 + 35.0f * x4 * y2 - 7.0f * x6 + y6);
  c[50] = 5.2919213236038001f * xy * z * (-10.0f * x2 * y2 + 3.0f * x4 + 3.0f * y4);
  c[51] = -0.51891557872026028f * y * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + 5.0f * x4 + y4);
  c[52] = 4.1513246297620823f * xy * z * (x2 - y2) * (13.0f * z2 - 3.0f);
  c[53] = -0.15645893386229404f * y * (3.0f * x2 - y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
  c[54] = 0.44253269244498261f * xy * z * (-110.0f * z2 + 143.0f * z4 + 15.0f);
  c[55] = 0.090331607582517306f * y * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
  c[56] = 0.068284276912004949f * z * (315.0f * z2 - 693.0f * z4 + 429.0f * z6 - 35.0f);
  c[57] = 0.090331607582517306f * x * (-135.0f * z2 + 495.0f * z4 - 429.0f * z6 + 5.0f);
  c[58] = 0.07375544874083044f * z * (x2 - y2) * (143.0f * z2 * (3.0f * z2 - 1.0f) - 187.0f * z2 + 45.0f);
  c[59] = -0.15645893386229404f * x * (x2 - 3.0f * y2) * (13.0f * z2 * (11.0f * z2 - 3.0f) - 27.0f * z2 + 3.0f);
  c[60] = 1.0378311574405206f * z * (13.0f * z2 - 3.0f) * (-6.0f * x2 * y2 + x4 + y4);
  c[61] = -0.51891557872026028f * x * (13.0f * z2 - 1.0f) * (-10.0f * x2 * y2 + x4 + 5.0f * y4);
  c[62] = 2.6459606618019f * z * (15.0f * x2 * y4 - 15.0f * x4 * y2 + x6 - y6);
  c[63] = 0.70716273252459627f * x * (-35.0f * x2 * y4 + 21.0f * x4 * y2 - x6 + 7.0f * y6);
}

// T/include/tiny-cuda-nn/encodings/spherical_harmonics.h:57-152 (degree <= 8),
// T/include/tiny-cuda-nn/encodings/frequency.h:56-92, identity.h:64-65.
void encode_dir_one(const nrfo_model* m, const float d01[3], uint16_t* out) {
  const nrf_model_desc& d = m->d;
  const uint32_t pad = m->dir_width - m->dir_raw;
  if (d.dir_encoding == NRF_DIR_SH) {
    uint16_t* o = out;
    for (uint32_t j = 0; j < pad; ++j) *o++ = f2h(1.0f);  // SH pads in FRONT (:57-64)
    const float x = d01[0] * 2.f - 1.f, y = d01[1] * 2.f - 1.f, z = d01[2] * 2.f - 1.f;  // (x 2: exact, fused or not)
    ShOut c[64];
    tl_fuse_right = m->contract == 2;
    if (m->contract) sh_coefficients<Fx>(d.sh_degree, Fx(x), Fx(y), Fx(z), c);
    else sh_coefficients<float>(d.sh_degree, x, y, z, c);
    for (uint32_t j = 0; j < m->dir_raw; ++j) o[j] = f2h(c[j].v);
  } else if (d.dir_encoding == NRF_DIR_FREQUENCY) {
    const float PI = 3.14159265358979323846f;
    const uint32_t nf = d.n_frequencies;
    for (uint32_t j = 0; j < m->dir_raw; ++j) {
      const uint32_t feat = j / (nf * 2);
      const uint32_t log2_frequency = (j / 2) % nf;
      const float phase_shift = (float)(j % 2) * (PI / 2);
      const float x = scalbnf(d01[feat], (int)log2_frequency);
      const float input = mad(m->contract, x, PI, phase_shift);  // frequency.h:88
      out[j] = f2h(sinf(input));  // reference uses __sinf (approximate); tolerance applies
    }
    for (uint32_t j = m->dir_raw; j < m->dir_width; ++j) out[j] = f2h(1.0f);  // trailing pad
  } else {  // Identity, scale 1 offset 0
    for (uint32_t j = 0; j < 3; ++j) out[j] = f2h(d01[j]);
    for (uint32_t j = 3; j < m->dir_width; ++j) out[j] = f2h(1.0f);
  }
}

// y = act(W x) chains, T/src/fully_fused_mlp.cu:500-558.  fp16 in; acc_block == 0 (the arithmetic contract shared with
// the HIP path): fp32 accumulate in ascending k, activation on the fp32 sum, fp16 store per layer.
// acc_block == n > 0 (nrfo_set_mlp_accumulate): the reference's own accumulator type.  Its result fragments are
// `wmma::fragment<accumulator, 16, 16, 16, __half>` in every layer (fully_fused_mlp.cu:69 hidden, :334 input, :437 last;
// OUT_T = __half, :573-634) and one `mma_sync` per 16-wide K block adds 16 exact fp16 x fp16 products to it
// (:100-104, :373-376, :461-465): the running sum is an fp16 value after every block of n = 16 products.  How the tensor
// core sums INSIDE a block is not specified; here the block's products are summed in fp32 (ascending k, each product
// exact in fp32) together with the accumulator and rounded to fp16 once (RNE).  n = 1 rounds after every product (the
// pessimistic bound), n = 4 / 8 are the K granularities of older HMMA forms.  The activation then works on the fp16
// accumulator value (`warp_activation<__half>`, common_device.h:68-114: float math on the fp16 value, fp16 result).
void mlp_one(const std::vector<std::vector<float>>& w, const std::vector<uint32_t>& dims,
             uint32_t act, uint32_t out_act, uint32_t acc_block, const float* in, float* out /*16, fp16-rounded*/) {
  float buf0[128], buf1[128];
  const float* cur = in;
  float* nxt = buf0;
  const size_t nl = dims.size() - 1;
  for (size_t l = 0; l < nl; ++l) {
    const uint32_t K = dims[l], N = dims[l + 1];
    const float* W = w[l].data();
    const bool last = (l + 1 == nl);
    float* dst = last ? out : nxt;
    for (uint32_t o = 0; o < N; ++o) {
      float acc = 0.0f;
      const float* row = W + (size_t)o * K;
      if (acc_block == 0) {
        for (uint32_t k = 0; k < K; ++k) acc += row[k] * cur[k];
      } else {
        for (uint32_t kb = 0; kb < K; kb += acc_block) {
          float part = acc;  // the fp16 accumulator enters the block's sum
          const uint32_t ke = kb + acc_block < K ? kb + acc_block : K;
          for (uint32_t k = kb; k < ke; ++k) part += row[k] * cur[k];
          acc = h2f(f2h(part));
        }
      }
      const uint32_t a = last ? out_act : act;
      if (a == NRF_ACT_RELU) {  // relu_half with ONE conversion pair: its result is an fp16 value already (h, +-0 or NaN)
        // relu_half's value up to the sign of a zero: negative -> +0 here, -0 there.  No later value can tell -- every consumer of
        // a hidden activation is a dot product whose running sum starts at +0, and +0 + (-0) = +0 --, while NaN and -inf -> NaN
        // (the observable part of tcnn's product form) are kept.  The product form costs the whole MLP 35 % on the CPU (the sign
        // of a pre-activation is a coin toss), and bench.py times this loop as the CPU baseline.
        const float h = h2f(f2h(acc));
        float r = h > 0.0f ? h : 0.0f;
        if (!(h >= -FLT_MAX)) r = h * 0.0f;  // NaN, -inf: NaN
        dst[o] = r;
      } else {
        dst[o] = h2f(f2h(activate(a, acc)));
      }
    }
    cur = dst;
    nxt = (dst == buf0) ? buf1 : buf0;
  }
}

// R/include/nerf-cuda/nerf_network.h:148-196 for one sample on encoded inputs.
// out4 = (r, g, b, sigma) as fp16 bits: rows 0..2 and 3 of network_output.
void network_encoded_one(const nrfo_model* m, const uint16_t* feat, const uint16_t* dirfeat,
                         uint16_t* out4) {
  float in[128], dens[16], rgbin[128], rgb[16];
  for (uint32_t j = 0; j < m->feat_width; ++j) in[j] = h2f(feat[j]);
  mlp_one(m->dens_w, m->dens_dims, m->d.density_activation, m->d.density_output_activation, m->mlp_acc_block, in, dens);
  for (uint32_t j = 0; j < 16; ++j) rgbin[j] = dens[j];  // rows 0..15 (nerf_network.h:162-164)
  for (uint32_t j = 0; j < m->dir_width; ++j) rgbin[16 + j] = h2f(dirfeat[j]);  // rows 16.. (:177-182)
  mlp_one(m->rgb_w, m->rgb_dims, m->d.rgb_activation, m->d.rgb_output_activation, m->mlp_acc_block, rgbin, rgb);
  out4[0] = f2h(rgb[0]);
  out4[1] = f2h(rgb[1]);
  out4[2] = f2h(rgb[2]);
  // extract_density, nerf_network.h:49-61 + wrap_a_activation :32-47: fp32 math, fp16 store
  float s = dens[0];
  switch (m->d.sigma_activation) {
    case NRF_ACT_RELU: s = relu_half(s); break;
    case NRF_ACT_EXPONENTIAL: s = expf(s); break;
    case NRF_ACT_SIGMOID: s = logistic(s); break;
    default: break;  // wrap_a_activation returns the value unchanged otherwise
  }
  out4[3] = f2h(s);
}

// Affine maps R/src/nerf_render.cu:311-314 + network + decompose (render_utils.h:308-334)
// (+ density scale nerf_render.cu:328, as a float multiply: DESIGN.md deviation D-8).
void network_one(const nrfo_model* m, float density_scale, const float xyz[3], const float dir[3],
                 float* sigma, float* rgb) {
  const float wpos = (float)(1.0 / (2 * (double)m->d.bound));  // `1.0/(2 * m_bound)` -> float arg
  float p01[3], d01[3];
  for (int c = 0; c < 3; ++c) {
    p01[c] = mad(m->contract, wpos, xyz[c], 0.5f);  // linear_transformer, common_device.cuh:34: weight * input + bias
    d01[c] = mad(m->contract, 0.5f, dir[c], 0.5f);
  }
  uint16_t feat[128], dirfeat[128], out4[4];
  encode_grid_one(m, p01, feat);
  encode_dir_one(m, d01, dirfeat);
  network_encoded_one(m, feat, dirfeat, out4);
  rgb[0] = h2f(out4[0]);
  rgb[1] = h2f(out4[1]);
  rgb[2] = h2f(out4[2]);
  float s = h2f(out4[3]);
  if (density_scale != 1.0f) s = density_scale * s;
  *sigma = s;
}

inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// set_rays_d (render_utils.h:31-52): Eigen's fixed-size reductions are
// unrolled as a + (b + c) (redux_novec_unroller splits [0,1) | [1,3)).
inline void ray_dir(const float R[9], const float cam[4], int px, int py, float d[3], int fm = 0) {
  const float i = (float)((double)px + 0.5);
  const float j = (float)((double)py + 0.5);
  const float zs = 1;
  const float xs = (i - cam[2]) / cam[0] * zs;
  const float ys = (j - cam[3]) / cam[1] * zs;
  const float n = sqrtf(mad(fm, xs, xs, mad2(fm, ys, ys, zs, zs)));
  const float v[3] = {xs / n, ys / n, zs / n};
  for (int r = 0; r < 3; ++r) d[r] = mad(fm, R[3 * r + 0], v[0], mad2(fm, R[3 * r + 1], v[1], R[3 * r + 2], v[2]));
}

// kernel_near_far_from_aabb, render_utils.h:353-391
inline void near_far(const float* aabb, const float o[3], const float d[3], float min_near,
                     float* near_out, float* far_out) {
  const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
  float near = (aabb[0] - o[0]) * rdx, far = (aabb[3] - o[0]) * rdx;
  if (near > far) std::swap(near, far);
  float near_y = (aabb[1] - o[1]) * rdy, far_y = (aabb[4] - o[1]) * rdy;
  if (near_y > far_y) std::swap(near_y, far_y);
  if (near > far_y || near_y > far) {
    *near_out = *far_out = FLT_MAX;
    return;
  }
  if (near_y > near) near = near_y;
  if (far_y < far) far = far_y;
  float near_z = (aabb[2] - o[2]) * rdz, far_z = (aabb[5] - o[2]) * rdz;
  if (near_z > far_z) std::swap(near_z, far_z);
  if (near > far_z || near_z > far) {
    *near_out = *far_out = FLT_MAX;
    return;
  }
  if (near_z > near) near = near_z;
  if (far_z < far) far = far_z;
  if (near < min_near) near = min_near;
  *near_out = near;
  *far_out = far;
}

// kernel_march_rays for one ray, render_utils.h:556-653.  Returns the number
// of emitted samples; xyz[k][3], delta[k][2].  *t_io is the march's own t.
// perturb != 0 (:585-589): t += MIN_STEPSIZE() * pcg32(n, perturb).next_float() ahead of the loop, n = the ray's position in
// the call's alive list; last_t starts at the shifted t, so the shift does not enter deltas[1] (nor rays_t).
inline uint32_t march_one(const nrfo_model* m, float dt_gamma, const float o[3], const float d[3],
                          float far, float t, uint32_t n_step, float* xyz, float* delta, uint32_t perturb = 0, uint64_t n = 0) {
  const float bound = m->d.bound;
  const uint32_t C = m->d.cascade, H = m->d.density_grid_size;
  const float* grid = m->density_grid.data();
  const float density_thresh = fminf(0.01f, m->d.mean_density);  // :560
  const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
  const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
  const float dt_min = 2 * 1.7320508075688772f / 1024;  // MIN_STEPSIZE :181-183
  const float dt_max = 2 * bound / (float)H;
  const float Hm1 = (float)(H - 1);
  const int fm = m->contract;  // nrfo_set_contract: :595-597 `ox + t * dx`, :609-614 `x * mip_rbound + 1`, :643-645 `(..) * mip_bound - x`
  uint32_t step = 0;
  if (perturb) t = mad(fm, dt_min, pcg32_first_float(n, (uint64_t)perturb), t);
  float last_t = t;
  while (t < far && step < n_step) {
    const float x = clampf(mad(fm, t, dx, ox), -bound, bound);
    const float y = clampf(mad(fm, t, dy, oy), -bound, bound);
    const float z = clampf(mad(fm, t, dz, oz), -bound, bound);
    // mip_from_pos :148-155
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    const int level = (int)fminf((float)C - 1, fmaxf(0, (float)exponent));
    const float mip_bound = fminf(exp2f((float)level), bound);
    const float mip_rbound = 1 / mip_bound;
    // `0.5 * (x*mip_rbound + 1) * H` is double arithmetic on a float operand, narrowed to float
    const int nx = (int)clampf((float)(0.5 * (double)mad(fm, x, mip_rbound, 1) * (double)H), 0.0f, Hm1);
    const int ny = (int)clampf((float)(0.5 * (double)mad(fm, y, mip_rbound, 1) * (double)H), 0.0f, Hm1);
    const int nz = (int)clampf((float)(0.5 * (double)mad(fm, z, mip_rbound, 1) * (double)H), 0.0f, Hm1);
    const uint32_t index = (uint32_t)level * H * H * H + (uint32_t)nx * H * H + (uint32_t)ny * H + (uint32_t)nz;
    const float density = grid[index];
    if (density > density_thresh) {
      xyz[3 * step + 0] = x;
      xyz[3 * step + 1] = y;
      xyz[3 * step + 2] = z;
      const float dt = clampf(t * dt_gamma, dt_min, dt_max);
      t += dt;
      delta[2 * step + 0] = dt;
      delta[2 * step + 1] = t - last_t;
      last_t = t;
      step++;
    } else {
      // (`n + 0.5f + 0.5f * sign` and `q / (H - 1) * 2 - 1` hold products by 0.5 and 2: exact, fused or not)
      const float tx = mad(fm, ((float)nx + 0.5f + 0.5f * copysignf(1.0f, dx)) / Hm1 * 2 - 1, mip_bound, -x) * rdx;
      const float ty = mad(fm, ((float)ny + 0.5f + 0.5f * copysignf(1.0f, dy)) / Hm1 * 2 - 1, mip_bound, -y) * rdy;
      const float tz = mad(fm, ((float)nz + 0.5f + 0.5f * copysignf(1.0f, dz)) / Hm1 * 2 - 1, mip_bound, -z) * rdz;
      const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
      do {
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);
        t += dt;
      } while (t < tt);
    }
  }
  return step;
}

// kernel_composite_rays for one ray, render_utils.h:679-749.  st = (ws, depth, r, g, b).
// Returns the new rays_t (-1 = dead).
// fm (nrfo_set_contract): `d += weight * t` and the three colour sums as fused multiply-adds, and `weight_sum += weight` as
// fmaf(alpha, T, weight_sum) -- the product `alpha * T` has other uses, which the NVPTX back end fuses all the same
// (enableAggressiveFMAFusion); `weight` itself stays the rounded product.
inline float composite_one(const float* sigmas, const float* rgbs, const float* deltas,
                           uint32_t n_step, float t, float* st, int fm = 0) {
  float weight_sum = st[0], dd = st[1], r = st[2], g = st[3], b = st[4];
  uint32_t step = 0;
  while (step < n_step) {
    if (deltas[2 * step] == 0) break;
    const float alpha = 1.0f - expf(-sigmas[step] * deltas[2 * step]);  // reference: __expf
    const float T = 1 - weight_sum;
    const float weight = alpha * T;
    weight_sum = fm == 1 ? fmaf(alpha, T, weight_sum) : weight_sum + weight;  // (mode 2: a product with other uses is NOT fused)
    t += deltas[2 * step + 1];
    dd = mad(fm, weight, t, dd);
    r = mad(fm, weight, rgbs[3 * step + 0], r);
    g = mad(fm, weight, rgbs[3 * step + 1], g);
    b = mad(fm, weight, rgbs[3 * step + 2], b);
    if ((double)T < 1e-4) break;  // `T < 1e-4` compares against a double literal
    step++;
  }
  st[0] = weight_sum;
  st[1] = dd;
  st[2] = r;
  st[3] = g;
  st[4] = b;
  return step < n_step ? -1.0f : t;
}

struct RayState {
  float o[3], d[3], near, far;
  float st[5];
  float t;
  uint32_t id;         // the ray's number: its pixel, py * W + px (the index of rays_o / rays_d in the reference)
  uint32_t n_emitted;  // samples the march emitted for this ray
  uint64_t hash;       // FNV-1a over the (dt, t - last_t) bits of every emitted sample: equal hashes <=> the same sample set
};
inline void note_samples(RayState& r, const float* delta, uint32_t cnt) {
  for (uint32_t k = 0; k < 2 * cnt; ++k) {
    uint32_t bits;
    std::memcpy(&bits, delta + k, 4);
    for (int b = 0; b < 4; ++b) r.hash = (r.hash ^ ((bits >> (8 * b)) & 0xffu)) * 0x100000001b3ull;
  }
  r.n_emitted += cnt;
}

// The loop of R/src/nerf_render.cu:269-338 over one group of rays.
// n_total plays the role of N (n_step = clamp(N/num_alive,1,8)).
// skip_missed (TILE64 schedule only): rays with near >= far never enter the
// alive list; they would emit no sample and die in their first composite, so
// the image is unchanged.
// fixed_n_step > 0 overrides the N/num_alive rule (PER_RAY schedule: 1).
void render_group(const nrfo_model* m, const nrf_options* opt, std::vector<RayState>& rays,
                  bool parallel, bool skip_missed, int fixed_n_step, uint64_t* n_samples, uint64_t* n_rounds) {
  const int N = (int)rays.size();
  std::vector<int> alive, next;
  alive.reserve(N);
  for (int i = 0; i < N; ++i) {
    rays[i].t = rays[i].near;   // init_step0, render_utils.h:221-239
    if (!skip_missed || rays[i].near < rays[i].far) alive.push_back(i);
  }
  next.reserve(N);
  int step = 0;
  uint64_t samples = 0, rounds = 0;
  bool first = true;
  while (true) {
    if (step >= opt->max_steps) break;  // :270
    if (!first) {  // kernel_compact_rays :394-415 (stable order here; order is immaterial)
      next.clear();
      for (int id : alive)
        if (rays[id].t >= 0) next.push_back(id);
      alive.swap(next);
    }
    first = false;
    const int num_alive = (int)alive.size();
    if (num_alive <= 0) break;  // :294
    const int n_step = fixed_n_step > 0 ? fixed_n_step : std::max(std::min(N / num_alive, 8), 1);  // :300
    uint64_t round_samples = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : round_samples) if (parallel)
    for (int a = 0; a < num_alive; ++a) {
      RayState& r = rays[alive[a]];
      float xyz[8 * 3], delta[8 * 2], sig[8], rgb[8 * 3];
      for (int k = 0; k < n_step; ++k) delta[2 * k] = delta[2 * k + 1] = 0.0f;  // deviation D-1
      // (perturb: n = the ray's position in the alive list.  The reference compacts with atomicAdd, render_utils.h:394-415: its
      //  order -- and with it every later round's random numbers -- differs from run to run; here the order is stable)
      const uint32_t cnt = march_one(m, opt->dt_gamma, r.o, r.d, r.far, r.t, (uint32_t)n_step, xyz, delta, (uint32_t)opt->perturb, (uint64_t)a);
      note_samples(r, delta, cnt);
      for (uint32_t k = 0; k < cnt; ++k) network_one(m, opt->density_scale, xyz + 3 * k, r.d, sig + k, rgb + 3 * k);
      for (uint32_t k = cnt; k < (uint32_t)n_step; ++k) sig[k] = rgb[3 * k] = rgb[3 * k + 1] = rgb[3 * k + 2] = 0.0f;
      r.t = composite_one(sig, rgb, delta, (uint32_t)n_step, r.t, r.st, m->contract);
      round_samples += cnt;
    }
    samples += round_samples;
    rounds++;
    step += n_step;  // :336
  }
  *n_samples += samples;
  *n_rounds += rounds;
}

// The PER_RAY schedule without the rounds: with n_step == 1 nothing a ray does depends on any other ray (the alive list
// only decides WHEN it is served), so every ray runs the loop of nerf_render.cu:269-338 to its own end -- march one sample,
// evaluate, composite, until it dies or max_steps iterations have passed -- under a dynamic schedule: no barrier per
// round, no serial compaction.  Bit-identical to render_group(.., fixed_n_step = 1) (tests/test_config1.py); what a CPU
// implementation of the path would do, and what bench.py times as `cpu_baseline`.  n_rounds = the rounds the global loop
// would have run = the longest ray's iteration count.
void render_rays_independent(const nrfo_model* m, const nrf_options* opt, std::vector<RayState>& rays, uint64_t* n_samples,
                             uint64_t* n_rounds) {
  const int64_t N = (int64_t)rays.size();
  uint64_t samples = 0, rounds = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+ : samples) reduction(max : rounds)
  for (int64_t i = 0; i < N; ++i) {
    RayState& r = rays[i];
    r.t = r.near;                   // init_step0
    if (!(r.near < r.far)) continue;  // never enters the alive list (skip_missed)
    uint64_t it = 0;
    while (it < (uint64_t)opt->max_steps && r.t >= 0) {
      float xyz[3], delta[2] = {0.0f, 0.0f}, sig = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f};
      // (perturb: n = the ray's own number -- its pixel, what the reference's FIRST round uses for every ray; the schedule-free
      //  definition the HIP kernel implements.  Differs from the round loop, whose n changes as other rays die)
      const uint32_t cnt = march_one(m, opt->dt_gamma, r.o, r.d, r.far, r.t, 1u, xyz, delta, (uint32_t)opt->perturb, (uint64_t)r.id);
      note_samples(r, delta, cnt);
      if (cnt) network_one(m, opt->density_scale, xyz, r.d, &sig, rgb);
      r.t = composite_one(&sig, rgb, delta, 1u, r.t, r.st, m->contract);
      samples += cnt;
      ++it;
    }
    if (it > rounds) rounds = it;
  }
  *n_samples += samples;
  *n_rounds += rounds;
}

}  // namespace

extern "C" {

int nrfo_encode_grid(const nrfo_model* m, const float* pos01, uint32_t n, uint16_t* out) {
  if (!m || !pos01 || !out) return fail(NRF_E_INVALID, "null argument");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) encode_grid_one(m, pos01 + 3 * i, out + (size_t)m->feat_width * i);
  return NRF_OK;
}

int nrfo_encode_dir(const nrfo_model* m, const float* dir01, uint32_t n, uint16_t* out) {
  if (!m || !dir01 || !out) return fail(NRF_E_INVALID, "null argument");
  for (uint32_t i = 0; i < n; ++i) encode_dir_one(m, dir01 + 3 * i, out + (size_t)m->dir_width * i);
  return NRF_OK;
}

int nrfo_mlp_forward(const nrfo_model* m, const uint16_t* feat, const uint16_t* dirfeat, uint32_t n,
                     uint16_t* out4) {
  if (!m || !feat || !dirfeat || !out4) return fail(NRF_E_INVALID, "null argument");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i)
    network_encoded_one(m, feat + (size_t)m->feat_width * i, dirfeat + (size_t)m->dir_width * i, out4 + 4 * i);
  return NRF_OK;
}

int nrfo_network(const nrfo_model* m, const float* xyz, const float* dir, uint32_t n, float* sigma,
                 float* rgb) {
  if (!m || !xyz || !dir || !sigma || !rgb) return fail(NRF_E_INVALID, "null argument");
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) network_one(m, 1.0f, xyz + 3 * i, dir + 3 * i, sigma + i, rgb + 3 * i);
  return NRF_OK;
}

int nrfo_generate_rays(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H,
                       const nrf_options* o, float* rays_o, float* rays_d, float* nears, float* fars) {
  if (!m || !cam || !pose || !o) return fail(NRF_E_INVALID, "null argument");
  float np[16];
  nrfo_nerf_matrix_to_ngp(pose, m->d.scale, np);
  const float R[9] = {np[0], np[1], np[2], np[4], np[5], np[6], np[8], np[9], np[10]};
  const float org[3] = {np[3], np[7], np[11]};
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const size_t i = (size_t)py * W + px;
      float d[3], nr, fr;
      ray_dir(R, cam, px, py, d, m->contract);
      near_far(m->d.aabb, org, d, o->min_near, &nr, &fr);
      if (rays_o) { rays_o[3 * i] = org[0]; rays_o[3 * i + 1] = org[1]; rays_o[3 * i + 2] = org[2]; }
      if (rays_d) { rays_d[3 * i] = d[0]; rays_d[3 * i + 1] = d[1]; rays_d[3 * i + 2] = d[2]; }
      if (nears) nears[i] = nr;
      if (fars) fars[i] = fr;
    }
  return NRF_OK;
}

int nrfo_march(const nrfo_model* m, const nrf_options* o, const float* rays_o, const float* rays_d,
               const float* rays_t, const float* fars, uint32_t n, uint32_t n_step, float* xyzs,
               float* dirs, float* deltas) {
  if (!m || !o || n_step < 1 || n_step > 8) return fail(NRF_E_INVALID, "bad argument");
  for (uint32_t i = 0; i < n; ++i) {
    float* xyz = xyzs + (size_t)i * n_step * 3;
    float* dl = deltas + (size_t)i * n_step * 2;
    float* dr = dirs + (size_t)i * n_step * 3;
    std::fill(xyz, xyz + n_step * 3, 0.0f);
    std::fill(dl, dl + n_step * 2, 0.0f);
    std::fill(dr, dr + n_step * 3, 0.0f);
    const uint32_t cnt = march_one(m, o->dt_gamma, rays_o + 3 * i, rays_d + 3 * i, fars[i], rays_t[i], n_step, xyz, dl, (uint32_t)o->perturb, (uint64_t)i);
    for (uint32_t k = 0; k < cnt; ++k)
      for (int c = 0; c < 3; ++c) dr[3 * k + c] = rays_d[3 * i + c];
  }
  return NRF_OK;
}

// The trip STARTS of kernel_march_rays (render_utils.h:593-653) along one ray of an EMPTY volume: every trip takes the
// `else` branch (hop to the next voxel, :639-651), and the t at which each trip begins is recorded.  This is what decides
// where the march tests the occupancy grid; tests/test_barrier_lemma.py checks the HIP path's barrier fast-forward
// (nerf-cuda_amd/csrc/nrf_device.h) against it.  Returns the number of trips (the first `cap` starts are stored).
uint32_t nrfo_march_trip_starts(float bound, uint32_t C, uint32_t H, float dt_gamma, const float o[3], const float d[3], float t,
                                float far, float* starts, uint32_t cap) {
  const float ox = o[0], oy = o[1], oz = o[2], dx = d[0], dy = d[1], dz = d[2];
  const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
  const float dt_min = 2 * 1.7320508075688772f / 1024;
  const float dt_max = 2 * bound / (float)H;
  const float Hm1 = (float)(H - 1);
  uint32_t n = 0;
  while (t < far) {
    if (n < cap) starts[n] = t;
    ++n;
    const float x = clampf(ox + t * dx, -bound, bound);
    const float y = clampf(oy + t * dy, -bound, bound);
    const float z = clampf(oz + t * dz, -bound, bound);
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    const int level = (int)fminf((float)C - 1, fmaxf(0, (float)exponent));
    const float mip_bound = fminf(exp2f((float)level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, Hm1);
    const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, Hm1);
    const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, Hm1);
    const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, dx)) / Hm1 * 2 - 1) * mip_bound - x) * rdx;
    const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, dy)) / Hm1 * 2 - 1) * mip_bound - y) * rdy;
    const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, dz)) / Hm1 * 2 - 1) * mip_bound - z) * rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
      const float dt = clampf(t * dt_gamma, dt_min, dt_max);
      t += dt;
    } while (t < tt);
  }
  return n;
}

int nrfo_composite(const float* sigmas, const float* rgbs, const float* deltas, uint32_t n,
                   uint32_t n_step, float* rays_t, float* state) {
  for (uint32_t i = 0; i < n; ++i)
    rays_t[i] = composite_one(sigmas + (size_t)i * n_step, rgbs + (size_t)i * n_step * 3,
                              deltas + (size_t)i * n_step * 2, n_step, rays_t[i], state + 5 * (size_t)i);
  return NRF_OK;
}

}  // extern "C"
namespace {
int render_impl(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H, const nrf_options* o, int schedule,
                int n_threads, float* rgba, float* depth, nrf_stats* stats, uint32_t* ray_samples, uint64_t* ray_hash) {
  if (!m || !cam || !pose || !o || !rgba || !depth || W <= 0 || H <= 0) return fail(NRF_E_INVALID, "bad argument");
  if (o->perturb < 0) return fail(NRF_E_INVALID, "perturb must be >= 0 (0: off; > 0: the random seed, render_utils.h:550)");
#ifdef _OPENMP
  const int saved = omp_get_max_threads();
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
  float np[16];
  nrfo_nerf_matrix_to_ngp(pose, m->d.scale, np);
  const float R[9] = {np[0], np[1], np[2], np[4], np[5], np[6], np[8], np[9], np[10]};
  const float org[3] = {np[3], np[7], np[11]};
  uint64_t n_samples = 0, n_rounds = 0;

  auto init_ray = [&](RayState& r, int px, int py) {
    r.o[0] = org[0]; r.o[1] = org[1]; r.o[2] = org[2];
    ray_dir(R, cam, px, py, r.d, m->contract);
    near_far(m->d.aabb, r.o, r.d, o->min_near, &r.near, &r.far);
    for (float& v : r.st) v = 0.0f;  // zero fills nerf_render.cu:262-264
    r.id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    r.n_emitted = 0;
    r.hash = 0xcbf29ce484222325ull;
  };
  // get_image_and_depth, render_utils.h:257-264; deviation D-3: a ray that
  // misses the aabb (near == far) gets depth 0 instead of 0/0.
  auto finish = [&](const RayState& r, int px, int py) {
    const size_t i = (size_t)py * W + px;
    const float bg = o->bg_color;
    rgba[4 * i + 0] = mad(m->contract, 1 - r.st[0], bg, r.st[2]);  // image + (1 - weights_sum) * bg_color
    rgba[4 * i + 1] = mad(m->contract, 1 - r.st[0], bg, r.st[3]);
    rgba[4 * i + 2] = mad(m->contract, 1 - r.st[0], bg, r.st[4]);
    rgba[4 * i + 3] = r.st[0];
    if (ray_samples) ray_samples[i] = r.n_emitted;
    if (ray_hash) ray_hash[i] = r.hash;
    const float span = r.far - r.near;
    depth[i] = span > 0.0f ? fmaxf(r.st[1] - r.near, 0.0f) / span : 0.0f;
  };

  const bool per_ray_rounds = schedule == -NRFO_SCHED_PER_RAY;  // (nrfo_render_per_ray_rounds)
  if (per_ray_rounds) schedule = NRFO_SCHED_PER_RAY;
  if (schedule == NRFO_SCHED_REFERENCE || schedule == NRFO_SCHED_PER_RAY) {
    std::vector<RayState> rays((size_t)W * H);
#pragma omp parallel for schedule(static)
    for (int py = 0; py < H; ++py)
      for (int px = 0; px < W; ++px) init_ray(rays[(size_t)py * W + px], px, py);
    if (per_ray_rounds) render_group(m, o, rays, true, true, 1, &n_samples, &n_rounds);
    else if (schedule == NRFO_SCHED_PER_RAY) render_rays_independent(m, o, rays, &n_samples, &n_rounds);
    else render_group(m, o, rays, true, false, 0, &n_samples, &n_rounds);
#pragma omp parallel for schedule(static)
    for (int py = 0; py < H; ++py)
      for (int px = 0; px < W; ++px) finish(rays[(size_t)py * W + px], px, py);
  } else {
    const int tx_n = (W + 7) / 8, ty_n = (H + 7) / 8;
    uint64_t s_acc = 0, r_acc = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : s_acc, r_acc)
    for (int tile = 0; tile < tx_n * ty_n; ++tile) {
      const int tx = tile % tx_n, ty = tile / tx_n;
      // a wave tile always has 64 lanes; lanes outside the image are rays that never start
      std::vector<RayState> rays(64);
      for (int l = 0; l < 64; ++l) {
        const int px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (px < W && py < H) {
          init_ray(rays[l], px, py);
        } else {
          std::memset(&rays[l], 0, sizeof(RayState));
          rays[l].near = rays[l].far = FLT_MAX;
        }
      }
      uint64_t s = 0, r = 0;
      render_group(m, o, rays, false, true, 0, &s, &r);
      s_acc += s;
      r_acc += r;
      for (int l = 0; l < 64; ++l) {
        const int px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (px < W && py < H) finish(rays[l], px, py);
      }
    }
    n_samples = s_acc;
    n_rounds = r_acc;
  }
  if (stats) {
    stats->n_rays = (uint64_t)W * H;
    stats->n_samples = n_samples;
    stats->n_rounds = n_rounds;
    stats->n_network_evals = n_samples;  // D-5: the oracle evaluates live samples only, no tile padding
    stats->render_ms = 0.0f;
    stats->shader_clock_mhz = 0.0f;
    // the per-ray schedule emits exactly the samples a ray composites; the other schedules (n_step > 1) also evaluate
    // samples behind a ray's terminating one and do not count the composited ones apart: 0 = not counted
    stats->n_composited = schedule == NRFO_SCHED_PER_RAY ? n_samples : 0;
  }
#ifdef _OPENMP
  omp_set_num_threads(saved);
#endif
  return NRF_OK;
}
}  // namespace
extern "C" {

int nrfo_render(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H, const nrf_options* o, int schedule,
                int n_threads, float* rgba, float* depth, nrf_stats* stats) {
  return render_impl(m, cam, pose, W, H, o, schedule, n_threads, rgba, depth, stats, nullptr, nullptr);
}

// The same render, also returning per ray (row-major [H][W]) the number of samples its march emitted and a hash of their
// (dt, t - last_t) bits: two renders whose hashes agree for a ray took the same samples along it.
int nrfo_render_rays(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H, const nrf_options* o, int schedule,
                     int n_threads, float* rgba, float* depth, nrf_stats* stats, uint32_t* ray_samples, uint64_t* ray_hash) {
  return render_impl(m, cam, pose, W, H, o, schedule, n_threads, rgba, depth, stats, ray_samples, ray_hash);
}

// the PER_RAY schedule through the global round loop (render_group with n_step fixed to 1), as rounds 1-4 ran it: kept as the
// cross-check of render_rays_independent
int nrfo_render_per_ray_rounds(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H, const nrf_options* o,
                               float* rgba, float* depth, nrf_stats* stats) {
  return render_impl(m, cam, pose, W, H, o, -NRFO_SCHED_PER_RAY, 0, rgba, depth, stats, nullptr, nullptr);
}

// NerfRender::generate_density_grid, R/src/nerf_render.cu:388-429, completed as nerfhip.h nrf_generate_density_grid
// documents (the reference's version is dead: its density query is commented out at :415): init_xyzs
// (render_utils.h:91-108), dd_scale by bound_c - bound_c/H (nerf_render.cu:409-413), the density of the network at
// that position (the render path's own affine map + encoding + density MLP + sigma activation, fp16), dd_scale by
// 0.001691 (:417), dg_update (render_utils.h:120-128) n_iterations times from 1/64 (:393).
int nrfo_density_grid(const nrfo_model* m, int n_iterations, float decay, float* grid, float* mean_density) {
  if (!m || !grid || n_iterations < 1) return fail(NRF_E_INVALID, "bad argument");
  const uint32_t H = m->d.density_grid_size, C = m->d.cascade;
  const int64_t n = (int64_t)H * H * H;
  const float step = 2.f / (float)(H - 1);
  for (uint32_t cas = 0; cas < C; ++cas) {
    const float bound = (float)(1u << cas) < m->d.bound ? (float)(1u << cas) : m->d.bound;
    const float half_grid_size = bound / (float)H;
    const float k = bound - half_grid_size;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t id[3] = {(uint32_t)(i / ((int64_t)H * H)), (uint32_t)((i % ((int64_t)H * H)) / H), (uint32_t)(i % H)};
      float xyz[3];
      for (int a = 0; a < 3; ++a) {
        float v = step * (float)id[a];
        v = -1.f + v;
        xyz[a] = k * v;
      }
      const float dir[3] = {0.0f, 0.0f, 1.0f};
      float sigma, rgb[3];
      network_one(m, 1.0f, xyz, dir, &sigma, rgb);
      const float tmp = 0.001691f * sigma;
      float g = 1.0f / 64;
      for (int it = 0; it < n_iterations; ++it) {
        if (g >= 0) {
          const float gd = g * decay;
          g = gd > tmp ? gd : tmp;
        }
      }
      grid[(size_t)cas * n + i] = g;
    }
  }
  double sum = 0.0;
  for (int64_t i = 0; i < n * C; ++i) sum += grid[i] > 0.0f ? (double)grid[i] : 0.0;
  if (mean_density) *mean_density = (float)(sum / (double)(n * C));
  return NRF_OK;
}

// R/src/nerf_render.cu:352-359: (unsigned char)(255.0 * x), here saturating and
// NaN -> 0 (deviation D-2).
void nrfo_quantize_u8(const float* rgba, const float* depth, int n_px, uint8_t* rgb, uint8_t* depth_u8) {
  auto q = [](float v) -> uint8_t {
    const double s = 255.0 * (double)v;
    if (!(s > 0.0)) return 0;
    if (s >= 255.0) return 255;
    return (uint8_t)s;
  };
  for (int i = 0; i < n_px; ++i) {
    if (rgb) {
      rgb[3 * i + 0] = q(rgba[4 * i + 0]);
      rgb[3 * i + 1] = q(rgba[4 * i + 1]);
      rgb[3 * i + 2] = q(rgba[4 * i + 2]);
    }
    if (depth_u8) depth_u8[i] = q(depth[i]);
  }
}

// ---- render buffer chain -----------------------------------------------------------------
// colour helpers: R/include/nerf-cuda/common_device.cuh:38-60
static inline float srgb_to_linear1(float srgb) {
  return srgb <= 0.04045f ? srgb / 12.92f : std::pow((srgb + 0.055f) / 1.055f, 2.4f);
}
static inline float linear_to_srgb1(float linear) {
  return linear < 0.0031308f ? 12.92f * linear : 1.055f * std::pow(linear, 0.41666f) - 0.055f;
}

// R/src/render_buffer.cu:224-259
void nrfo_rb_accumulate(const float* frame, float* accum, int n, float sample_count, int color_space) {
  for (int i = 0; i < n; ++i) {
    float color[4] = {frame[4 * i], frame[4 * i + 1], frame[4 * i + 2], frame[4 * i + 3]};
    float* tmp = accum + 4 * (size_t)i;
    if (color_space == NRF_CS_VISPOSNEG) {
      const float val = color[0] - color[1];
      float tmp_val = tmp[0] - tmp[1];
      tmp_val = (tmp_val * sample_count + val) / (sample_count + 1);
      tmp[0] = fmaxf(tmp_val, 0.0f);
      tmp[1] = fmaxf(-tmp_val, 0.0f);
    } else {
      if (color_space == NRF_CS_SRGB)
        for (int k = 0; k < 3; ++k) color[k] = linear_to_srgb1(color[k]);
      for (int k = 0; k < 3; ++k) tmp[k] = (tmp[k] * sample_count + color[k]) / (sample_count + 1);
    }
    tmp[3] = (tmp[3] * sample_count + color[3]) / (sample_count + 1);
  }
}

// R/src/render_buffer.cu:261-318
static void tonemap_curve(float c[3], int curve) {
  if (curve == NRF_TM_IDENTITY) return;
  for (int i = 0; i < 3; ++i) c[i] = fmaxf(c[i], 0.f);
  float k0, k1, k2, k3, k4, k5;
  if (curve == NRF_TM_ACES) {
    k0 = 0.6f * 0.6f * 2.51f; k1 = 0.6f * 0.03f; k2 = 0.0f;
    k3 = 0.6f * 0.6f * 2.43f; k4 = 0.6f * 0.59f; k5 = 0.14f;
  } else if (curve == NRF_TM_HABLE) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    k0 = A * F - A * E; k1 = C * B * F - B * E; k2 = 0.0f;
    k3 = A * F; k4 = B * F; k5 = D * F * F;
    const float W = 11.2f;
    const float nom = k0 * (W * W) + k1 * W + k2;
    const float denom = k3 * (W * W) + k4 * W + k5;
    const float white_scale = denom / nom;
    k0 = 4.0f * k0 * white_scale; k1 = 2.0f * k1 * white_scale; k2 = k2 * white_scale;
    k3 = 4.0f * k3; k4 = 2.0f * k4;
  } else {
    const float Y = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];
    const float s = 1.f / (Y + 1.0f);
    for (int i = 0; i < 3; ++i) c[i] = c[i] * s;
    return;
  }
  for (int i = 0; i < 3; ++i) {
    const float sq = c[i] * c[i];
    c[i] = (sq * k0 + k1 * c[i] + k2) / (k3 * sq + k4 * c[i] + k5);
  }
}

// R/src/render_buffer.cu:320-342 and :529-556
void nrfo_rb_tonemap(const float* accum, float* surface, int n, float exposure, const float bg_in[4], int color_space,
                     int output_color_space, int curve, int clamp_output) {
  float bg[4] = {bg_in[0], bg_in[1], bg_in[2], bg_in[3]};
  if (color_space != NRF_CS_SRGB)
    for (int k = 0; k < 3; ++k) bg[k] = srgb_to_linear1(bg[k]);
  const float gain = std::pow(2.0f, exposure);
  for (int i = 0; i < n; ++i) {
    const float* color = accum + 4 * (size_t)i;
    const float weight = (1 - color[3]) * bg[3];
    float c[3] = {color[0] + bg[0] * weight, color[1] + bg[1] * weight, color[2] + bg[2] * weight};
    const float a = color[3] + weight;
    if (color_space == NRF_CS_SRGB)
      for (int k = 0; k < 3; ++k) c[k] = srgb_to_linear1(c[k]);
    for (int k = 0; k < 3; ++k) c[k] *= gain;
    tonemap_curve(c, curve);
    if (output_color_space == NRF_CS_SRGB)
      for (int k = 0; k < 3; ++k) c[k] = linear_to_srgb1(c[k]);
    float o[4] = {c[0], c[1], c[2], a};
    if (clamp_output)
      for (int k = 0; k < 4; ++k) o[k] = fminf(fmaxf(o[k], 0.f), 1.f);
    for (int k = 0; k < 4; ++k) surface[4 * (size_t)i + k] = o[k];
  }
}

}  // extern "C"
