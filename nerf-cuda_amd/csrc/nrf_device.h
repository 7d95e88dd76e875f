// nrf_device.h -- device-side building blocks of the gfx950 render path.
//
// Everything here is written for CDNA4 only: 64-lane wavefronts, MFMA
// 16x16x32 f16 tiles, ballot/mbcnt compaction.  Compiled with
// -ffp-contract=off so that every fp32 operation of ray generation, marching
// and the hash-grid address math is individually rounded -- the same contract
// as the CPU oracle, which makes those stages bit-exact against it.
//
// Reference behaviour restated here (R/ = reference repo, T/ = its tiny-cuda-nn):
//   raygen        R/include/nerf-cuda/render_utils.h:31-66
//   near/far      R/include/nerf-cuda/render_utils.h:338-392
//   march         R/include/nerf-cuda/render_utils.h:524-655
//   hash grid     T/include/tiny-cuda-nn/encodings/grid.h:81-117,139-268
//   SH / freq     T/include/tiny-cuda-nn/encodings/spherical_harmonics.h:46-96, frequency.h:46-93
//   fused MLP     T/src/fully_fused_mlp.cu:500-558 (semantics: y = act(W x), no bias)
//   composite     R/include/nerf-cuda/render_utils.h:658-751
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerfhip.h"

namespace nrf {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// index modes of one grid level (T/.../grid.h:100-117)
enum : uint32_t {
  LV_DENSE = 0,    // x + y*res + z*res^2, then one conditional subtract (index < 2*size always)
  LV_HASH_POW2 = 1,  // fast_hash & (size-1)
  LV_GENERIC = 2,  // literal restatement with integer modulo (tiled grids, non-pow2 hash sizes)
  LV_ADD_POW2 = 3  // additive index on a power-of-two table: (x + y * m1 + z * m2) & (size - 1) with the multipliers grid_index
                   // ends up with in uint32 (grid.h:106-114) -- m1 = res, m2 = res * res mod 2^32, or 0 for a term the stride
                   // loop skips.  Tiled grids, and what becomes of a HASH level whose stride overflows: res^3 >= 2^32
                   // wraps `stride` below the table size, `hashmap_size < stride` is false, and the level is indexed
                   // additively, no hash -- res 65536 (instant-ngp at aabb_scale 32: m2 = 0, a 2-D level), and the finest
                   // levels (res 1626 ... 2048) of every table of 2^22 entries or more (m2 = res^2 is a multiple of... or
                   // collides with ... the table size: for res 2048 on 2^22 entries the z term vanishes as well).
};

struct LevelParams {
  float scale;      // exp2f(l*log2f(b))*Nmin - 1, computed on the host
  uint32_t res;     // ceil(scale)+1
  uint32_t offset;  // first entry of the level (in half2 entries)
  uint32_t size;    // entries in the level ("hashmap_size")
  uint32_t mode;    // LV_*
  uint32_t hashed;  // grid_type == Hash (for LV_GENERIC)
  // per-level constants of the byte-offset index arithmetic (level_gather), precomputed on the host:
  uint32_t off_b;   // offset << 2
  uint32_t my_b;    // y stride in bytes: LV_DENSE res << 2, LV_HASH_POW2 2654435761 << 2
  uint32_t mz_b;    // z stride in bytes: LV_DENSE (res * res) << 2, LV_HASH_POW2 805459861 << 2
  uint32_t mask_b;  // LV_HASH_POW2 (size - 1) << 2, else 0xffffffff
  // cell-major quad copy of the level (round 6; level_gather_quad): q_off_b != 0 iff the level has one
  uint32_t q_off_b;  // byte offset of the level's first quad (a multiple of 16) -- in units of 16 bytes for a FAR level (below)
  uint32_t q_my_b;   // res << 4            (far: res)
  uint32_t q_mz_b;   // (res * res) << 4    (far: res * res); all below 2^24 (v_mad_u32_u24)
  uint32_t q_max;    // res - 1: the largest cell coordinate (far levels clamp to it: their gathers have no range check)
  uint32_t pad0, pad1;
};
static_assert(sizeof(LevelParams) == 64, "LevelParams layout");

// MFMA weight fragments, packed on the host (nrf_model.cpp: pack_fragments):
// one fragment = 64 lanes x 8 halves (1 KiB), lane l, element j holds
// W[16*m + (l&15)][kmap(s, l>>4, j)].
enum : int {
  FRAG_D0 = 0,   // density 32->64 : m = 0..3
  FRAG_D1 = 4,   // density 64->16 : s = 0..1
  FRAG_R0 = 6,   // rgb 32->64     : m = 0..3
  FRAG_R1 = 10,  // rgb 64->64     : (m,s) -> 10 + 2*m + s
  FRAG_R2 = 18,  // rgb 64->16     : s = 0..1
  N_FRAGS = 20,       // what render_kernel / network_kernel keep in LDS
  FRAG_D0_NATURAL = 20,  // density 32->64 again with the natural K order (k = 8g + j): mlp_forward_kernel reads its
                         // [n][32] input rows as one 16-byte load per lane
  N_FRAGS_ALL = 24,
  // "wide" instance (direction encodings of 32..80 values: an rgb input of up to 96 columns = RK_WIDE K steps):
  FRAG_R0X = 24,         // rgb first layer, K steps s = 1, 2: fragment FRAG_R0X + 4 (s - 1) + m, natural K order
                         // (column 32 s + 8 g + j, i.e. direction entry 32 s - 16 + 8 g + j); zero beyond the real width
  N_FRAGS_WIDE_ALL = 32
};
constexpr int RK_WIDE = 3;  // K steps of the wide instance's first rgb layer
// The same fragment order for the other widths of tcnn's FullyFusedMLP (16 / 32 / 128 neurons, T/src/fully_fused_mlp.cu:700-725;
// one hidden layer in the density MLP, two in the rgb MLP as in base.json): a layer of W outputs is MT = W / 16 fragments per
// K step, a layer of W inputs KS = ceil(W / 32) K steps (W = 16: one step whose upper half is zero).  For W = 64 these are
// the FRAG_* values above.
template <int W>
struct MlpShape {
  static_assert(W == 16 || W == 32 || W == 64 || W == 128, "FullyFusedMLP widths");
  static constexpr int MT = W / 16, KS = (W + 31) / 32;
  static constexpr int D0 = 0, D1 = MT, R0 = MT + KS, R1 = 2 * MT + KS, R2 = 2 * MT + KS + MT * KS, N = 2 * MT + 2 * KS + MT * KS;
};
static_assert(MlpShape<64>::D1 == FRAG_D1 && MlpShape<64>::R0 == FRAG_R0 && MlpShape<64>::R1 == FRAG_R1 && MlpShape<64>::R2 == FRAG_R2 &&
                  MlpShape<64>::N == N_FRAGS, "MlpShape<64> is the base.json layout");

// DEPTH instance (64 neurons, any number of hidden layers in either MLP, tcnn's FullyFusedMLP n_hidden_layers): the layers of
// base.json's shape plus a RUNTIME number of 64 -> 64 layers.  Fragment order in LDS / DevModel::wfrag_hot:
//   D0 [64][32] m = 0..3 | D1 [16][64] s = 0..1 | R0 [64][32] m = 0..3 | R2 [16][64] s = 0..1 | the density MLP's 64 -> 64 layers
//   (8 fragments each: 2 m + s) | the rgb MLP's 64 -> 64 layers
enum : int { DF_D0 = 0, DF_D1 = 4, DF_R0 = 6, DF_R2 = 10, DF_WW = 12, DEPTH_MAX_WW = 5, DEPTH_FRAGS = DF_WW + 8 * DEPTH_MAX_WW };

constexpr uint32_t HOT_WIDTH_ACT = 65;  // DevModel::hot_width of the NET_ACT instance (64 neurons, runtime hidden activations)
struct DevModel {
  const uint32_t* grid;      // half2 entries
  const uint32_t* occ_bits;  // 1 bit per density-grid cell: grid[cell] > min(0.01, mean_density)
  const uint32_t* occ_coarse;  // OR over 4x4x4 cell blocks, [C][(H/4)^3] bits; nullptr if H % 4 != 0
  const float* cell_bound;     // [C][H+1] cell-boundary table (see march_next)
  uint32_t grid_bytes;         // size of the device hash table (< 4 GiB): num_records of its buffer resource
  const uint32_t* occ_dilated;  // [C][dilated_level_words]: coarse cells within one density cell of an occupied density cell
  uint32_t dilated_level_words;
  uint32_t pos_w_pow2;         // pos_w = 1 / (2 bound) is a power of two: pos_w * x + 0.5 is then ONE fma (the product is exact)
  const uint4* wfrag;        // hot instance: N_FRAGS_ALL * 64 uint4; generic instance: the fragments GenModel::layer names
  const LevelParams* lv;     // 16 entries (device memory)
  float aabb[6];
  float occ_box[6];  // world-space box around every occupied cell, inflated by 2 cells; min > max when nothing is occupied
  float bound;
  float rbound;  // 1.0f / bound
  float pos_w;   // (float)(1.0/(2*bound)), R/src/nerf_render.cu:311-312
  uint32_t cascade;
  uint32_t H;
  uint32_t n_levels;
  uint32_t dir_encoding, sh_degree, n_frequencies;
  uint32_t density_activation, density_output_activation, sigma_activation;
  uint32_t rgb_activation, rgb_output_activation;
  uint32_t uni_modes;    // 2 bits per unrolled step jl = 0..3 of the fused kernel (levels 4*jl + g): 0 mixed, 1 all dense,
                         // 2 all power-of-two hashed (host: nrf_load_model)
  uint32_t generic;      // 0: the shape of the reference's base.json (L = 16, F = 2, 64 neurons, 1 + 2 hidden layers, a
                         // 16-wide direction encoding -- or a Frequency encoding of up to 80 values: `wide` --, hidden
                         // ReLU / density output None / sigma Exponential / rgb output None or Sigmoid, linear
                         // interpolation, every level dense, power-of-two hashed or LV_ADD_POW2): the register-resident
                         // instance (this file);
                         // 1: everything else: the generic instance (nrf_generic.h), described by `gen`
  const struct GenModel* gen;  // device memory; nullptr unless generic
  uint32_t gen_wave_bytes;     // generic instance: LDS bytes per wave of the direction rows + activation rows
  uint32_t wide;               // register-resident instance with a 32..80-wide Frequency direction encoding: the first rgb
                               // layer takes RK_WIDE K steps, the extra direction entries are evaluated in-lane per sample
  uint32_t coarse_shift;    // 2 or 0
  uint32_t lds_coarse_words;  // words of occ_coarse staged in LDS by render_kernel (0: read it from global)
  uint32_t lds_ctab_floats;   // floats of cell_bound staged in LDS (0: read it from global)
  uint32_t lds_dilated_words;  // words of occ_dilated that fit the (not yet used) weight area of LDS during ray setup (0: global)
  uint32_t persistent;      // 1: render_persistent_kernel renders this model (every march table fits in LDS beside its waves)
  uint32_t persist_waves;   // waves of its workgroup (16 hot, 12 wide, 12 or 8 generic)
  uint32_t gen_frag_bytes;  // generic instance: bytes of its weight fragments
  uint32_t gen_weights_lds; // generic instance, persistent kernel: the fragments are staged in LDS (they fit beside rows and tables)
  uint32_t n_cus;           // compute units of the device: workgroups of the persistent kernel
  // a model of the base.json SHAPE with 16 / 32 / 128 neurons: `generic` is set (stage entry points, the per-strip kernel and
  // the density-grid generation run the generic instance), but its frames are rendered by a register-resident instance of
  // the persistent kernel of that width, from fragments in the MlpShape<width> order
  uint32_t hot_width;       // 0, 16, 32 or 128 -- or 64: the DEPTH instance (64 neurons, other numbers of hidden layers: depth_xd / depth_xr)
                            // -- or 65 (HOT_WIDTH_ACT): 64 neurons with hidden activations other than ReLU, the NET_ACT instance
  uint32_t depth_xd, depth_xr;  // hot_width == 64: 64 -> 64 layers of the density MLP (hidden layers - 1) and of the rgb MLP (hidden layers - 1)
  const uint4* wfrag_hot;   // MlpShape<hot_width>::N * 64 uint4; wide_sh: the wide layout (N_FRAGS_WIDE_ALL fragments)
  uint32_t hot_grid;        // 0, or F = 1 / 2 / 4 / 8 (F = 1: round 5): a grid other than base.json's 16 x 2 -- fewer than 16 levels at F = 2, F = 4 / 8 with up to 32
                            // features in all, Linear or Smoothstep -- in front of base.json's MLPs: the register-resident GRID instance
                            // (NET_GRID2 / 4 / 8, persistent kernel only; fragments in wfrag_hot with that grid's K order)
  uint32_t grid_smooth;     // the grid interpolates with Smoothstep (GRID instances)
  uint32_t grid_nearest;    // InterpolationType::Nearest (grid.h:215-232): the entry at floor(pos), no weights -- ONE gather per level (GRID instances)
  uint32_t wide_sh;         // SphericalHarmonics of degree 5..8 on the base.json shape: NET_WIDE_SH renders the frames (persistent kernel)
  uint32_t dir_w;           // padded width of the direction encoding (16 .. 80)
  uint32_t quad_far;        // bit jl: step jl's four levels have FAR quad copies (beyond a buffer resource's 4 GiB: level_gather_quad_far)
  uint32_t quad_mask;       // bit l: level l is gathered from its cell-major quad copy (level_gather_quad); granted four levels -- one unrolled
                            // step jl of the fused kernel, all of its lane groups -- at a time (nrf_load_model)
};

// One camera of a batched launch (nrf_render_views): what differs between the views of a batch.
struct ViewParams {
  float R[9];    // rotation of nerf_matrix_to_ngp(pose)
  float org[3];  // translation
  float cam[4];  // fl_x, fl_y, cx, cy
  int roi[4];    // x0, y0, x1, y1 (pixels, inclusive): no ray outside this rectangle enters the box of occupied
                 // cells (host: conservative projection of its corners, nrf_api.hip view_roi); x1 < x0: empty
  // persistent kernel: the local tiles [k_lo, k_hi) (multiples of 4 = whole strips) cover the q_rows strip rows (from row q_row0) the
  // rectangle touches; those rows are units [q_begin, q_begin + q_rows) of the launch's work queues; the view's other
  // tiles are background and are filled without the queues
  int k_lo, k_hi, q_begin, q_rows, q_row0;
};
// Statistics counters: COUNTER_SLOTS copies of 16 x u64 (one 128-byte line each); a workgroup adds to copy
// blockIdx % COUNTER_SLOTS.  Device-scope atomics on ONE address serialise at ~12 ns each across the 8 XCDs:
// two of them per wave made an all-background 1080p frame cost 0.79 ms.
constexpr int COUNTER_SLOTS = 64;
constexpr int COUNTER_BYTES = COUNTER_SLOTS * 16 * 8;
constexpr int MAX_VIEWS = 128; // == NRF_MAX_VIEWS: views of one render launch (by-value kernel argument: 12.8 KB; the kernarg
                               // segment takes it -- a 16 KB by-value struct was tried on gfx950)
struct ViewBatch {
  ViewParams v[MAX_VIEWS];
  int n_views;
  int blocks_per_view;                 // workgroups per view: block b renders view b / blocks_per_view
  int q_total;                         // persistent kernel: units of the launch (strip rows; sharded: local strips), all views
  int n_classes, class_cols;           // its work queues: class c holds, for every unit, the columns c, c + n_classes, ... < class_cols
  unsigned long long view_stride_px;   // pixels between consecutive views in the output planes
};

enum : int { OUT_F32 = 0, OUT_RGBD8 = 1, OUT_U8 = 2 };
struct FrameParams {
  float R[9];    // view 0 (stage kernels): rotation of nerf_matrix_to_ngp(pose)
  float org[3];  // translation
  float cam[4];  // fl_x, fl_y, cx, cy
  int W, H;
  int tiles_x, tiles_y;
  int shard_index, shard_count;
  int n_local_tiles;
  int tile_major;
  float bg_color, min_near, dt_gamma, density_scale;
  int max_steps;
  int march_budget;  // cell trips a lane may spend per round (tuning knob, default 256)
  int queue_classes;  // persistent kernel, unsharded frames: work queues (8: one per XCD; 1: a single queue); 0: default
  int out_mode;     // OUT_F32: float planes; OUT_RGBD8: packed 8-bit pixels (r | g << 8 | b << 16 | depth << 24) passed as the depth
                    // plane; OUT_U8: the reference's host Image layout -- rgb u8 [px][3] passed as the rgba plane, depth u8 [px]
  int skip_outside; // persistent kernel: the tiles outside the strip rows a view's region of interest touches are NOT written
                    // (the host-frame path fills those rows of its pinned buffer itself and copies only the other rows)
  int centre_out;   // persistent kernel: a view's strip rows are queued from the middle of its region of interest outwards
  // persistent kernel, OUT_U8 host frames: progress reporting, so that the host can start copying a frame's finished rows
  // while the rest still renders.  prog_done [n_views][tiles_y] (device, zeroed per call): tiles of that strip row whose
  // pixels have been written (write-through, acknowledged); the wave that completes a row stores prog_epoch into
  // prog_flags [n_views][tiles_y] (pinned host memory).  nullptr: off.
  int prog_epoch;
  int fast_interp;  // nrf_options::fast_interp: the FAST instances of the persistent register-resident kernel / encode_grid_kernel
  int tail_split;   // persistent kernel: 1 = waves that find the queues empty take rays off the rendering waves of their
                    // workgroup (tail splitting, nrf_render.h); 0 = they leave (A/B runs: NRF_TAIL_SPLIT=0)
  int march_ff;     // 1 = a ray steps straight to its last barrier plane ahead of t_skip (fast_forward_to_barrier); 0 = every
                    // trip of that stretch is simulated (A/B runs and the equality tests: NRF_MARCH_FF=0)
  int perturb;      // nrf_options.perturb (render_utils.h:550, 585-589): > 0 = the seed of the march's per-ray shift of t (the host turns
                    // the barrier fast-forward off with it: march_ff == 0)
  int sample_cap;   // the samples a ray may queue per round shrink with its transmittance T: fewer samples evaluated behind a ray's
                    // terminating one, frames unchanged (per-ray semantics).  2 (default) = what the ray still needs to reach
                    // T < 1e-4 if every sample halves T (clamp(exponent(T) + 13, 1, 8): never short of the need unless alpha > 0.5);
                    // 1 = 8 / 4 / 2 / 1 for T >= 0.4 / 0.1 / 0.02 / below; 0 = always up to 8 (A/B runs: NRF_SAMPLE_CAP)
  unsigned* prog_done;
  unsigned* prog_flags;
  // persistent kernel: the launch's queue order (plan_sort_kernel): entry [class offset + i] = the queue position the i-th pull of
  // that class renders -- a permutation of each class's positions, heaviest strips first.  nullptr: positions in order.
  const unsigned* plan_order;
};

// ------------------------------------------------------------------ misc ----
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ uint32_t h2_bits(half2_t v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ half2_t bits_h2(uint32_t v) { return __builtin_bit_cast(half2_t, v); }
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
  half2_t v;
  v.x = (half_t)a;  // v_cvt_f16_f32: round-to-nearest-even
  v.y = (half_t)b;
  return h2_bits(v);
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// fp32 -> fp16 of a value that was ROUNDED to fp32 first (what `(half)(float expression)` means in the reference).
// Without the empty asm the compiler folds a preceding fmul / fadd into v_fma_mixlo_f16, which rounds the exact
// result once: 1 in ~130 000 SH coefficients then differs from the two-step rounding by one fp16 ulp
// (scripts/sh_probe.py found 240 of 32 M before this helper existed).
__device__ __forceinline__ half_t f2h_rne(float v) {
  asm("" : "+v"(v));
  return (half_t)v;
}

// tcnn activations on an fp32 pre-activation (T/include/tiny-cuda-nn/common_device.h:68-114)
__device__ __forceinline__ float activate(uint32_t act, float v) {
  switch (act) {
    case NRF_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case NRF_ACT_EXPONENTIAL: return expf(v);
    case NRF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case NRF_ACT_SQUAREPLUS: {
      const float x = v * 10.0f;
      return 0.5f * (x + sqrtf(x * x + 4)) / 10.0f;
    }
    case NRF_ACT_SOFTPLUS: return logf(expf(v * 10.0f) + 1.0f) / 10.0f;
    case NRF_ACT_SINE: return sinf(v);
    default: return v;
  }
}

// The same activations on the hardware's transcendental instructions (v_exp_f32, v_log_f32, v_rcp_f32, v_sqrt_f32: ~1 ulp of
// fp32 each, no range reduction, no division sequence) -- for the hidden layers of the register-resident DEPTH instance
// (mlp_tiles_depth), whose results are rounded to fp16 (11 bits) right after: the libm forms above cost that instance ~70
// spilled registers.  Sine keeps the generic instance (v_sin_f32 loses accuracy with the argument's size).
__device__ __forceinline__ float activate_native(uint32_t act, float v) {
  switch (act) {
    case NRF_ACT_RELU: return fmaxf(v, 0.0f);
    case NRF_ACT_EXPONENTIAL: return __expf(v);
    case NRF_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    case NRF_ACT_SQUAREPLUS: {
      const float x = v * 10.0f;
      return (0.5f * (x + __builtin_amdgcn_sqrtf(x * x + 4))) * 0.1f;
    }
    case NRF_ACT_SOFTPLUS: return __logf(__expf(v * 10.0f) + 1.0f) * 0.1f;
    default: return v;
  }
}

// ------------------------------------------------------------- ray setup ----
// set_rays_d: the fixed-size Eigen reductions are a + (b + c).
__device__ __forceinline__ void ray_dir(const float* R, const float* cam, int px, int py, float d[3]) {
  const float i = (float)((double)px + 0.5);
  const float j = (float)((double)py + 0.5);
  const float zs = 1.0f;
  const float xs = (i - cam[2]) / cam[0] * zs;
  const float ys = (j - cam[3]) / cam[1] * zs;
  const float n = sqrtf(xs * xs + (ys * ys + zs * zs));
  const float v0 = xs / n, v1 = ys / n, v2 = zs / n;
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = R[3 * r + 0] * v0 + (R[3 * r + 1] * v1 + R[3 * r + 2] * v2);
}

__device__ __forceinline__ void near_far(const float* aabb, const float o[3], const float d[3], float min_near,
                                         float& near_out, float& far_out) {
  const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
  float nr = (aabb[0] - o[0]) * rdx, fr = (aabb[3] - o[0]) * rdx;
  if (nr > fr) { const float c = nr; nr = fr; fr = c; }
  float ny = (aabb[1] - o[1]) * rdy, fy = (aabb[4] - o[1]) * rdy;
  if (ny > fy) { const float c = ny; ny = fy; fy = c; }
  bool miss = (nr > fy) || (ny > fr);
  if (ny > nr) nr = ny;
  if (fy < fr) fr = fy;
  float nz = (aabb[2] - o[2]) * rdz, fz = (aabb[5] - o[2]) * rdz;
  if (nz > fz) { const float c = nz; nz = fz; fz = c; }
  miss = miss || (nr > fz) || (nz > fr);
  if (nz > nr) nr = nz;
  if (fz < fr) fr = fz;
  if (nr < min_near) nr = min_near;
  near_out = miss ? 3.402823466e+38f : nr;
  far_out = miss ? 3.402823466e+38f : fr;
}

// Interval of t on which the ray is inside `box` (slab test); empty when t_in > t_out.  Used with
// the inflated box of occupied cells: a march trip at a t outside this interval tests a cell
// that is certainly empty, so no sample exists there and the ray may stop at t_out (exact).
__device__ __forceinline__ void box_interval(const float* box, const float o[3], float rdx, float rdy, float rdz,
                                             float& t_in, float& t_out) {
  float a = (box[0] - o[0]) * rdx, b = (box[3] - o[0]) * rdx;
  t_in = fminf(a, b);
  t_out = fmaxf(a, b);
  a = (box[1] - o[1]) * rdy; b = (box[4] - o[1]) * rdy;
  t_in = fmaxf(t_in, fminf(a, b));
  t_out = fminf(t_out, fmaxf(a, b));
  a = (box[2] - o[2]) * rdz; b = (box[5] - o[2]) * rdz;
  t_in = fmaxf(t_in, fminf(a, b));
  t_out = fminf(t_out, fmaxf(a, b));
}

// Conservative visibility of the occupied set along one ray (single cascade): a 3-D DDA over the
// coarse grid (cells of 4x4x4 density cells); a coarse bit is set when the cell contains, or lies
// within ONE density cell of, an occupied density cell (host: nrf_load_model).
// A march trip can only find a sample at a point of the ray that lies in an occupied density cell
// f.  The DDA visits the coarse cell containing that point or, when the point sits within fp error
// of a coarse boundary, its neighbour across that boundary -- which is then within one density
// cell of f.  Either way the visited cell's bit is set.  So: no set bit on the way -> the ray
// cannot produce a sample; otherwise no sample exists beyond the exit of the last set cell.  Returns false when the ray
// is sample-free; otherwise t_last is the exit parameter of the last set cell and t_first the entry parameter of
// the first one: a march trip at t < t_first tests a density cell that is certainly empty (same argument), so
// march_next skips the occupancy lookups there (the hop arithmetic of the trip is unchanged).
__device__ __forceinline__ bool coarse_visibility(const uint32_t* __restrict__ dil, int Hc, float mip_bound, const float o[3],
                                                  const float d[3], float rdx, float rdy, float rdz, float t0, float t1,
                                                  float& t_first, float& t_last) {
  const float cs = 2.0f * mip_bound / (float)Hc, rcs = (float)Hc / (2.0f * mip_bound);
  const float rd[3] = {rdx, rdy, rdz};
  int i[3], step[3];
  float tmax[3], tdelta[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = o[a] + t0 * d[a];
    int c = (int)floorf((p + mip_bound) * rcs);
    c = c < 0 ? 0 : (c > Hc - 1 ? Hc - 1 : c);
    i[a] = c;
    step[a] = d[a] >= 0.0f ? 1 : -1;
    const float edge = (float)(c + (d[a] >= 0.0f ? 1 : 0)) * cs - mip_bound;
    const bool flat = !(fabsf(rd[a]) <= 3.0e38f);  // d == 0 (or NaN): never crosses along this axis
    tmax[a] = flat ? 3.0e38f : (edge - o[a]) * rd[a];
    tdelta[a] = flat ? 3.0e38f : cs * fabsf(rd[a]);
  }
  bool any = false;
  float t = t0;
  t_last = t0;
  t_first = t0;
  for (int guard = 0; guard < 3 * Hc + 3; ++guard) {
    const uint32_t cc = ((uint32_t)i[0] * Hc + (uint32_t)i[1]) * Hc + (uint32_t)i[2];
    const float t_exit = fminf(tmax[0], fminf(tmax[1], tmax[2]));
    if ((dil[cc >> 5] >> (cc & 31u)) & 1u) {
      if (!any) t_first = t;  // entry of the first cell that can hold a sample
      any = true;
      t_last = t_exit;
    }
    if (!(t_exit < t1)) break;
    t = t_exit;
    // advance along the axis (axes) reaching their boundary first
    if (tmax[0] <= t_exit) { i[0] += step[0]; tmax[0] += tdelta[0]; }
    if (tmax[1] <= t_exit) { i[1] += step[1]; tmax[1] += tdelta[1]; }
    if (tmax[2] <= t_exit) { i[2] += step[2]; tmax[2] += tdelta[2]; }
    if ((unsigned)i[0] >= (unsigned)Hc || (unsigned)i[1] >= (unsigned)Hc || (unsigned)i[2] >= (unsigned)Hc) break;
  }
  return any;
}

// ----------------------------------------------------------------- march ----
// kernel_march_rays (render_utils.h:524-655), restated so that one loop trip costs ~40 VALU and,
// in empty space, no global load and no division:
//   * cell_bound[level][v] = ((v / (H-1)) * 2 - 1) * mip_bound for v = 0..H is tabulated on the
//     host with the reference's own fp32 operation order (render_utils.h:643), since
//     nx + 0.5f + 0.5f*sign(d) is exactly the integer nx or nx+1;
//   * 1/mip_bound is 2^-level or the host's 1/bound (both correctly rounded, as `1 / mip_bound`);
//   * a coarse occupancy bitfield (4x4x4 cells OR-ed) answers "empty" without touching the fine one.
// Every value that reaches a sample (x, y, z, dt, t) is produced by the same individually rounded
// fp32 operations as the reference/oracle, so marching stays bit-exact.
struct MarchConst {
  float bound, rbound, dt_gamma, dt_min, dt_max, Hf, Hm1;
  uint32_t H, C, HH, HHH;
  uint32_t coarse_shift;  // 2 when the coarse grid is present, else 0
  uint32_t Hc;            // H >> coarse_shift
  uint32_t log2H;         // MARCH_UNIT / MARCH_POW2 instances only (H is a power of two there)
  float halfH;            // 0.5f * H
  int log2_bound;         // MARCH_POW2 instances only: bound == 2^log2_bound
};

__device__ __forceinline__ MarchConst march_const(const DevModel& M, float dt_gamma) {
  MarchConst c;
  c.bound = M.bound;
  c.rbound = M.rbound;
  c.dt_gamma = dt_gamma;
  c.dt_min = 2 * 1.7320508075688772f / 1024;  // MIN_STEPSIZE, render_utils.h:181-183
  c.dt_max = 2 * M.bound / (float)M.H;
  c.H = M.H;
  c.C = M.cascade;
  c.HH = M.H * M.H;
  c.HHH = M.H * M.H * M.H;
  c.Hf = (float)M.H;
  c.Hm1 = (float)(M.H - 1);
  c.coarse_shift = M.coarse_shift;
  c.Hc = M.H >> M.coarse_shift;
  c.log2H = 31u - (uint32_t)__builtin_clz(M.H | 1u);
  c.halfH = 0.5f * (float)M.H;
  {
    int e;
    (void)frexpf(M.bound, &e);
    c.log2_bound = e - 1;  // exact when bound is a power of two (the only case it is used in)
  }
  return c;
}

// The perturb branch of kernel_march_rays (render_utils.h:585-589): `pcg32 rng(n, perturb); t += MIN_STEPSIZE() * rng.next_float()`
// ahead of every march call.  pcg32_first_float = pcg32(initstate, initseq).next_float() (T/dependencies/pcg32/pcg32.h:48-62
// seed, :65-71 next_uint, :108-117 next_float): the one number a freshly seeded generator is asked for.
__device__ __forceinline__ float pcg32_first_float(uint64_t initstate, uint64_t initseq) {
  const uint64_t MULT = 0x5851f42d4c957f2dULL;  // PCG32_MULT
  const uint64_t inc = (initseq << 1u) | 1u;
  uint64_t state = inc;            // seed(): state = 0; next_uint()
  state += initstate;
  state = state * MULT + inc;      // seed(): next_uint()
  const uint32_t xorshifted = (uint32_t)(((state >> 18u) ^ state) >> 27u);  // next_float(): next_uint()'s output of this state
  const uint32_t rot = (uint32_t)(state >> 59u);
  const uint32_t r = (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
  return __builtin_bit_cast(float, (r >> 9) | 0x3f800000u) - 1.0f;
}

enum : int { MARCH_FOUND = 0, MARCH_EXHAUSTED = 1, MARCH_OUT_OF_BUDGET = 2 };

// Advances t until the next occupied sample (MARCH_FOUND: x,y,z = clamped position, dt_out = its
// step; t is NOT yet advanced by dt), until t >= far (MARCH_EXHAUSTED) or until `budget` loop trips
// are spent (MARCH_OUT_OF_BUDGET, t rests on the next candidate).  One loop trip = one trip of the
// `while (t < far && step < n_step)` loop of render_utils.h:593-653.
//   occ      fine bitfield (global)     coarse  coarse bitfield or nullptr     ctab  cell_bound table
// UNIT == true: single cascade with mip_bound == 1 (bound >= 1) and H a power of two: level 0,
// `x * mip_rbound` is `x * 1` (exact), so the level / mip arithmetic disappears; (0.5f*(x+1))*H is
// (x+1)*(0.5f*H) (both factors are powers of two, so neither product rounds); cell indices are
// shifts and ors.  clamp() is v_med3_f32 (same value as fminf(hi, fmaxf(lo, x)) for non-NaN x).
__device__ __forceinline__ float clamp3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// MODE: MARCH_GENERIC = the literal arithmetic; MARCH_UNIT = one cascade, mip_bound 1, H = 2^k (see above);
// MARCH_POW2 = several cascades with H = 2^k and bound = 2^b: every mip_bound is a power of two, so
// (0.5f * (x * mip_rbound + 1)) * H == fma(x, 2^(k-1-lb), 2^(k-1)) (power-of-two scalings do not round and
// commute with the one rounding of the addition), the level clamp is an integer med3, indices are shifts.
enum : int { MARCH_GENERIC = 0, MARCH_UNIT = 1, MARCH_POW2 = 2 };
template <bool COARSE, int MODE>
__device__ __forceinline__ int march_next(const MarchConst& c, const uint32_t* __restrict__ occ, const uint32_t* coarse,
                                          const float* ctab, float ox, float oy, float oz, float dx, float dy, float dz,
                                          float rdx, float rdy, float rdz, int sx, int sy, int sz, float far, float t_skip,
                                          int& budget, float& t, float& x, float& y, float& z, float& dt_out) {
  while (t < far) {
    if (budget <= 0) return MARCH_OUT_OF_BUDGET;
    --budget;
#ifdef NRF_DIAG_EXTRA_MARCH
    // diagnostic build (scripts/marginal_cost.sh, never shipped): NRF_DIAG_EXTRA_MARCH extra v_mul_f32 per cell trip whose
    // results nobody reads: what does a vector instruction of the march phase cost in situ?
#pragma unroll
    for (int e = 0; e < NRF_DIAG_EXTRA_MARCH; ++e) {
      float sink;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sink) : "v"(t), "v"(dx));
    }
#endif
#ifdef NRF_DIAG_FF_UPPER
    // diagnostic build (never shipped; WRONG pictures): every trip before t_skip is a bare step -- the upper bound of what an
    // exact fast-forward through the empty stretch ahead of the object could return
    if (t < t_skip) {
      do {
        t += clamp3(t * c.dt_gamma, c.dt_min, c.dt_max);
      } while (t < t_skip && t < far);
      continue;
    }
#endif
    x = clamp3(ox + t * dx, -c.bound, c.bound);
    y = clamp3(oy + t * dy, -c.bound, c.bound);
    z = clamp3(oz + t * dz, -c.bound, c.bound);
    constexpr bool UNIT = MODE == MARCH_UNIT, POW2 = MODE == MARCH_POW2;
    int level = 0;
    int nx, ny, nz;
    if (UNIT) {
      // mip_bound == 1: (x * 1 + 1) == (x + 1); H = 2^k: (0.5f * v) * H == v * (0.5f * H), no rounding; and
      // round(x + 1) * 2^j == round(x * 2^j + 2^j) (scaling by a power of two commutes with rounding), so the
      // add and the multiply are ONE fused multiply-add with the same value as the reference's three operations
      nx = (int)clamp3(__builtin_fmaf(x, c.halfH, c.halfH), 0.0f, c.Hm1);
      ny = (int)clamp3(__builtin_fmaf(y, c.halfH, c.halfH), 0.0f, c.Hm1);
      nz = (int)clamp3(__builtin_fmaf(z, c.halfH, c.halfH), 0.0f, c.Hm1);
    } else if (POW2) {
      const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
      int exponent;
      (void)frexpf(mx, &exponent);
      level = min(max(exponent, 0), (int)c.C - 1);
      const float scale = ldexpf(c.halfH, -min(level, c.log2_bound));  // halfH / mip_bound
      nx = (int)clamp3(__builtin_fmaf(x, scale, c.halfH), 0.0f, c.Hm1);
      ny = (int)clamp3(__builtin_fmaf(y, scale, c.halfH), 0.0f, c.Hm1);
      nz = (int)clamp3(__builtin_fmaf(z, scale, c.halfH), 0.0f, c.Hm1);
    } else {
      float mip_bound = fminf(1.0f, c.bound), mip_rbound;
      if (c.C > 1) {
        const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int exponent;
        (void)frexpf(mx, &exponent);
        level = (int)fminf((float)c.C - 1, fmaxf(0.0f, (float)exponent));
        mip_bound = fminf(ldexpf(1.0f, level), c.bound);
      }
      mip_rbound = (mip_bound == c.bound) ? c.rbound : ldexpf(1.0f, -level);  // == 1 / mip_bound
      // `0.5 * (x*mip_rbound + 1) * H` is double arithmetic in the reference; for H < 2^24 the
      // double product is exact, so its narrowing to float equals the fp32 product (0.5f*v)*H.
      nx = (int)clamp3((0.5f * (x * mip_rbound + 1)) * c.Hf, 0.0f, c.Hm1);
      ny = (int)clamp3((0.5f * (y * mip_rbound + 1)) * c.Hf, 0.0f, c.Hm1);
      nz = (int)clamp3((0.5f * (z * mip_rbound + 1)) * c.Hf, 0.0f, c.Hm1);
    }
    // all loads of the trip are issued together (addresses depend only on the cell), so the
    // trip pays one memory latency instead of three dependent ones
    const uint32_t cell = (UNIT || POW2) ? ((((((uint32_t)level << c.log2H) | (uint32_t)nx) << c.log2H) | (uint32_t)ny) << c.log2H) | (uint32_t)nz
                                         : (uint32_t)level * c.HHH + (uint32_t)nx * c.HH + (uint32_t)ny * c.H + (uint32_t)nz;
    const float* tab = UNIT ? ctab : ctab + (uint32_t)level * (c.H + 1);
    const float bx = tab[nx + sx], by = tab[ny + sy], bz = tab[nz + sz];
    bool occupied = false;
    if (!(t < t_skip)) {  // before t_skip the cell is known to be empty (coarse_visibility)
      if (COARSE) {
        const uint32_t lc = c.log2H - 2u;
        const uint32_t cc = (UNIT || POW2) ? (((((((uint32_t)level << lc) | ((uint32_t)nx >> 2)) << lc) | ((uint32_t)ny >> 2)) << lc) | ((uint32_t)nz >> 2))
                                           : ((uint32_t)level * c.Hc + ((uint32_t)nx >> 2)) * c.Hc * c.Hc + ((uint32_t)ny >> 2) * c.Hc + ((uint32_t)nz >> 2);
        const bool coarse_occ = (coarse[cc >> 5] >> (cc & 31u)) & 1u;
        if (coarse_occ) occupied = (occ[cell >> 5] >> (cell & 31u)) & 1u;
      } else {
        occupied = (occ[cell >> 5] >> (cell & 31u)) & 1u;
      }
    }
    if (occupied) {
      dt_out = clamp3(t * c.dt_gamma, c.dt_min, c.dt_max);
      return MARCH_FOUND;
    }
    const float tx = (bx - x) * rdx;  // (((nx+0.5f+0.5f*sign)/(H-1)*2-1)*mip_bound - x) * rdx
    const float ty = (by - y) * rdy;
    const float tz = (bz - z) * rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
      t += clamp3(t * c.dt_gamma, c.dt_min, c.dt_max);
    } while (t < tt);
  }
  return MARCH_EXHAUSTED;
}

#ifndef NRF_MARCH_FF
#define NRF_MARCH_FF 1
#endif
// ------------------------------------------------- barrier fast-forward ----
// The stretch of a ray between its entry into the aabb and t_skip (the first coarse cell that can hold a sample) used to be
// simulated trip by trip -- ~100-200 trips of ~45 vector instructions per ray that test nothing (DESIGN.md "March").  It
// cannot simply be skipped: which of the steps t_{k+1} = t_k + dt(t_k) are trip STARTS (= tested positions) beyond t_skip
// depends on the hops before.  But every trip start is a member of that step sequence, and the hop of
// render_utils.h:641-651 has a structure that makes some members certain trip starts:
//   for an axis a with d_a < 0 the hop target of a trip in slab n (cell index n along a) is at most the time the ray
//   reaches the plane b_a(n) = ((n / (H-1)) * 2 - 1) -- the `/(H-1)` boundary, which lies INSIDE slab n;
//   * a trip that starts above the plane (slab n' >= n: its own plane b_a(n') comes earlier) ends at or before the first
//     member behind the plane:    tt = t + max(0, min(tx, ty, tz)) <= t + tx = T_a(n) (+- rounding);
//   * a trip that starts below the plane, still in slab n, has tx <= 0: tt = t, exactly one step;
//   * a trip that starts within rounding of the plane has a hop of rounding size: one step.
//   Hence no trip jumps over e = the first member >= T_a(n) + eps (eps > the rounding of tt, see below): whatever the trips
//   before did, one of them ENDS at e, so e is a trip start (if the member before e lies above the plane its trip ends at
//   the first member >= tt, tt in (it, T + rounding]: e; if it lies within eps of, or below, the plane: one step: e).
// All members before e start before T + eps <= t_skip, where no trip can find a sample: the reference's state at e is
// (t = e, last_t and the sample count untouched) -- the same as after stepping through the members without any trip.
// So: take the LAST such plane ahead of t_skip over the ray's negative axes and step to it.  Rays without a negative
// direction component, or whose last plane lies before their start, keep their trips.
//   eps: tt is computed as fl(t + fl(fl(b - x) * rd)) with x = fl(o + fl(t d)): x is within ~2 ulp(|o| + |t d|) of the
//   true position, the products / sums add ~3 more relative roundings of a value <= t: |tt - T| <= ~1e-6 (1 + |rd_a|) for
//   magnitudes up to ~8; eps = 4e-6 (t_skip + bound + 2) (1 + |rd_a|) leaves a factor of four and scales with the magnitudes.
// One cascade (mip_bound = min(1, bound)); with several cascades the planes of different levels are not ordered along a ray:
// fast_forward_to_barrier_pow2 below.
__device__ __forceinline__ float barrier_before(const float* ctab, int H, float mb, float mag, float o, float d, float rd, float t_skip) {
  // the last plane of a NEGATIVE axis the ray passes before t_skip: returns T + eps, or -inf when there is none.
  // mb = the level's mip_bound.  Slab n's plane is ctab[n], n = 0 .. H-1; a position beyond the grid (one cascade with
  // bound > 1) clamps into slab H-1, whose plane it has not passed yet -- ctab[H] is no slab's plane.
  const float NONE = -3.402823466e+38f;
  if (!(d < 0.0f) || !(rd > -3.0e38f)) return NONE;
  const float xs = o + t_skip * d;                                    // (approximate) position at t_skip
  const float v = clamp3(ceilf((xs / mb + 1.0f) * (0.5f * (float)(H - 1))), 0.0f, (float)H);  // first plane index at or above it
  int n = (int)v;
  if (n > H - 1) return NONE;
  const float eps = 4.0e-6f * mag * (1.0f + fabsf(rd));  // mag = t_skip + bound + 2 >= the magnitudes of o, t d and the position
  float e = (ctab[n] - o) * rd + eps;
  if (!(e <= t_skip)) {  // the approximate position put the plane a hair behind t_skip: the one before it
    if (++n > H - 1) return NONE;
    e = (ctab[n] - o) * rd + eps;
  }
  return e <= t_skip ? e : NONE;
}

// mb: the mip_bound of the only level, min(1, bound)
__device__ __forceinline__ float fast_forward_to_barrier(const MarchConst& c, const float* ctab, float mb, const float o[3], const float d[3],
                                                         float rdx, float rdy, float rdz, float t, float t_skip, float far) {
  const float mag = t_skip + c.bound + 2.0f;
  const float tb = fminf(fmaxf(barrier_before(ctab, (int)c.H, mb, mag, o[0], d[0], rdx, t_skip),
                               fmaxf(barrier_before(ctab, (int)c.H, mb, mag, o[1], d[1], rdy, t_skip),
                                     barrier_before(ctab, (int)c.H, mb, mag, o[2], d[2], rdz, t_skip))), far);
  // every step below is a member before e: t < tb <= T + eps <= t_skip, and t < far as in `while (t < far ...)`
  // (an exact k-step jump -- t + k dt_max is one exact fma while t stays in its binade and dt_max is a multiple of its ulp --
  //  was built and measured: 0.4 % SLOWER than this four-instruction loop; the division and frexp per binade cost more)
  while (t < tb) t += clamp3(t * c.dt_gamma, c.dt_min, c.dt_max);
  return t;
}

// Several cascades (any H, any bound).  A trip's level follows its position: level L where the max-norm m of
// the position lies in [2^(L-1), 2^L) (L = 0: m < 1; L = C-1: everything beyond), and a level-L trip hops on level L's
// planes b_L(n) = ((n / (H-1)) * 2 - 1) * mip_bound(L).  Planes of different levels are not ordered along a ray, so a level-L
// plane at time T (negative axis a) is a barrier only when no trip of another level can reach past it:
//   (W) every trip that starts in W = [T - slab(L+1) |rd_a| - eps, T + eps] has level L: the ray stays inside the (deflated)
//       cube of half-size 2^L and outside the (inflated) one of half-size 2^(L-1) throughout W.  Then the single-level
//       argument holds inside W, and trips of levels <= L + 1 that start before W end before T: a hop along a is at most one
//       slab of the trip's own level, slab(L') = 2 mip_bound(L') / H <= slab(L + 1);
//   (O) for every level L' >= L + 2: no trip of level L' starts within its own reach of T -- the ray is inside the (deflated)
//       cube of half-size 2^(L'-1) during [T - slab(L') |rd_a| - eps, T].
// The cubes are moved by `pad` (1e-4 of the magnitudes involved, ~1000x the rounding of a position) so that a computed level
// cannot differ from the one assumed here; a NaN in a slab test (a ray in a face plane) rejects the plane.  The search starts
// with the level at t_skip and, when none of its planes qualifies (t_skip sits just behind a shell boundary), moves on to the
// stretch before the last shell crossing.  Validity rests on (W), (O) and T + eps <= t_skip alone, not on how T was found.
__device__ __forceinline__ bool cube_interval(float s, const float o[3], const float rd[3], float& t_in, float& t_out) {
  bool ok = true;
  t_in = -3.402823466e+38f;
  t_out = 3.402823466e+38f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = (-s - o[a]) * rd[a], v = (s - o[a]) * rd[a];
    ok = ok && (u == u) && (v == v);
    t_in = fmaxf(t_in, fminf(u, v));
    t_out = fminf(t_out, fmaxf(u, v));
  }
  return ok;
}

__device__ __forceinline__ float fast_forward_to_barrier_pow2(const MarchConst& c, const float* ctab, const float o[3], const float d[3],
                                                              float rdx, float rdy, float rdz, float t, float t_skip, float far) {
  const float NONE = -3.402823466e+38f;
  const float rd[3] = {rdx, rdy, rdz};
  const int C = (int)c.C, H = (int)c.H;
  const float mag = t_skip + c.bound + 2.0f;
  const float pad = 1.0e-4f * mag;
  const float slab0 = 2.0f / (float)H;  // slab(L) = slab0 * mip_bound(L)
  float t_hi = t_skip, best = NONE;
  for (int it = 0; it <= C && !(best > NONE) && t_hi > t; ++it) {
    // the level just before t_hi (march_next's arithmetic; a wrong guess only costs the candidates)
    const float qx = clamp3(o[0] + t_hi * d[0], -c.bound, c.bound), qy = clamp3(o[1] + t_hi * d[1], -c.bound, c.bound),
                qz = clamp3(o[2] + t_hi * d[2], -c.bound, c.bound);
    int ex;
    (void)frexpf(fmaxf(fabsf(qx), fmaxf(fabsf(qy), fabsf(qz))), &ex);
    const int L = min(max(ex, 0), C - 1);
    const float mb = fminf(ldexpf(1.0f, L), c.bound);  // mip_bound(L)
    const float* tab = ctab + (uint32_t)L * (uint32_t)(H + 1);
    const float slab_next = slab0 * fminf(ldexpf(1.0f, L + 1), c.bound);
    const bool has_in = L >= 1, has_out = L <= C - 2;
    float ai = 0.f, bi = 0.f, ao = 0.f, bo = 0.f;
    const bool ok_in = has_in ? cube_interval(ldexpf(1.0f, L - 1) + pad, o, rd, ai, bi) : true;
    const bool ok_out = has_out ? cube_interval(ldexpf(1.0f, L) - pad, o, rd, ao, bo) : true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!(d[a] < 0.0f) || !(rd[a] > -3.0e38f)) continue;
      const float ard = fabsf(rd[a]);
      const float eps = 4.0e-6f * mag * (1.0f + ard);
      const float xs = o[a] + t_hi * d[a];
      int n = (int)clamp3(ceilf((xs / mb + 1.0f) * (0.5f * (float)(H - 1))), 0.0f, (float)H);
      if (n > H - 1) continue;  // (beyond the level's grid: clamped into slab H-1, above its plane)
      float T = (tab[n] - o[a]) * rd[a], e = T + eps;
      if (!(e <= t_hi)) {
        if (++n > H - 1) continue;
        T = (tab[n] - o[a]) * rd[a];
        e = T + eps;
      }
      if (!(e <= t_hi)) continue;
      const float w_lo = T - slab_next * ard - eps;
      bool ok = ok_in && ok_out;
      if (has_out) ok = ok && ao <= w_lo && e <= bo;                 // inside the outer cube of shell L throughout W
      if (has_in) ok = ok && (ai > bi || e <= ai || w_lo >= bi);     // never inside its inner cube
      for (int Lp = L + 2; Lp <= C - 1 && ok; ++Lp) {                 // (O)
        float a2, b2;
        ok = cube_interval(ldexpf(1.0f, Lp - 1) - pad, o, rd, a2, b2);
        ok = ok && a2 <= T - slab0 * fminf(ldexpf(1.0f, Lp), c.bound) * ard - eps && e <= b2;
      }
      if (ok) best = fmaxf(best, e);
    }
    if (best > NONE) break;
    // nothing here: the stretch before the last crossing of shell L's boundaries
    float nxt = NONE;
    const float lim = t_hi - 1.0e-6f * mag;
    if (has_in && ok_in) {
      if (ai < lim) nxt = fmaxf(nxt, ai);
      if (bi < lim) nxt = fmaxf(nxt, bi);
    }
    if (has_out && ok_out) {
      if (ao < lim) nxt = fmaxf(nxt, ao);
      if (bo < lim) nxt = fmaxf(nxt, bo);
    }
    t_hi = nxt - 4.0f * pad;  // (well into the neighbouring shell; NONE ends the search)
  }
  const float tb = fminf(best, far);
  while (t < tb) t += clamp3(t * c.dt_gamma, c.dt_min, c.dt_max);
  return t;
}

// ------------------------------------------------------------- hash grid ----
// (half)(w * (float)h) for both halves of a table entry: fp32 product rounded to fp32, then to fp16
// (grid.h:258-260).  v_fma_mix_f32 reads the fp16 half directly and returns fma(w, h, -0.0) in fp32,
// which is bit-for-bit `w * (float)h` (the conversion is exact, x + (-0.0) == x including the sign
// of zero): one instruction instead of v_cvt_f32_f16 + v_mul_f32.  The conversion to fp16 stays a
// separate v_cvt_pk_f16_f32: v_fma_mixlo/hi_f16 was measured to round the exact product ONCE,
// which differs from these two roundings for about 1 in 30 000 products.
__device__ __forceinline__ half2_t weight_times_entry(float w, uint32_t entry) {
  float lo, hi;
  const float neg_zero = -0.0f;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(lo) : "v"(w), "v"(entry), "v"(neg_zero));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(hi) : "v"(w), "v"(entry), "v"(neg_zero));
  half2_t p;
  p.x = (half_t)lo;
  p.y = (half_t)hi;
  return p;
}

// One (sample, level) of kernel_grid<half,3,2>: 8 corner gathers of a half2,
// fp16 accumulation in corner order (grid.h:236-262).  Returns the packed
// half2 (feature 0 in the low half).
//   Every level is LV_DENSE or LV_HASH_POW2 (anything else runs in the generic instance, nrf_generic.h); dense
//   levels are stored with res^2 + res + 1 wrapped entries appended (nrf_api.hip), so `index % size` of
//   grid.h:116 needs no instruction: a dense index never exceeds size + res^2 + res.
//   UNI: 0 = the lanes of the wave may mix dense and hashed levels (per-lane select);
//        1 = every lane's level is dense, 2 = every lane's level is power-of-two hashed.
// The level is evaluated in two halves so that a caller can put the gathers of SEVERAL levels in
// flight before consuming any of them (network_from_lds: all 4 levels of a sample, 32 loads per
// lane): the network phase is bound by gather latency, not by instruction issue.
//   level_gather: corner indices + the 8 loads (results not touched) + the fractional position
//   level_interp: trilinear weights and the fp16 accumulation in corner order
// level_offsets: the eight corners' BYTE offsets into the table.  SH = log2 of an entry's bytes: 2 for F = 2 (one dword per entry,
// the base.json shape), 3 / 4 for F = 4 / 8 (nrf_load_model shifts the per-level constants off_b / my_b / mz_b / mask_b alike).
template <int UNI = 0, int SH = 2>
__device__ __forceinline__ void level_offsets(const LevelParams L, float px, float py, float pz, uint32_t (&off)[8], float (&frac)[3]) {
  float fx = px * L.scale; fx = fx + 0.5f;
  float fy = py * L.scale; fy = fy + 0.5f;
  float fz = pz * L.scale; fz = fz + 0.5f;
  // positions are in [0,1] (march clamps to the aabb), so f >= 0.5: the truncating
  // conversion is floor, and v_fract_f32 returns f - floor(f) exactly (the subtraction is exact
  // for f >= 0; the instruction's clamp to 1-ulp only concerns tiny negative inputs)
  const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
  frac[0] = __builtin_amdgcn_fractf(fx);
  frac[1] = __builtin_amdgcn_fractf(fy);
  frac[2] = __builtin_amdgcn_fractf(fz);

  // The gathers are MUBUF loads: address = table base (buffer resource, SGPRs) + a 32-bit BYTE
  // offset per lane, so no 64-bit address arithmetic is spent per corner.  The shift by 2 is folded
  // into the per-axis terms ((a ^ b ^ d) << 2 == (a<<2) ^ (b<<2) ^ (d<<2), (g * P) << 2 == g * (P << 2)
  // mod 2^32); nrf_load_model rejects tables of 4 GiB or more.
  const uint32_t level_off = L.off_b;
  {
    // dense and power-of-two hashed levels share the per-axis parts; only the combiner differs
    const bool hashed = UNI == 2 || (UNI == 0 && L.mode == LV_HASH_POW2);
    const uint32_t my = UNI == 2 ? (2654435761u << SH) : L.my_b;
    const uint32_t mz = UNI == 2 ? (805459861u << SH) : L.mz_b;
    const uint32_t mask = UNI == 1 ? 0xffffffffu : L.mask_b;
    const uint32_t ax0 = (gx << SH) + (hashed ? 0u : level_off);  // dense: the level offset rides on the x term
    const uint32_t ax[2] = {ax0, ax0 + (1u << SH)};
    const uint32_t ay0 = gy * my, az0 = gz * mz;
    const uint32_t ay[2] = {ay0, ay0 + my};
    const uint32_t az[2] = {az0, az0 + mz};
    if (UNI == 2) {
      // power-of-two hashed levels start at a multiple of their size (nrf_load_model), so the level offset has no
      // bit in common with the mask: ((a ^ b ^ d) & mask) + off == ((a & mask) | off) ^ ((b ^ d) & mask) -- the
      // masking moves from the 8 corners to 2 + 4 per-axis terms
      const uint32_t am[2] = {(ax[0] & mask) | level_off, (ax[1] & mask) | level_off};
      uint32_t bd[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bd[q] = (ay[q & 1] ^ az[q >> 1]) & mask;
#pragma unroll
      for (int c = 0; c < 8; ++c) off[c] = am[c & 1] ^ bd[c >> 1];
    } else if (UNI == 1) {
#pragma unroll
      for (int c = 0; c < 8; ++c) off[c] = ax[c & 1] + ay[(c >> 1) & 1] + az[(c >> 2) & 1];
    } else {
      // lanes of one instruction mix dense and hashed levels: both 2-term forms, one v_cndmask per corner.
      // The additive form is masked as well: dense levels carry mask = ~0, an LV_ADD_POW2 level its (size - 1) << 2
      // with mz_b = 0 (nrf_load_model); the level offset is added last (hashed / XY levels are aligned, see above)
      const uint32_t axr[2] = {gx << SH, (gx << SH) + (1u << SH)};
      const uint32_t am[2] = {(axr[0] & mask) | level_off, (axr[1] & mask) | level_off};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t b = ay[q & 1], d = az[q >> 1];
        const uint32_t bd_x = (b ^ d) & mask, bd_s = b + d;
#pragma unroll
        for (int e = 0; e < 2; ++e) off[2 * q + e] = hashed ? (am[e] ^ bd_x) : (((axr[e] + bd_s) & mask) + level_off);
      }
    }
  }
}

template <int UNI = 0>
__device__ __forceinline__ void level_gather(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                             float py, float pz, uint32_t (&v)[8], float (&frac)[3]) {
  uint32_t off[8];
  level_offsets<UNI, 2>(L, px, py, pz, off, frac);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
#ifdef NRF_DIAG_HASH_PAIRS
  // diagnostic build (never shipped; WRONG values for odd x): every (x, x + 1) corner pair of a hashed level as ONE aligned
  // 8-byte gather -- the upper bound of what pairing the corners of hashed levels could return
  if (UNI == 2) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[2 * q] & ~4u, 0, 0);
      v[2 * q] = w.x;
      v[2 * q + 1] = w.y;
    }
    return;
  }
#endif
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[c], 0, 0);
}

// Cell-major quad copy of a level (round 6): nrf_load_model stores, for every cell (x, y, z) a sample of the level can fall into
// (x, y < res, z <= res), ONE 16-byte entry holding the four half2 table entries of the corners (x, y, z), (x + 1, y, z),
// (x, y + 1, z), (x + 1, y + 1, z) -- each COPIED from the entry grid_index (grid.h:100-117) names for that corner, hash collisions
// and wrapped dense indices included, so the eight values are bit for bit what level_gather returns, in the same corner order
// (bit 0 of the corner number = x, bit 1 = y, bit 2 = z: grid.h:236-262).  A level's eight corners are then TWO aligned 16-byte
// gathers -- the quads of cells (x, y, z) and (x, y, z + 1) -- instead of eight 4-byte ones: the texture addressers charge per
// lane address (profiles/r02/gather_probe.txt), and the index arithmetic shrinks from ~20 vector instructions to 5.
__device__ __forceinline__ void level_gather_quad(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                                  float py, float pz, uint32_t (&v)[8], float (&frac)[3]) {
  float fx = px * L.scale; fx = fx + 0.5f;
  float fy = py * L.scale; fy = fy + 0.5f;
  float fz = pz * L.scale; fz = fz + 0.5f;
  const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;  // (floor: f >= 0.5, see level_offsets)
  frac[0] = __builtin_amdgcn_fractf(fx);
  frac[1] = __builtin_amdgcn_fractf(fy);
  frac[2] = __builtin_amdgcn_fractf(fz);
  // byte offset of quad (gx, gy, gz): every factor is below 2^24 (nrf_load_model grants quads to levels of res < 1024 only)
  const uint32_t row = __umul24(gy, L.q_my_b) + L.q_off_b;
  const uint32_t slab = __umul24(gz, L.q_mz_b) + row;
  const uint32_t off0 = (gx << 4) + slab;
  const uint32_t off1 = off0 + L.q_mz_b;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off0, 0, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off1, 0, 0);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// The same for quad copies beyond the 4 GiB a buffer resource's 32-bit byte offset reaches (levels 8..11 of base.json's grid are
// 0.16 + 0.41 + 1.08 + 2.85 GB of quads): global loads from a 64-bit address, base + 16 x (a 32-bit quad number: 64 GiB of reach).
// A global load has no range check, so the cell coordinates are clamped to the level's last cell -- the identity for positions
// in [0, 1] (floor(scale + 0.5) <= ceil(scale) = res - 1), NaN converts to 0.
__device__ __forceinline__ void level_gather_quad_far(const uint32_t* __restrict__ grid, const LevelParams L, float px, float py, float pz,
                                                      uint32_t (&v)[8], float (&frac)[3]) {
  float fx = px * L.scale; fx = fx + 0.5f;
  float fy = py * L.scale; fy = fy + 0.5f;
  float fz = pz * L.scale; fz = fz + 0.5f;
  uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
  frac[0] = __builtin_amdgcn_fractf(fx);
  frac[1] = __builtin_amdgcn_fractf(fy);
  frac[2] = __builtin_amdgcn_fractf(fz);
  gx = gx < L.q_max ? gx : L.q_max;
  gy = gy < L.q_max ? gy : L.q_max;
  gz = gz < L.q_max ? gz : L.q_max;
  const uint32_t row = __umul24(gy, L.q_my_b) + L.q_off_b;
  const uint32_t cell = __umul24(gz, L.q_mz_b) + row + gx;
  const uint4* q = reinterpret_cast<const uint4*>(grid) + cell;
  const uint4 a = q[0];
  const uint4 b = q[L.q_mz_b];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// FAST (opt-in, nrf_options::fast_interp; never the default): acc = (half)(w * h + acc) as ONE v_fma_mixlo/hi_f16 per half and
// corner -- the exact fp32 fma rounded once to fp16 -- instead of the reference's three roundings (product to fp32, to fp16,
// fp16 sum): 2 instead of 4 half-rate instructions per corner.  More accurate than the reference's arithmetic, but not its
// bits: features differ by an fp16 ulp now and then (tests/test_parity_gpu.py states the tolerance).
template <bool FAST = false>
__device__ __forceinline__ uint32_t level_interp(const uint32_t (&v)[8], const float (&frac)[3]) {
  const float wx[2] = {1 - frac[0], frac[0]};
  const float wy[2] = {1 - frac[1], frac[1]};
  const float wz[2] = {1 - frac[2], frac[2]};
  if constexpr (FAST) {
    uint32_t a = 0u;  // packed (feature 0, feature 1)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];
      asm("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(w), "v"(v[c]));
      asm("v_fma_mixhi_f16 %0, %1, %2, %0 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "+v"(a) : "v"(w), "v"(v[c]));
    }
    return a;
  }
  half2_t acc = {(half_t)0.0f, (half_t)0.0f};
#ifdef NRF_DIAG_DROP_INTERP_CVT
  float diag_acc_lo = 0.f, diag_acc_hi = 0.f;
#endif
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    // weight = ((1 * wx) * wy) * wz in dimension order (grid.h:240-252)
    const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];
#ifdef NRF_DIAG_DROP_INTERP_CVT
    // diagnostic build (scripts/marginal_cost.sh, never shipped): the same gathers and fp32 products, but the
    // v_cvt_pk_f16_f32 + v_pk_add_f16 of every corner are gone (wrong values): what do those two instructions cost in situ?
    {
      float lo, hi;
      const float neg_zero = -0.0f;
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(lo) : "v"(w), "v"(v[c]), "v"(neg_zero));
      asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(hi) : "v"(w), "v"(v[c]), "v"(neg_zero));
      diag_acc_lo += lo;  // (one full-rate v_add_f32 each keeps the products alive)
      diag_acc_hi += hi;
    }
#else
    acc = acc + weight_times_entry(w, v[c]);  // v_pk_add_f16, RNE: result += (T)(weight * data)
#endif
#ifdef NRF_DIAG_EXTRA_INTERP
    // diagnostic build: NRF_DIAG_EXTRA_INTERP extra v_cvt_pk_f16_f32 per corner whose results nobody reads (right values)
#pragma unroll
    for (int e = 0; e < NRF_DIAG_EXTRA_INTERP; ++e) {
      uint32_t sink;
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(sink) : "v"(w), "v"(frac[e % 3]));
    }
#endif
  }
#ifdef NRF_DIAG_DROP_INTERP_CVT
  return pack_h2(diag_acc_lo, diag_acc_hi);
#else
  return h2_bits(acc);
#endif
}

// ---- grids with F = 4 / 8 features per level (tcnn's n_features_per_level; T/.../grid.h:1365-1386): an entry is 8 / 16 bytes, a
// corner ONE aligned 8- / 16-byte MUBUF gather (the texture path charges per lane address, not per byte: profiles/r02/
// gather_probe.txt), DW = F / 2 packed half2 accumulators per level.  Same arithmetic per feature as level_interp (grid.h:236-262).
template <int UNI, int DW>
__device__ __forceinline__ void level_gather_wide(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                                  float py, float pz, uint32_t (&v)[8 * DW], float (&frac)[3]) {
  static_assert(DW == 2 || DW == 4, "F = 4 or 8");
  uint32_t off[8];
  level_offsets<UNI, DW == 2 ? 3 : 4>(L, px, py, pz, off, frac);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if constexpr (DW == 2) {
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[c], 0, 0);
      v[2 * c] = w.x;
      v[2 * c + 1] = w.y;
    } else {
      const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[c], 0, 0);
      v[4 * c] = w.x;
      v[4 * c + 1] = w.y;
      v[4 * c + 2] = w.z;
      v[4 * c + 3] = w.w;
    }
  }
}

template <int DW>
__device__ __forceinline__ void level_interp_wide(const uint32_t (&v)[8 * DW], const float (&frac)[3], uint32_t (&out)[DW]) {
  const float wx[2] = {1 - frac[0], frac[0]};
  const float wy[2] = {1 - frac[1], frac[1]};
  const float wz[2] = {1 - frac[2], frac[2]};
  half2_t acc[DW];
#pragma unroll
  for (int e = 0; e < DW; ++e) acc[e] = half2_t{(half_t)0.0f, (half_t)0.0f};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float w = (wx[c & 1] * wy[(c >> 1) & 1]) * wz[(c >> 2) & 1];  // ((1 * wx) * wy) * wz in dimension order (grid.h:240-252)
#pragma unroll
    for (int e = 0; e < DW; ++e) acc[e] = acc[e] + weight_times_entry(w, v[DW * c + e]);  // result += (T)(weight * data), fp16 RNE
  }
#pragma unroll
  for (int e = 0; e < DW; ++e) out[e] = h2_bits(acc[e]);
}

// ---- grids with ONE feature per level (F = 1): an entry is a single half, a corner a 2-byte gather (buffer_load_ushort) at the
// same index arithmetic with a shift of 1.  The value rides in the LOW half of a dword whose high half is zero, so level_interp
// serves unchanged: its high accumulator stays +0 and the low one is the feature, rounded as grid.h:236-262 rounds it.
template <int UNI>
__device__ __forceinline__ void level_gather_f1(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                                float py, float pz, uint32_t (&v)[8], float (&frac)[3]) {
  uint32_t off[8];
  level_offsets<UNI, 1>(L, px, py, pz, off, frac);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, off[c], 0, 0);
}
template <int UNI>
__device__ __forceinline__ uint32_t level_nearest_f1(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                                     float py, float pz) {
  uint32_t off[8];
  float frac[3];
  level_offsets<UNI, 1>(L, px, py, pz, off, frac);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, off[0], 0, 0);
}

// InterpolationType::Nearest (T/.../grid.h:215-232): a level's features ARE the entry at floor(pos) -- one aligned gather of the
// whole entry (4 / 8 / 16 bytes for F = 2 / 4 / 8), no weights, no arithmetic on the values.  The index is corner 0 of
// level_offsets (the other seven are dead code here).
template <int UNI, int DW>
__device__ __forceinline__ void level_nearest(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L, float px,
                                              float py, float pz, uint32_t (&out)[DW]) {
  static_assert(DW == 1 || DW == 2 || DW == 4, "F = 2, 4 or 8");
  uint32_t off[8];
  float frac[3];
  level_offsets<UNI, DW == 1 ? 2 : (DW == 2 ? 3 : 4)>(L, px, py, pz, off, frac);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(grid), 0, grid_bytes, 0x00020000);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if constexpr (DW == 1) {
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[0], 0, 0);
  } else if constexpr (DW == 2) {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[0], 0, 0);
    out[0] = w.x;
    out[1] = w.y;
  } else {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[0], 0, 0);
    out[0] = w.x;
    out[1] = w.y;
    out[2] = w.z;
    out[3] = w.w;
  }
}

// Smoothstep interpolation (grid.h InterpolationType::Smoothstep): val * val * (3 - 2 val) on the fractions, T/.../common_device.h:379-381
__device__ __forceinline__ void smoothstep_fractions(float (&frac)[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float f = frac[d], sq = f * f, b = 2.0f * f;
    frac[d] = sq * (3.0f - b);
  }
}

template <int UNI = 0, bool FAST = false>
__device__ __forceinline__ uint32_t encode_level(const uint32_t* __restrict__ grid, uint32_t grid_bytes, const LevelParams L,
                                                 float px, float py, float pz) {
  uint32_t v[8];
  float frac[3];
  level_gather<UNI>(grid, grid_bytes, L, px, py, pz, v, frac);
  return level_interp<FAST>(v, frac);
}

// ------------------------------------------------------ direction encoding ----
// 16 fp16 values per direction (the padded width the rgb network expects).
// d01 = 0.5*d + 0.5 (R/src/nerf_render.cu:313-314).
__device__ __forceinline__ void encode_dir16(const DevModel& M, float d01x, float d01y, float d01z, half_t out[16]) {
  if (M.dir_encoding == NRF_DIR_SH) {
    const float x = d01x * 2.f - 1.f, y = d01y * 2.f - 1.f, z = d01z * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float c[16];
    c[0] = 0.28209479177387814f;
    c[1] = -0.48860251190291987f * y;
    c[2] = 0.48860251190291987f * z;
    c[3] = -0.48860251190291987f * x;
    c[4] = 1.0925484305920792f * xy;
    c[5] = -1.0925484305920792f * yz;
    c[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    c[7] = -1.0925484305920792f * xz;
    c[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    c[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    c[10] = 2.8906114426405538f * xy * z;
    c[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    c[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    c[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    c[14] = 1.4453057213202769f * z * (x2 - y2);
    c[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    // SH pads with LEADING ones up to the alignment of 16 (spherical_harmonics.h:57-64)
#define NRF_SH_CASE(DEG)                                                        \
  case DEG: {                                                                   \
    constexpr int n = DEG * DEG, pad = 16 - n;                                  \
    _Pragma("unroll") for (int j = 0; j < pad; ++j) out[j] = (half_t)1.0f;      \
    _Pragma("unroll") for (int j = 0; j < n; ++j) out[pad + j] = f2h_rne(c[j]);  \
  } break;
    switch (M.sh_degree) {
      NRF_SH_CASE(1)
      NRF_SH_CASE(2)
      NRF_SH_CASE(3)
      default:
      NRF_SH_CASE(4)
    }
#undef NRF_SH_CASE
  } else if (M.dir_encoding == NRF_DIR_FREQUENCY) {
    const float PI = 3.14159265358979323846f;
    const uint32_t nf = M.n_frequencies, raw = 6 * nf;
    const float in[3] = {d01x, d01y, d01z};
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
      float v = 1.0f;  // trailing pad (frequency.h:72-74)
      if (j < raw) {
        const uint32_t feat = j / (nf * 2);
        const uint32_t log2_frequency = (j / 2) % nf;
        const float phase_shift = (float)(j % 2) * (PI / 2);
        const float xin = feat == 0 ? in[0] : (feat == 1 ? in[1] : in[2]);
        const float xs = ldexpf(xin, (int)log2_frequency);
        v = __sinf(xs * PI + phase_shift);
      }
      out[j] = f2h_rne(v);
    }
  } else {  // Identity: scale 1, offset 0, trailing ones
    out[0] = f2h_rne(d01x);
    out[1] = f2h_rne(d01y);
    out[2] = f2h_rne(d01z);
#pragma unroll
    for (uint32_t j = 3; j < 16; ++j) out[j] = (half_t)1.0f;
  }
}

// Wide instance: the eight Frequency-encoding entries e0 .. e0 + 7 (e0 a multiple of 8) of one direction, as the lane's
// B fragment of an extra K step of the first rgb layer.  frequency.h:72-89: entry e belongs to input dimension
// e / (2 nf), octave (e / 2) % nf, phase (e % 2) * pi/2; entries from 6 nf on are the padding ones.  With u = e / 2 < 3 nf
// the division is two compares.
__device__ __forceinline__ half8_t dir_entries8(uint32_t nf, uint32_t e0, float d01x, float d01y, float d01z) {
  const float PI = 3.14159265358979323846f;
  half8_t r;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t u = (e0 >> 1) + (uint32_t)p;
    const uint32_t feat = (u >= nf ? 1u : 0u) + (u >= 2u * nf ? 1u : 0u);
    const uint32_t k = u - feat * nf;
    const float in = feat == 0u ? d01x : (feat == 1u ? d01y : d01z);
    const float xs = ldexpf(in, (int)k) * PI;
    const bool pad = u >= 3u * nf;
    r[2 * p] = pad ? (half_t)1.0f : f2h_rne(__sinf(xs));
    r[2 * p + 1] = pad ? (half_t)1.0f : f2h_rne(__sinf(xs + PI / 2));
  }
  return r;
}

// ------------------------------------------------------------- fused MLP ----
// Both MLPs of NerfNetwork::inference_mixed_precision_impl (nerf_network.h:148-196)
// for NT tiles of 16 samples held by ONE wavefront, entirely in registers:
//   Y[out][sample] = W[out][k] X[k][sample]  on v_mfma_f32_16x16x32_f16,
//   A = weight fragment (from LDS), B = activations, D = 16 outs x 16 samples.
// Lane l = (g = l>>4, c = l&15) holds sample c of each tile.  A layer's D
// fragment (lane holds outs 16m+4g+r) is re-packed in-lane as the next
// layer's B fragment; the K permutation this implies is baked into the weight
// fragments (pack_fragments), so no activation ever crosses lanes or LDS.
__device__ __forceinline__ float4_t mfma16(half8_t a, half8_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ half8_t frag_load(const uint4* wl, int f, int lane) {
  const uint4 v = wl[f * 64 + lane];
  return __builtin_bit_cast(half8_t, v);
}

// pack two D fragments (after the hidden activation, rounded to fp16) into one B fragment.
// The activations of this instance are compile-time constants (hidden ReLU, outputs None, sigma Exponential: the
// reference's base.json); any other combination runs in the generic instance (nrf_generic.h).
// ReLU commutes with the (monotonic, sign-preserving) rounding to fp16, so it is applied
// to the packed halves: one v_pk_max_f16 per two values instead of two v_max_f32 per value.
__device__ __forceinline__ half8_t pack_acc(float4_t lo, float4_t hi) {
#ifdef NRF_PROBE_NO_PACK  // scripts/mlp_probe only: the MFMA chain with NO re-packing work (wrong values, same dependencies)
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(half8_t, __builtin_bit_cast(u4, lo) ^ __builtin_bit_cast(u4, hi));  // 4 full-rate VALU instead of 8 half-rate
#endif
  half8_t r;
  r[0] = (half_t)lo[0]; r[1] = (half_t)lo[1]; r[2] = (half_t)lo[2]; r[3] = (half_t)lo[3];
  r[4] = (half_t)hi[0]; r[5] = (half_t)hi[1]; r[6] = (half_t)hi[2]; r[7] = (half_t)hi[3];
  const half8_t zero8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  return __builtin_elementwise_max(r, zero8);
}

// feat[n]  : B fragment of the density MLP input  (hash features 8g..8g+7 of sample c, tile n)
// dirf[n]  : 4 halves = dir-encoding entries 4g..4g+3 of that sample
// out[n]   : valid in lanes g == 0: (r, g, b, sigma) as fp32 values of the fp16 outputs
// frag(f) returns weight fragment f (FRAG_*) of this lane: from LDS (LdsFrags) or from registers.
// In LDS the fragments 0 .. N_FRAGS - 1 are followed directly by the wide instance's FRAG_R0X ones.
struct LdsFrags {
  const uint4* wl;
  int lane;
  __device__ __forceinline__ half8_t operator()(int f) const { return frag_load(wl, f >= FRAG_R0X ? f - FRAG_R0X + N_FRAGS : f, lane); }
};
struct LdsFragsPlain {  // the other widths (MlpShape<W>): fragment f of the LDS copy, no wide-instance remapping
  const uint4* wl;
  int lane;
  __device__ __forceinline__ half8_t operator()(int f) const { return frag_load(wl, f, lane); }
};
// Outputs of mlp_tiles for NT tiles of 16 samples (fp16 values, as the reference's network_output holds them):
//   rg[n], bx[n]  packed halves (r, g) and (b, row 3 of the rgb output) of sample c of tile n -- valid in lanes g == 0
//   sigma         extract_density's activation of density row 0 (nerf_network.h:49-61), for sample c of tile g --
//                 valid in lanes g < NT: the sixteen density values of every tile are first moved into lane row g
//                 (v_permlane16_swap), so that ONE expf sequence serves all tiles of the pass
template <int NT>
struct MlpOut {
  uint32_t rg[NT], bx[NT];
  half_t sigma;
};
// rgb_sigmoid (wave-uniform): the rgb MLP's output activation is Sigmoid instead of None -- tcnn's logistic
// 1 / (1 + expf(-x)) on the fp32 sums (common_device.h:84-88), which is how instant-ngp's colour activation reaches
// this network when one of its snapshots is loaded (nerfhip.py "instant-ngp snapshots").
// RK > 1 (wide instance): dirx[n][s - 1] = B fragment of K step s of the first rgb layer = direction entries
// 32 s - 16 + 8 g .. + 7 of sample c of tile n (dir_entries8).
// (An explicit issue order for this body -- one MFMA, then k vector instructions, by __builtin_amdgcn_sched_group_barrier -- was
//  measured in round 3 and removed: k = 2 equals the compiler's own schedule, k = 3 / 4 are 5-7 % slower,
//  profiles/r03/mlp_interleave.txt.)
template <int NT, int D0_BASE = FRAG_D0, typename Frags = LdsFrags, int RK = 1, int W = 64>
__device__ __forceinline__ void mlp_tiles(const Frags frag, const half8_t (&feat)[NT], const half4_t (&dirf)[NT], MlpOut<NT>& out,
                                          bool rgb_sigmoid = false, const half8_t (*dirx)[RK_WIDE - 1] = nullptr) {
  static_assert(NT == 1 || NT == 2 || NT == 4, "tiles per pass");
  using S = MlpShape<W>;
  constexpr int MT = S::MT, KS = S::KS;
  const float4_t zero = {0.f, 0.f, 0.f, 0.f};
  float4_t acc[NT][MT];
  half8_t hb[NT][KS];
  // a layer's MT accumulator fragments -> the next layer's KS B fragments (W = 16: the upper half of the one step is zero)
  auto repack = [&]() {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int s = 0; s < KS; ++s) hb[n][s] = pack_acc(acc[n][2 * s], 2 * s + 1 < MT ? acc[n][2 * s + 1] : zero);
  };
  // ---- density layer 0: 32 -> W
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const half8_t a = frag(D0_BASE + m);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, feat[n], zero);
  }
  repack();
  // ---- density layer 1: W -> 16
  float4_t dacc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) dacc[n] = zero;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8_t a = frag(S::D1 + s);
#pragma unroll
    for (int n = 0; n < NT; ++n) dacc[n] = mfma16(a, hb[n][s], dacc[n]);
  }
  // density output (fp16), rows 4g..4g+3; rgb input = [density out | dir encoding]
  half8_t rin[NT];
  uint32_t d01[NT];  // packed halves of density rows 4g, 4g + 1 (row 0 = the density itself lives in lanes g == 0)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    half8_t r;
    r[0] = (half_t)dacc[n][0];
    r[1] = (half_t)dacc[n][1];
    r[2] = (half_t)dacc[n][2];
    r[3] = (half_t)dacc[n][3];
    r[4] = dirf[n][0];
    r[5] = dirf[n][1];
    r[6] = dirf[n][2];
    r[7] = dirf[n][3];
    rin[n] = r;
    half2_t lo;
    lo.x = r[0]; lo.y = r[1];
    d01[n] = h2_bits(lo);
  }
  // tile n's lane row 0 -> lane row n of one register (v_permlane16_swap: odd rows of the first operand <-> even rows
  // of the second; v_permlane32_swap: upper half of the first <-> lower half of the second)
  uint32_t dall = d01[0];
  if constexpr (NT >= 2) dall = __builtin_amdgcn_permlane16_swap(d01[0], d01[1], false, false)[0];
  if constexpr (NT == 4) {
    const uint32_t hi = __builtin_amdgcn_permlane16_swap(d01[2], d01[3], false, false)[0];
    dall = __builtin_amdgcn_permlane32_swap(dall, hi, false, false)[0];
  }
  // extract_density: fp32 activation (Exponential) of the fp16 density output, stored as fp16
  out.sigma = (half_t)expf((float)bits_h2(dall).x);
  // ---- rgb layer 0: 32 (or 32 RK) -> W
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const half8_t a = frag(S::R0 + m);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, rin[n], zero);
  }
  if constexpr (RK > 1) {
    static_assert(W == 64, "the wide form exists for 64 neurons");
#pragma unroll
    for (int s = 1; s < RK; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const half8_t a = frag(FRAG_R0X + 4 * (s - 1) + m);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, dirx[n][s - 1], acc[n][m]);
      }
  }
  repack();
  // ---- rgb layer 1: W -> W
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n][m] = zero;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const half8_t a = frag(S::R1 + KS * m + s);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, hb[n][s], acc[n][m]);
    }
  }
  repack();
  // ---- rgb layer 2: W -> 16 (3 used)
#pragma unroll
  for (int n = 0; n < NT; ++n) dacc[n] = zero;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8_t a = frag(S::R2 + s);
#pragma unroll
    for (int n = 0; n < NT; ++n) dacc[n] = mfma16(a, hb[n][s], dacc[n]);
  }
  if (rgb_sigmoid) {
    asm volatile("" ::: "memory");  // keeps this a (wave-uniform) branch: without it the compiler evaluates the six
                                    // logistic functions speculatively and selects -- 130 VALU instructions per pass
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 3; ++r) dacc[n][r] = 1.0f / (1.0f + expf(-dacc[n][r]));
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {  // network_output rows 0..2 (fp16)
    out.rg[n] = pack_h2(dacc[n][0], dacc[n][1]);
    out.bx[n] = pack_h2(dacc[n][2], dacc[n][3]);
  }
}

// mlp_tiles for 64 neurons with a RUNTIME number of hidden layers (the DEPTH instance): xd / xr 64 -> 64 layers between the
// first and the output layer of the density / rgb MLP (base.json: 0 / 1).  The same in-lane D -> B chaining, the same K
// permutation in every 64-wide input (pack_fragments_depth); fragments in the DF_* order.
template <int NT, typename Frags, bool ACTS = false>
__device__ __forceinline__ void mlp_tiles_depth(const Frags frag, const half8_t (&feat)[NT], const half4_t (&dirf)[NT], MlpOut<NT>& out,
                                                bool rgb_sigmoid, uint32_t xd, uint32_t xr, uint32_t act_d = NRF_ACT_RELU,
                                                uint32_t act_r = NRF_ACT_RELU) {
  static_assert(NT == 1 || NT == 2, "tiles per pass");
  constexpr int MT = 4, KS = 2;
  const float4_t zero = {0.f, 0.f, 0.f, 0.f};
  float4_t acc[NT][MT];
  half8_t hb[NT][KS];
  // Hidden activation of the MLP at hand (wave-uniform; round 6): ReLU on the packed halves (pack_acc); Squareplus, Softplus,
  // Sigmoid, Exponential or None (T/include/tiny-cuda-nn/common_device.h:68-114) on the fp32 accumulators (activate_native), then
  // the fp16 store -- the order the oracle's contract has (mlp_one: activation of the fp32 sum, rounded once).
  uint32_t act = act_d;
  auto repack = [&]() {
    if (!ACTS || act == NRF_ACT_RELU) {  // (ACTS = false: the NET_DEPTH instance, ReLU in both MLPs)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int s = 0; s < KS; ++s) hb[n][s] = pack_acc(acc[n][2 * s], acc[n][2 * s + 1]);
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          half8_t r;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r[j] = (half_t)activate_native(act, acc[n][2 * s][j]);
            r[4 + j] = (half_t)activate_native(act, acc[n][2 * s + 1][j]);
          }
          hb[n][s] = r;
        }
    }
  };
  auto hidden = [&](int base) {  // one 64 -> 64 layer on hb, result back in hb
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n][m] = zero;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const half8_t a = frag(base + KS * m + s);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, hb[n][s], acc[n][m]);
      }
    }
    repack();
  };
  // ---- density MLP: 32 -> 64, xd x (64 -> 64), 64 -> 16
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const half8_t a = frag(DF_D0 + m);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, feat[n], zero);
  }
  repack();
  for (uint32_t e = 0; e < xd; ++e) hidden(DF_WW + 8 * (int)e);  // (wave-uniform trip count)
  float4_t dacc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) dacc[n] = zero;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8_t a = frag(DF_D1 + s);
#pragma unroll
    for (int n = 0; n < NT; ++n) dacc[n] = mfma16(a, hb[n][s], dacc[n]);
  }
  half8_t rin[NT];
  uint32_t d01[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    half8_t r;
    r[0] = (half_t)dacc[n][0];
    r[1] = (half_t)dacc[n][1];
    r[2] = (half_t)dacc[n][2];
    r[3] = (half_t)dacc[n][3];
    r[4] = dirf[n][0];
    r[5] = dirf[n][1];
    r[6] = dirf[n][2];
    r[7] = dirf[n][3];
    rin[n] = r;
    half2_t lo;
    lo.x = r[0]; lo.y = r[1];
    d01[n] = h2_bits(lo);
  }
  uint32_t dall = d01[0];
  if constexpr (NT >= 2) dall = __builtin_amdgcn_permlane16_swap(d01[0], d01[1], false, false)[0];
  out.sigma = (half_t)expf((float)bits_h2(dall).x);
  // ---- rgb MLP: 32 -> 64, xr x (64 -> 64), 64 -> 16
  act = act_r;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const half8_t a = frag(DF_R0 + m);
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n][m] = mfma16(a, rin[n], zero);
  }
  repack();
  for (uint32_t e = 0; e < xr; ++e) hidden(DF_WW + 8 * (int)(xd + e));
#pragma unroll
  for (int n = 0; n < NT; ++n) dacc[n] = zero;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8_t a = frag(DF_R2 + s);
#pragma unroll
    for (int n = 0; n < NT; ++n) dacc[n] = mfma16(a, hb[n][s], dacc[n]);
  }
  if (rgb_sigmoid) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 3; ++r) dacc[n][r] = 1.0f / (1.0f + expf(-dacc[n][r]));
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    out.rg[n] = pack_h2(dacc[n][0], dacc[n][1]);
    out.bx[n] = pack_h2(dacc[n][2], dacc[n][3]);
  }
}

}  // namespace nrf
