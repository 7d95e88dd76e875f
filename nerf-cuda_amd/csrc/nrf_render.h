// nrf_render.h -- the tile program of the render hot path and its two schedulings (render_persistent_kernel, render_kernel)
// as templates: every translation unit that instantiates kernel instances (nrf_kernels_*.hip) includes this header, so an edit
// of one family of instances rebuilds that family only and `make -j` compiles the families side by side.
//
//
// The render kernel is the product: ONE launch per frame -- or per batch of up to 128 camera views -- replaces the
// reference's host-driven loop of ~15 launches + a blocking D2H copy per march iteration
// (R/src/nerf_render.cu:269-338).  One wavefront owns one 8x8 pixel tile (64 rays) for the tile's whole life:
//     raygen -> near/far -> { march (ballot/mbcnt sample compaction into LDS)
//                             -> hash-grid gather + SH -> both MLPs on MFMA
//                             -> alpha compositing } until every ray is dead
//     -> background blend -> RGBA / depth store.
// Nothing but the final pixels ever goes to HBM; the only global reads are the hash table, the occupancy bitfield and
// the weight fragments (kept in LDS).  Two schedulings of that tile program (tile_rounds is shared):
//   render_persistent_kernel  one workgroup per CU for the whole launch, tables staged once, waves pull strips from
//                             per-XCD work queues -- the default;
//   render_kernel             one workgroup per strip of 4 tiles, scheduled by the dispatcher -- models whose march
//                             tables do not fit beside the persistent workgroup, and NRF_PERSISTENT=0.
// The stage kernels below it expose the same device functions one stage at a
// time for the parity tests (include/nerfhip.h "stage entry points").

#pragma once
#include <cstddef>
#include <cstdlib>

#include "nrf_device.h"
#include "nrf_generic.h"
#include "nrf_launch.h"

namespace nrf {

// ---------------------------------------------------------------- LDS map ----
#ifndef NRF_SLOTS
#define NRF_SLOTS 64
#endif
constexpr int SLOTS = NRF_SLOTS;  // sample slots a wave fills per round
struct WaveLds {
  union {
    float4 pos[SLOTS];   // sample slot before the network phase: x, y, z (world, clamped), ray lane (bit pattern)
    float4 out[SLOTS];   // sample slot after it: r, g, b, sigma (each lane overwrites only slots it has consumed)
  };
  float2 aux[SLOTS];     // sample slot: dt, composited t
  uint32_t dirf[64][8];  // ray lane: 16 fp16 direction-encoding values
};
static_assert(sizeof(WaveLds) == 2048 + 24 * SLOTS, "WaveLds layout");

constexpr int LDS_WFRAG_BYTES = N_FRAGS * 64 * 16;  // 20480
// network instances of the kernels below
// NET_W16 / NET_W32 / NET_W128: the register-resident instance for the other widths of tcnn's FullyFusedMLP (persistent kernel only)
// NET_WIDE_SH: the wide form for SphericalHarmonics of degree 5..8 (32..64 direction values): the entries beyond the first sixteen
// are computed once per ray into an LDS row (not per sample in-lane as for Frequency) -- persistent kernel only
// NET_DEPTH: 64 neurons with other numbers of hidden layers than base.json's 1 + 2 (mlp_tiles_depth) -- persistent kernel only
// NET_ACT (round 6): the same with hidden activations other than ReLU (Squareplus, Softplus, Sigmoid, Exponential, None: activate_native
// on the fp32 accumulators) -- 12 waves per workgroup: the activation sequences need the registers
// NET_GRID2 / 4 / 8: base.json's MLPs behind another grid -- F = 2 with fewer than 16 levels, F = 4 / 8 with up to 32 features in all,
// Linear or Smoothstep (grid_features) -- persistent kernel only
enum : int { NET_HOT = 0, NET_GENERIC = 1, NET_WIDE = 2, NET_W16 = 3, NET_W32 = 4, NET_W128 = 5, NET_WIDE_SH = 6, NET_DEPTH = 7,
             NET_GRID2 = 8, NET_GRID4 = 9, NET_GRID8 = 10, NET_GRID1 = 11, NET_ACT = 12 };
__host__ __device__ constexpr int net_grid_f(int net) {
  return net == NET_GRID2 ? 2 : (net == NET_GRID4 ? 4 : (net == NET_GRID8 ? 8 : (net == NET_GRID1 ? 1 : 0)));
}
constexpr int SH_ROW_HALVES = 72;                     // a ray's row: up to 64 direction values + 8 halves of padding (rows 4 banks apart)
constexpr int LDS_SHROW_BYTES = 64 * SH_ROW_HALVES * 2;  // 9216 per wave
__host__ __device__ constexpr int net_width(int net) { return net == NET_W16 ? 16 : (net == NET_W32 ? 32 : (net == NET_W128 ? 128 : 64)); }
__host__ __device__ constexpr int net_wfrag_bytes(int net) {  // weight fragments a workgroup keeps in LDS
  return (net == NET_WIDE || net == NET_WIDE_SH) ? (N_FRAGS + 4 * (RK_WIDE - 1)) * 1024
       : net == NET_W16 ? MlpShape<16>::N * 1024 : net == NET_W32 ? MlpShape<32>::N * 1024 : net == NET_W128 ? MlpShape<128>::N * 1024
       : (net == NET_DEPTH || net == NET_ACT) ? DEPTH_FRAGS * 1024 : N_FRAGS * 1024;
}
constexpr int LDS_WFRAG_WIDE_BYTES = (N_FRAGS + 4 * (RK_WIDE - 1)) * 64 * 16;  // 28672: + the extra K steps of the first rgb layer
constexpr int LDS_RAYD_BYTES = 64 * 3 * 4;  // wide instance: 0.5 d + 0.5 of every ray of a wave (fp32)
constexpr int LDS_LEVEL_BYTES = 16 * (int)sizeof(LevelParams);  // 512
constexpr int RENDER_WAVES = 4;  // waves (8x8 pixel tiles) per workgroup
constexpr int RENDER_THREADS = 64 * RENDER_WAVES;
constexpr int LDS_FIXED_BYTES = LDS_WFRAG_BYTES + LDS_LEVEL_BYTES + RENDER_WAVES * (int)sizeof(WaveLds);
constexpr int LDS_TOTAL_BYTES = LDS_WFRAG_BYTES + LDS_LEVEL_BYTES + 4 * (int)sizeof(WaveLds);  // network_kernel (4 waves)
constexpr int LDS_MARCH_TABLE_MAX = 48 * 1024;    // beyond this the tables stay in global memory

// cross-lane hand-off through LDS inside ONE wavefront: LDS operations of a
// wave execute in order, so only the compiler has to be kept from reordering
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Diagnostic build only (make prof, -DNRF_PHASE_TIMING): s_memtime stamps around the phases of a
// round, summed per wave and added to counters[2..6]; the shipped kernel executes no stamp.
#ifdef NRF_PHASE_TIMING
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
// the device-wide constant-rate counter (100 MHz): s_memtime counts core cycles per clock domain, of which a chip has dozens --
// stamps of different compute units are only comparable on this one
__device__ __forceinline__ unsigned long long stamp_rt() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define NRF_STAMP(var) const unsigned long long var = stamp()
#define NRF_ACC(acc, a, b) acc += (b) - (a)
#else
#define NRF_STAMP(var)
#define NRF_ACC(acc, a, b)
#endif


// Encodes and evaluates the S (<= 16*NT) samples queued in the wave's LDS
// slots; results go to W->out[slot].  Lane (g, c): sample c of each tile,
// hash levels {g, 4+g, 8+g, 12+g}, direction entries 4g..4g+3.
// RK > 1 (wide instance): rayd = the wave's ray directions; the direction entries beyond the first sixteen are
// evaluated here, per sample, as B fragments of the first rgb layer's extra K steps.
// SHROWS (NET_WIDE_SH): those entries come from the ray's LDS row instead (rows = the wave's rows, SH_ROW_HALVES apart).
// The lane's 8 grid features (4 dwords) of a GRID instance: lane group g encodes the levels {g, 4 + g, ...} it can hold -- 4 levels
// of F = 2, 2 of F = 4, 1 of F = 8 -- features in tcnn's order level-major; levels the grid does not have are zero (the padding of
// the grid encoding is ZERO, grid.h:959-969).  K order of the first density layer: nrf_api.hip pack_fragments_grid.
// F = 1 (NET_GRID1, round 5): lane group g holds the ONE feature of each of its levels {g, 4 + g, 8 + g, 12 + g} -- four halves, the
// lane's other four features are zero columns (feat_w = 16: pack_fragments_grid) --, two levels to a dword, two levels in flight.
__device__ __forceinline__ void grid_features_f1(const DevModel& M, const LevelParams* lvs, float px, float py, float pz, int g, uint32_t (&fb)[4]) {
#pragma unroll
  for (int j0 = 0; j0 < 4; j0 += 2) {
    uint32_t gv[2][8];
    float gf[2][3];
    uint32_t r[2] = {0u, 0u};
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const int jl = j0 + jb;
      const uint32_t lv = (uint32_t)(4 * jl + g);
      if (lv < M.n_levels) {
        const LevelParams L = lvs[lv];
        const uint32_t uni = (M.uni_modes >> (2 * jl)) & 3u;
        if (M.grid_nearest) {
          r[jb] = uni == 2u ? level_nearest_f1<2>(M.grid, M.grid_bytes, L, px, py, pz) : level_nearest_f1<0>(M.grid, M.grid_bytes, L, px, py, pz);
        } else {
          if (uni == 2u) level_gather_f1<2>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
          else level_gather_f1<0>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
          if (M.grid_smooth) smoothstep_fractions(gf[jb]);
        }
      }
    }
    if (!M.grid_nearest) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const uint32_t lv = (uint32_t)(4 * (j0 + jb) + g);
        if (lv < M.n_levels) r[jb] = level_interp<false>(gv[jb], gf[jb]);
      }
    }
    fb[j0 >> 1] = (r[0] & 0xffffu) | (r[1] << 16);
  }
}

template <int GF>
__device__ __forceinline__ void grid_features_f248(const DevModel& M, const LevelParams* lvs, float px, float py, float pz, int g, uint32_t (&fb)[4]);
template <int GF>
__device__ __forceinline__ void grid_features(const DevModel& M, const LevelParams* lvs, float px, float py, float pz, int g, uint32_t (&fb)[4]) {
  if constexpr (GF == 1) grid_features_f1(M, lvs, px, py, pz, g, fb);
  else grid_features_f248<GF>(M, lvs, px, py, pz, g, fb);
}
template <int GF>
__device__ __forceinline__ void grid_features_f248(const DevModel& M, const LevelParams* lvs, float px, float py, float pz, int g, uint32_t (&fb)[4]) {
  constexpr int LPL = 8 / GF, DW = GF / 2;  // levels per lane, dwords per entry
  if (M.grid_nearest) {  // wave-uniform.  Nearest (grid.h:215-232): one gather per level, the entry is the result
#pragma unroll
    for (int jl = 0; jl < LPL; ++jl) {
      const uint32_t lv = (uint32_t)(4 * jl + g);
      if (lv < M.n_levels) {
        const LevelParams L = lvs[lv];
        const uint32_t uni = (M.uni_modes >> (2 * jl)) & 3u;
        uint32_t o[DW];
        if (uni == 2u) level_nearest<2, DW>(M.grid, M.grid_bytes, L, px, py, pz, o);
        else level_nearest<0, DW>(M.grid, M.grid_bytes, L, px, py, pz, o);
#pragma unroll
        for (int e = 0; e < DW; ++e) fb[DW * jl + e] = o[e];
      }
    }
    return;
  }
  // F = 2 holds four levels per lane: their gathers go out in two batches of two (16 dwords in flight, not 32).  All four at
  // once -- masked per level, with two index forms side by side -- spilled 8 to 21 VGPRs in these instances; a grid of up
  // to eight levels never reaches the second batch.
  constexpr int BATCH = GF == 2 ? 2 : LPL;
#pragma unroll
  for (int j0 = 0; j0 < LPL; j0 += BATCH) {
    uint32_t gv[BATCH][8 * DW];
    float gf[BATCH][3];
#pragma unroll
    for (int jb = 0; jb < BATCH; ++jb) {
      const int jl = j0 + jb;
      const uint32_t lv = (uint32_t)(4 * jl + g);
      if (lv < M.n_levels) {  // (masked lanes cost the texture path nothing)
        const LevelParams L = lvs[lv];
        const uint32_t uni = (M.uni_modes >> (2 * jl)) & 3u;  // the group's existing levels are all dense (1) / all hashed (2): wave-uniform
        if constexpr (GF == 2) {
          if (uni == 2u) level_gather<2>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
          else level_gather<0>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
        } else {
          if (uni == 2u) level_gather_wide<2, DW>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
          else if (uni == 1u) level_gather_wide<1, DW>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
          else level_gather_wide<0, DW>(M.grid, M.grid_bytes, L, px, py, pz, gv[jb], gf[jb]);
        }
        if (M.grid_smooth) smoothstep_fractions(gf[jb]);  // wave-uniform
      }
    }
#pragma unroll
    for (int jb = 0; jb < BATCH; ++jb) {
      const int jl = j0 + jb;
      const uint32_t lv = (uint32_t)(4 * jl + g);
      if (lv < M.n_levels) {
        if constexpr (GF == 2) fb[jl] = level_interp<false>(gv[jb], gf[jb]);
        else {
          uint32_t o[DW];
          level_interp_wide<DW>(gv[jb], gf[jb], o);
#pragma unroll
          for (int e = 0; e < DW; ++e) fb[DW * jl + e] = o[e];
        }
      }
    }
  }
}

// One step (levels 4 jl .. 4 jl + 3, this lane's: 4 jl + g) of a sample's grid gathers, in the form nrf_load_model chose for the step
template <int RK, bool SHROWS>
__device__ __forceinline__ void gather_step(const DevModel& M, const LevelParams* lvs, int jl, int g, float px, float py, float pz,
                                            uint32_t (&v)[8], float (&fr)[3]) {
  const LevelParams L = lvs[4 * jl + g];
  const uint32_t uni = (M.uni_modes >> (2 * jl)) & 3u;
  if ((M.quad_mask >> (4 * jl)) & 1u) {  // wave-uniform: the step's four levels are gathered from their cell-major quad copies
    // (NET_WIDE -- Frequency directions evaluated per sample in-lane -- has no registers left for the far form's 64-bit
    // addresses: nrf_load_model grants a wide model no far copies)
    constexpr bool FARQ = !(RK > 1 && !SHROWS);
    if (FARQ && ((M.quad_far >> jl) & 1u)) level_gather_quad_far(M.grid, L, px, py, pz, v, fr);
    else level_gather_quad(M.grid, M.grid_bytes, L, px, py, pz, v, fr);
  }
  else if (uni == 2u) level_gather<2>(M.grid, M.grid_bytes, L, px, py, pz, v, fr);
  else if (uni == 1u) level_gather<1>(M.grid, M.grid_bytes, L, px, py, pz, v, fr);
  else level_gather<0>(M.grid, M.grid_bytes, L, px, py, pz, v, fr);
}
__device__ __forceinline__ void sample_pos01(const DevModel& M, const float4 p, float& px, float& py, float& pz) {
  // xyz -> [0,1]: linear_transformer(1/(2 bound), 0.5), R/src/nerf_render.cu:311-312
  if (M.pos_w_pow2) {  // wave-uniform: the product cannot round, so the fma equals multiply-then-add
    px = __builtin_fmaf(M.pos_w, p.x, 0.5f);
    py = __builtin_fmaf(M.pos_w, p.y, 0.5f);
    pz = __builtin_fmaf(M.pos_w, p.z, 0.5f);
  } else {
    px = M.pos_w * p.x; px = px + 0.5f;
    py = M.pos_w * p.y; py = py + 0.5f;
    pz = M.pos_w * p.z; pz = pz + 0.5f;
  }
}

template <int NT, int RK = 1, bool FAST = false, int WD = 64, bool SHROWS = false, int DEPTH = 0, int GF = 0>
__device__ __forceinline__ void network_from_lds(const DevModel& M, const uint4* wl, const LevelParams* lvs, WaveLds* W,
                                                 const float* rayd, int S, int base, int lane, float density_scale,
                                                 const half_t* rows = nullptr) {
  const int g = lane >> 4, c = lane & 15;
  half8_t feat[NT];
  half4_t dirf[NT];
  half8_t dirx[NT][RK_WIDE - 1];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int slot = base + 16 * n + c;
    uint32_t fb[4] = {0u, 0u, 0u, 0u};
    uint2 db = make_uint2(0u, 0u);
    if (slot < S) {
      const float4 p = W->pos[slot];
      float px, py, pz;
      sample_pos01(M, p, px, py, pz);
      // lane group g encodes levels {g, 4+g, 8+g, 12+g}: for each unrolled step jl the four groups work
      // on four ADJACENT levels, which for the usual tables are all dense (jl = 0) or all hashed
      // (jl >= 2), so the index arithmetic is specialised per step by a wave-uniform branch.
      // all 32 gathers of the sample go out before the first one is consumed
      if constexpr (GF != 0) {
        grid_features<GF>(M, lvs, px, py, pz, g, fb);
      } else {
      uint32_t gv[4][8];
      float gf[4][3];
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) gather_step<RK, SHROWS>(M, lvs, jl, g, px, py, pz, gv[jl], gf[jl]);
#pragma unroll
      for (int jl = 0; jl < 4; ++jl) fb[jl] = level_interp<FAST>(gv[jl], gf[jl]);
      }
      const int ray = __builtin_bit_cast(int, p.w);
      db = *reinterpret_cast<const uint2*>(&W->dirf[ray][2 * g]);
      if constexpr (RK > 1 && SHROWS) {
        const half8_t z8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
        for (int s = 1; s < RK; ++s) {
          const uint32_t e0 = 32u * s - 16u + 8u * (uint32_t)g;  // first direction entry of this lane's B fragment
          dirx[n][s - 1] = e0 < M.dir_w ? *reinterpret_cast<const half8_t*>(rows + (size_t)ray * SH_ROW_HALVES + e0) : z8;  // (zero weights beyond)
        }
      } else if constexpr (RK > 1) {
        const float dx = rayd[3 * ray], dy = rayd[3 * ray + 1], dz = rayd[3 * ray + 2];
#pragma unroll
        for (int s = 1; s < RK; ++s) dirx[n][s - 1] = dir_entries8(M.n_frequencies, 32u * s - 16u + 8u * (uint32_t)g, dx, dy, dz);
      }
    } else if constexpr (RK > 1) {
      const half8_t z8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
      for (int s = 1; s < RK; ++s) dirx[n][s - 1] = z8;
    }
    const uint4 fv = make_uint4(fb[0], fb[1], fb[2], fb[3]);
    feat[n] = __builtin_bit_cast(half8_t, fv);
    dirf[n] = __builtin_bit_cast(half4_t, db);
  }
  MlpOut<NT> o;
  if constexpr (DEPTH == 2) mlp_tiles_depth<NT, LdsFragsPlain, true>(LdsFragsPlain{wl, lane}, feat, dirf, o, M.rgb_output_activation == NRF_ACT_SIGMOID, M.depth_xd,
                                                                     M.depth_xr, M.density_activation, M.rgb_activation);
  else if constexpr (DEPTH == 1) mlp_tiles_depth<NT>(LdsFragsPlain{wl, lane}, feat, dirf, o, M.rgb_output_activation == NRF_ACT_SIGMOID, M.depth_xd, M.depth_xr);
  else if constexpr (WD == 64) mlp_tiles<NT, FRAG_D0, LdsFrags, RK>(LdsFrags{wl, lane}, feat, dirf, o, M.rgb_output_activation == NRF_ACT_SIGMOID, dirx);
  else mlp_tiles<NT, 0, LdsFragsPlain, 1, WD>(LdsFragsPlain{wl, lane}, feat, dirf, o, M.rgb_output_activation == NRF_ACT_SIGMOID);
  if (g == 0) {  // decompose_network_in_and_out (render_utils.h:308-334): fp16 rows 0..2 -> fp32 rgb
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int slot = base + 16 * n + c;
      const half2_t rg = bits_h2(o.rg[n]), bx = bits_h2(o.bx[n]);
      if (slot < S) *reinterpret_cast<float3*>(&W->out[slot]) = make_float3((float)rg.x, (float)rg.y, (float)bx.x);
    }
  }
  if (g < NT) {  // lane row g holds the densities of tile g
    const int slot = base + 16 * g + c;
    float sigma = (float)o.sigma;
    if (density_scale != 1.0f) sigma = density_scale * sigma;  // R/src/nerf_render.cu:328 (float multiply)
    if (slot < S) W->out[slot].w = sigma;
  }
}

// At most NT_MAX tiles are evaluated together: NT_MAX = 2 keeps the kernel under 128 VGPRs
// (4 waves per SIMD); the 20 KiB of weight fragments are then read from LDS once per 32 samples.
constexpr int NT_MAX = 2;

// The generic instance's pass (nrf_generic.h): <= 32 of the queued samples, features through LDS rows.
// DENSITY_ONLY: sigma only (density-grid generation); rgb = 0.
template <bool DENSITY_ONLY>
__device__ __forceinline__ void gen_network_from_lds(const DevModel& M, const GenModel& G, const LevelParams* lvs, WaveLds* W,
                                                     const GenLds& Lw, int S, int base, int lane, float density_scale) {
  const int g = lane >> 4, c = lane & 15;
  float p01[GEN_TILES][3];
  bool valid[GEN_TILES];
  int ray[GEN_TILES];
#pragma unroll
  for (int n = 0; n < GEN_TILES; ++n) {
    const int slot = base + 16 * n + c;
    valid[n] = slot < S;
    ray[n] = 0;
    p01[n][0] = p01[n][1] = p01[n][2] = 0.5f;
    if (valid[n]) {
      const float4 p = W->pos[slot];
      float px = M.pos_w * p.x; px = px + 0.5f;  // linear_transformer(1/(2 bound), 0.5), R/src/nerf_render.cu:311-312
      float py = M.pos_w * p.y; py = py + 0.5f;
      float pz = M.pos_w * p.z; pz = pz + 0.5f;
      p01[n][0] = px; p01[n][1] = py; p01[n][2] = pz;
      ray[n] = __builtin_bit_cast(int, p.w) & 63;
    }
  }
  gen_encode_rows(M, G, lvs, Lw, lane, p01, valid);
  if constexpr (!DENSITY_ONLY) {  // direction rows of the pass: lane r < 32 encodes the direction of sample base + r
    const int slot = base + lane;
    if (lane < GEN_SAMPLES && slot < S) {
      const int r = __builtin_bit_cast(int, W->pos[slot].w) & 63;
      gen_encode_dir(M, G, Lw.rayd[3 * r], Lw.rayd[3 * r + 1], Lw.rayd[3 * r + 2], Lw.dir + (size_t)lane * G.dir_stride);
    }
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n) ray[n] = 16 * n + c;
  }
  gen_wave_sync();
  float4_t o[GEN_TILES];
  gen_mlps<DENSITY_ONLY>(M, G, Lw, lane, ray, o);
  if (g == 0) {
#pragma unroll
    for (int n = 0; n < GEN_TILES; ++n) {
      const int slot = base + 16 * n + c;
      float sigma = o[n][3];
      if (density_scale != 1.0f) sigma = density_scale * sigma;
      if (slot < S) W->out[slot] = make_float4(o[n][0], o[n][1], o[n][2], sigma);
    }
  }
  gen_wave_sync();  // the next pass overwrites the rows
}

template <int NET, bool FAST = false>
__device__ __forceinline__ void network_dispatch(const DevModel& M, const uint4* wl, const LevelParams* lvs, WaveLds* W,
                                                 const GenLds& Lw, int S, int lane, float density_scale) {
  constexpr int RK = (NET == NET_WIDE || NET == NET_WIDE_SH) ? RK_WIDE : 1;
  constexpr bool SHR = NET == NET_WIDE_SH;
  if constexpr (NET == NET_GENERIC) {
    const GenModel& G = *M.gen;
    for (int base = 0; base < S; base += GEN_SAMPLES)  // wave-uniform
      gen_network_from_lds<false>(M, G, lvs, W, Lw, S, base, lane, density_scale);
  } else {
    constexpr int WD = net_width(NET);
    constexpr int NTM = WD == 128 ? 1 : NT_MAX;  // 128 neurons: eight accumulator fragments per tile -- one tile per pass
    for (int base = 0; base < S; base += 16 * NTM) {  // wave-uniform
      const int ntile = (S - base + 15) >> 4;
      constexpr int DP = NET == NET_DEPTH ? 1 : (NET == NET_ACT ? 2 : 0);  // (2: runtime hidden activations)
      constexpr int GF = net_grid_f(NET);
      if (ntile <= 1 || NTM == 1) network_from_lds<1, RK, FAST, WD, SHR, DP, GF>(M, wl, lvs, W, Lw.rayd, S, base, lane, density_scale, Lw.dir);
      else network_from_lds<NTM, RK, FAST, WD, SHR, DP, GF>(M, wl, lvs, W, Lw.rayd, S, base, lane, density_scale, Lw.dir);
    }
  }
}

// LDS map of the kernels that evaluate the network.
//   hot instance:     [20 KiB weight fragments][level table][RENDER_WAVES x WaveLds][march tables]
//   wide instance:    [28 KiB weight fragments][level table][RENDER_WAVES x WaveLds][RENDER_WAVES x ray directions][march tables]
//   generic instance: [level table][RENDER_WAVES x WaveLds][RENDER_WAVES x direction rows][RENDER_WAVES x (X, Y)][march tables]
// (the generic instance streams its weights from global memory; WaveLds::dirf holds the density MLP's output and the
//  rays' directions there)
struct LdsMap {
  uint4* wl;            // hot: weight fragments; generic: the (X, Y) region (what the dilated bitfield borrows during ray setup)
  LevelParams* lvs;
  WaveLds* W;           // this wave's block
  GenLds gen;           // this wave's generic regions
  unsigned char* tables;
};
template <int NET>
__device__ __forceinline__ LdsMap lds_map(unsigned char* smem, const DevModel& M, int wave, int n_waves) {
  LdsMap m;
  if constexpr (NET == NET_GENERIC) {
    const GenModel& G = *M.gen;
    m.lvs = reinterpret_cast<LevelParams*>(smem);
    unsigned char* waves = smem + LDS_LEVEL_BYTES;
    m.W = reinterpret_cast<WaveLds*>(waves) + wave;
    unsigned char* dir = waves + n_waves * (int)sizeof(WaveLds);
    unsigned char* act = dir + n_waves * gen_dir_bytes(G);
    m.wl = reinterpret_cast<uint4*>(act);
    m.gen.dens = reinterpret_cast<half_t*>(&m.W->dirf[0][0]);   // 1 KiB: [32][16] halves
    m.gen.rayd = reinterpret_cast<float*>(&m.W->dirf[32][0]);    // 768 B of the second KiB
    m.gen.dir = reinterpret_cast<half_t*>(dir + wave * gen_dir_bytes(G));
    m.gen.X = reinterpret_cast<half_t*>(act + wave * gen_act_bytes(G));
    m.gen.Y = m.gen.X + GEN_SAMPLES * G.act_stride;
    m.gen.wfrag = M.wfrag;
    m.tables = act + n_waves * gen_act_bytes(G);
  } else {
    constexpr int WF = net_wfrag_bytes(NET);
    m.wl = reinterpret_cast<uint4*>(smem);
    m.lvs = reinterpret_cast<LevelParams*>(smem + WF);
    m.W = reinterpret_cast<WaveLds*>(smem + WF + LDS_LEVEL_BYTES) + wave;
    m.gen.dens = m.gen.dir = m.gen.X = m.gen.Y = nullptr;
    m.gen.wfrag = nullptr;
    unsigned char* after = smem + WF + LDS_LEVEL_BYTES + n_waves * (int)sizeof(WaveLds);
    m.gen.rayd = NET == NET_WIDE ? reinterpret_cast<float*>(after + wave * LDS_RAYD_BYTES) : nullptr;
    if constexpr (NET == NET_WIDE_SH) m.gen.dir = reinterpret_cast<half_t*>(after + wave * LDS_SHROW_BYTES);  // the wave's per-ray SH rows
    m.tables = after + (NET == NET_WIDE ? n_waves * LDS_RAYD_BYTES : (NET == NET_WIDE_SH ? n_waves * LDS_SHROW_BYTES : 0));
  }
  return m;
}

// copies the instance's weight fragments into LDS (hot: 0 .. N_FRAGS - 1; wide: followed by FRAG_R0X ..)
template <int NET>
__device__ __forceinline__ void stage_fragments(const DevModel& M, uint4* wl) {
  if constexpr (net_width(NET) != 64 || NET == NET_DEPTH || NET == NET_ACT || net_grid_f(NET) != 0) {  // (GRID: the hot layout, fragments 0 .. N_FRAGS - 1)
    for (int i = threadIdx.x; i < net_wfrag_bytes(NET) / 16; i += blockDim.x) wl[i] = M.wfrag_hot[i];
    return;
  }
  if constexpr (NET == NET_WIDE_SH) {  // the wide layout (fragments 0 .. 19, then FRAG_R0X ..) of a model whose other kernels are generic
    for (int i = threadIdx.x; i < N_FRAGS * 64; i += blockDim.x) wl[i] = M.wfrag_hot[i];
    for (int i = threadIdx.x; i < 4 * (RK_WIDE - 1) * 64; i += blockDim.x) wl[N_FRAGS * 64 + i] = M.wfrag_hot[FRAG_R0X * 64 + i];
    return;
  }
  for (int i = threadIdx.x; i < N_FRAGS * 64; i += blockDim.x) wl[i] = M.wfrag[i];
  if constexpr (NET == NET_WIDE)
    for (int i = threadIdx.x; i < 4 * (RK_WIDE - 1) * 64; i += blockDim.x) wl[N_FRAGS * 64 + i] = M.wfrag[FRAG_R0X * 64 + i];
}

// (unsigned char)(255.0 * x), saturating, NaN -> 0 (R/src/nerf_render.cu:352-359, deviation D-2)
__device__ __forceinline__ unsigned char quant_u8(float v) {
  const double s = 255.0 * (double)v;
  if (!(s > 0.0)) return 0;
  if (s >= 255.0) return 255;
  return (unsigned char)s;
}

// One pixel of a frame.  Three output forms (FrameParams::out_mode):
//   OUT_F32    float planes rgba [px][4], depth [px];
//   OUT_RGBD8  the reference's 8-bit values packed per pixel, r | g << 8 | b << 16 | depth << 24, written where the depth
//              plane would be (nrf_bind_output_rgbd8: what a rank of a multi-GPU step puts on the wire);
//   OUT_U8     the reference's host Image itself (R/src/nerf_render.cu:352-359, common.h:75-89): rgb u8 [px][3] where the
//              rgba plane would be, depth u8 [px] -- row-major frames only (store_tile_u8 below).
// Padding pixels of a shard's tile-major buffer are zero.
__device__ __forceinline__ uint32_t pack_rgbd8(float4 c, float d) {
  return (uint32_t)quant_u8(c.x) | ((uint32_t)quant_u8(c.y) << 8) | ((uint32_t)quant_u8(c.z) << 16) | ((uint32_t)quant_u8(d) << 24);
}
// The planes of view `view` of a launch (the view stride is in pixels whatever a pixel's size is).
struct OutPlanes {
  float4* rgba;  // OUT_F32: rgba; OUT_U8: rgb bytes; OUT_RGBD8: unused
  float* depth;  // OUT_F32: depth; OUT_RGBD8: packed pixels; OUT_U8: depth bytes
};
template <int OUT8 = -1>
__device__ __forceinline__ OutPlanes view_planes(const FrameParams& P, float4* rgba0, float* depth0, int view, unsigned long long stride_px) {
  OutPlanes o;
  const size_t off = (size_t)view * stride_px;
  if (OUT8 == 1 || (OUT8 < 0 && P.out_mode == OUT_U8)) {
    o.rgba = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(rgba0) + 3 * off);
    o.depth = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(depth0) + off);
  } else {
    o.rgba = rgba0 ? rgba0 + off : nullptr;
    o.depth = depth0 + off;
  }
  return o;
}
// The 8-bit planes are written THROUGH the L2 (system-scope stores: global_store ... sc0 sc1): once a wave's s_waitcnt
// vmcnt(0) has returned, the copy engine reads these bytes from memory while the kernel is still running (progress
// reporting of the persistent kernel, tile_written below).  8 MB per frame: the write combining they forgo is not missed.
template <typename T>
__device__ __forceinline__ void store_through(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Progress reporting (FrameParams::prog_*): called by the whole wave after it has stored `count` of the 64 pixels of a
// tile of strip row `row` of view `view` (a tile whose rays were split over several waves is reported in parts).  The wave
// waits for the acknowledgement of its (write-through) stores, then adds its pixels to the row's count; the wave that
// completes the row tells the host (one store into pinned memory).  Every other wave of the row had its stores
// acknowledged before it counted, so whoever sees the flag may read the row's bytes from memory.
__device__ __forceinline__ void tile_written(const FrameParams& P, int view, int row, int lane, unsigned count = 64u) {
  if (P.prog_done == nullptr) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    const size_t i = (size_t)view * P.tiles_y + row;
    const unsigned old = atomicAdd(P.prog_done + i, count);
    if (old + count == 64u * (unsigned)P.tiles_x) __hip_atomic_store(P.prog_flags + i, (unsigned)P.prog_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// OUT_U8, one 8x8 tile by one whole wave (every lane active).  A tile row is 24 rgb bytes and 8 depth bytes: for a tile
// that lies wholly inside a frame whose width is a multiple of 4 the wave assembles them as 6 + 2 aligned dwords per row
// (ds_bpermute / DPP moves of the packed pixels) and writes 48 + 16 dwords with two store instructions instead of 256
// byte stores; any other tile takes the byte stores.
__device__ __forceinline__ void store_tile_u8(const FrameParams& P, unsigned char* rgb8, unsigned char* depth8, int tx, int ty, int lane,
                                              int px, int py, bool in_img, uint32_t v) {
  const bool whole = (P.W & 3) == 0 && tx * 8 + 8 <= P.W && ty * 8 + 8 <= P.H;  // wave-uniform
  if (whole) {
    // (everything below depends on the lane only, i.e. is invariant in the persistent kernel's tile loop: without the
    //  empty asm the compiler hoists it out and keeps seven more VGPRs alive through the march and network phases)
    asm volatile("" : "+v"(lane));
    const int L = lane < 48 ? lane : 0;
    const int r = L / 6, q = L - 6 * r;  // dword q of tile row r: bytes 4q .. 4q + 3 = pixels pa, pa + 1
    const int pa = q + (q >= 3 ? 1 : 0);
    const uint32_t va = (uint32_t)__builtin_amdgcn_ds_bpermute((r * 8 + pa) << 2, (int)v);
    const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((r * 8 + pa + 1) << 2, (int)v);
    const int s = 8 * (q - 3 * (q >= 3 ? 1 : 0));  // 0, 8, 16: the first byte of the dword is channel s / 8 of pixel pa
    const uint32_t w = ((va & 0xffffffu) >> s) | (vb << (24 - s));
    uint32_t dd = v >> 24;
    dd |= (uint32_t)__shfl_down((int)dd, 1) << 8;
    dd |= (uint32_t)__shfl_down((int)dd, 2) << 16;
    if (lane < 48) store_through(reinterpret_cast<uint32_t*>(rgb8 + ((size_t)(ty * 8 + r) * P.W + (size_t)tx * 8) * 3 + 4 * q), w);
    if ((lane & 3) == 0) store_through(reinterpret_cast<uint32_t*>(depth8 + (size_t)py * P.W + px), dd);
  } else if (in_img) {
    const size_t idx = (size_t)py * P.W + px;
    store_through(rgb8 + 3 * idx, (unsigned char)(v & 0xffu));
    store_through(rgb8 + 3 * idx + 1, (unsigned char)((v >> 8) & 0xffu));
    store_through(rgb8 + 3 * idx + 2, (unsigned char)((v >> 16) & 0xffu));
    store_through(depth8 + idx, (unsigned char)(v >> 24));
  }
}
// The same for a tile whose 64 pixels all hold the packed value v (background tiles): no cross-lane traffic.
__device__ __forceinline__ void store_tile_u8_uniform(const FrameParams& P, unsigned char* rgb8, unsigned char* depth8, int tx, int ty, int lane,
                                                      int px, int py, bool in_img, uint32_t v) {
  const bool whole = (P.W & 3) == 0 && tx * 8 + 8 <= P.W && ty * 8 + 8 <= P.H;  // wave-uniform
  if (whole) {
    asm volatile("" : "+v"(lane));  // (as in store_tile_u8)
    const int L = lane < 48 ? lane : 0;
    const int r = L / 6, q = L - 6 * r;
    const int s = 8 * (q - 3 * (q >= 3 ? 1 : 0));
    const uint32_t w = ((v & 0xffffffu) >> s) | (v << (24 - s));
    const uint32_t d = v >> 24, dd = d | (d << 8) | (d << 16) | (d << 24);
    if (lane < 48) store_through(reinterpret_cast<uint32_t*>(rgb8 + ((size_t)(ty * 8 + r) * P.W + (size_t)tx * 8) * 3 + 4 * q), w);
    if ((lane & 3) == 0) store_through(reinterpret_cast<uint32_t*>(depth8 + (size_t)py * P.W + px), dd);
  } else if (in_img) {
    const size_t idx = (size_t)py * P.W + px;
    store_through(rgb8 + 3 * idx, (unsigned char)(v & 0xffu));
    store_through(rgb8 + 3 * idx + 1, (unsigned char)((v >> 8) & 0xffu));
    store_through(rgb8 + 3 * idx + 2, (unsigned char)((v >> 16) & 0xffu));
    store_through(depth8 + idx, (unsigned char)(v >> 24));
  }
}
// called by all 64 lanes of the tile's wave together (tx, ty, k_local are wave-uniform)
// OUT8: -1 = P.out_mode decides at run time (render_kernel); 1 / 0 = the instance is / is not the OUT_U8 one (the persistent
// kernel is at its 128-VGPR limit: with the 8-bit store's cross-lane code inlined at its three store sites the hot instance
// went from 121 to 128 VGPRs and the wide one spilled, so that code lives in instances of its own)
// UNIFORM: all 64 pixels of the tile have this colour (a background tile)
template <int OUT8 = -1, bool UNIFORM = false>
__device__ __forceinline__ void store_pixel(const FrameParams& P, float4* rgba, float* depth, int k_local, int lane, int px, int py,
                                            bool in_img, float4 color, float dn) {
  if (OUT8 == 1 || (OUT8 < 0 && P.out_mode == OUT_U8)) {
    if constexpr (UNIFORM)
      store_tile_u8_uniform(P, reinterpret_cast<unsigned char*>(rgba), reinterpret_cast<unsigned char*>(depth), px >> 3, py >> 3, lane, px, py,
                            in_img, pack_rgbd8(color, dn));
    else
      store_tile_u8(P, reinterpret_cast<unsigned char*>(rgba), reinterpret_cast<unsigned char*>(depth), px >> 3, py >> 3, lane, px, py, in_img,
                    pack_rgbd8(color, dn));
    return;
  }
  if constexpr (OUT8 == 1) return;
  if (!in_img && !P.tile_major) return;
  const size_t idx = P.tile_major ? (size_t)k_local * 64 + lane : (size_t)py * P.W + px;
  if (P.out_mode == OUT_RGBD8) {
    reinterpret_cast<uint32_t*>(depth)[idx] = in_img ? pack_rgbd8(color, dn) : 0u;
  } else {
    rgba[idx] = in_img ? color : make_float4(0.f, 0.f, 0.f, 0.f);
    depth[idx] = in_img ? dn : 0.f;
  }
}

// One pixel by its index in the view's planes (row-major: py * W + px; a shard's tile-major buffer: local tile * 64 + lane):
// the store of a tile whose rays were split over several waves (tail splitting, below), lane by lane.
template <int OUT8>
__device__ __forceinline__ void store_ray_pixel(const FrameParams& P, float4* rgba, float* depth, uint32_t idx, float4 color, float dn) {
  if (OUT8 == 1 || (OUT8 < 0 && P.out_mode == OUT_U8)) {
    const uint32_t v = pack_rgbd8(color, dn);
    unsigned char* rgb8 = reinterpret_cast<unsigned char*>(rgba);
    unsigned char* depth8 = reinterpret_cast<unsigned char*>(depth);
    store_through(rgb8 + 3 * (size_t)idx, (unsigned char)(v & 0xffu));
    store_through(rgb8 + 3 * (size_t)idx + 1, (unsigned char)((v >> 8) & 0xffu));
    store_through(rgb8 + 3 * (size_t)idx + 2, (unsigned char)((v >> 16) & 0xffu));
    store_through(depth8 + idx, (unsigned char)(v >> 24));
    return;
  }
  if constexpr (OUT8 == 1) return;
  if (P.out_mode == OUT_RGBD8) {
    reinterpret_cast<uint32_t*>(depth)[idx] = pack_rgbd8(color, dn);
  } else {
    rgba[idx] = color;
    depth[idx] = dn;
  }
}

// ---- tail splitting (persistent kernel) ----
// One 8x8 tile is one wave's work for its whole life, and the heaviest tiles of a frame are ~1.7 M cycles of strictly
// sequential rounds: a frame rendered alone ends with a handful of waves finishing such tiles while the other 4 000 have
// nothing left to do (1.05 ms for one 1080p view against 0.80 ms per view in batches).  Per-ray semantics do not depend on
// which wave evaluates a ray, so a wave that finds the work queues empty offers itself to the waves of ITS OWN workgroup
// that are still rendering: it sets its bit in `idle_mask` and waits for mail.  A rendering wave looks at the mask at the
// top of every round; if a helper is waiting and at least two of its rays are alive, it claims the helper (clears its
// bit), writes the state of every other live ray into the helper's -- idle -- LDS block, posts the count in the helper's
// mail word and goes on with the rays it kept.  The helper rebuilds the rays (direction encoding included), runs the same
// tile_rounds on them -- donating again if more helpers wait -- stores their pixels one by one and offers itself again.
// `busy` counts the waves that may still produce work (rendering waves, plus a helper from the moment it is claimed):
// a waiting helper leaves when it is zero -- nobody is left who could send mail -- so every wave ends.
struct HelpLds {
  unsigned idle_mask;  // bit w: wave w of the workgroup waits for rays
  int busy;            // waves of the workgroup that are rendering (or about to: claimed helpers)
  unsigned mail[16];   // wave w's mail word: 0 = none, else 0x80000000 | view << 8 | rays (1..32)
};
enum : int { HELP_PIX = 0, HELP_D0, HELP_D1, HELP_D2, HELP_NEAR, HELP_FAR, HELP_FAR_M, HELP_T, HELP_TC, HELP_WS, HELP_DEP, HELP_CR, HELP_CG,
             HELP_CB, HELP_NSAMP, HELP_FIELDS };  // the mailbox: uint32 [HELP_FIELDS][32] in the helper's WaveLds
struct HelpArgs {
  HelpLds* hl;
  void* wave_blocks;  // WaveLds of wave 0 of the workgroup (the blocks are contiguous)
  int view;
};

// What a ray has composited so far, and a wave's statistics (both live in registers).
struct TileAcc {
  float ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
};
struct TileStats {
  unsigned n_samples = 0, n_rounds = 0, n_tile_slots = 0;  // slots: 16-sample MFMA tiles evaluated x 16 (padding included)
  unsigned n_composited = 0;  // samples that reached a ray's compositing sum (= what the reference's per-ray schedule emits: the
                              // samples a ray queues behind its terminating one are evaluated -- n_samples -- but never used)
#ifdef NRF_PHASE_TIMING
  unsigned long long c_march = 0, c_net = 0, c_comp = 0;
  unsigned n_lane_trips = 0, n_wave_iters = 0;
#endif
};

// The rounds of one 8x8 tile (one wave, no workgroup barrier inside): march -> network -> compositing until no ray
// of the tile is alive.  t / tc / alive: the rays' state after ray generation and the visibility walk.
// acc: in = what the rays have composited before (zero for a fresh tile), out = after their last round.
// The number n of a ray for the march's perturb branch (render_utils.h:585-589: pcg32(n, perturb)): its pixel, py * W + px -- what the
// reference's first round uses for every ray (rays_alive starts as the identity).  pix_idx is that number for a row-major
// frame; in the shard layout it is (local tile) * 64 + lane, from which the pixel follows through the strip arithmetic of
// nrf_options (strip id % shard_count == shard_index).
__device__ __forceinline__ uint32_t ray_number(const FrameParams& P, uint32_t pix_idx) {
  if (!P.tile_major) return pix_idx;
  const uint32_t k_local = pix_idx >> 6, l = pix_idx & 63u;
  const uint32_t strips_x = ((uint32_t)P.tiles_x + 3u) >> 2;
  const uint32_t strip = (k_local >> 2) * (uint32_t)P.shard_count + (uint32_t)P.shard_index;
  const uint32_t tx = (strip % strips_x) * 4u + (k_local & 3u), ty = strip / strips_x;
  return (ty * 8u + (l >> 3)) * (uint32_t)P.W + tx * 8u + (l & 7u);
}

// HELP (persistent kernel): tail splitting -- ha names the workgroup's HelpLds; pix_idx / near / far travel with a ray that
// is handed to a helper; *given = the lane's ray was handed over (its pixel is the helper's to store).
// PERTURB: the march's perturb branch (render_utils.h:585-589; nrf_options.perturb > 0).  Dead in the reference (m_perturb = false,
// no setter), so it lives in instances of its own -- render_kernel<.., PERTURB = true>, nrf_kernels_strip.hip -- and costs the
// product path no register: several persistent instances sit at their register limit.
template <int NET, bool COARSE_LDS, int MARCH, bool HELP = false, bool FAST = false, bool PERTURB = false>
__device__ __forceinline__ void tile_rounds(const DevModel& M, const FrameParams& P, const MarchConst& mc, const LdsMap& lm,
                                            const uint32_t* coarse_lds, const float* ctab_lds, int lane, const float (&o)[3],
                                            const float (&d)[3], float rdx, float rdy, float rdz, int sx, int sy, int sz,
                                            float far_m, float t_skip, float t, float tc, bool alive, TileAcc& acc,
                                            TileStats& ts, int n_ray_samples = 0, const HelpArgs* ha = nullptr, uint32_t pix_idx = 0u,
                                            float near = 0.f, float far = 0.f, bool* given_out = nullptr) {
  const uint4* wl = lm.wl;
  const LevelParams* lvs = lm.lvs;
  WaveLds* W = lm.W;
  float ws = acc.ws, dep = acc.dep, cr = acc.cr, cg = acc.cg, cb = acc.cb;
  bool given = false;
#ifdef NRF_PHASE_TIMING
  unsigned long long c_march = 0, c_net = 0, c_comp = 0;
  unsigned n_lane_trips = 0, n_wave_iters = 0;
#endif
  unsigned n_samples = 0, n_rounds = 0, n_tile_slots = 0;
  unsigned n_comp = 0;  // per lane

  while (true) {
    const unsigned long long alive_mask = __ballot(alive);
    if (alive_mask == 0ull) break;
    if constexpr (HELP) {
      // ---- tail splitting: a helper of this workgroup is waiting -> it gets every other live ray
      if (__popcll(alive_mask) >= 2) {
        HelpLds* hl = ha->hl;
        const unsigned im = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&hl->idle_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (im != 0u) {
          const int hw = __builtin_ctz(im);
          unsigned old = 0u;
          if (lane == 0) old = __hip_atomic_fetch_and(&hl->idle_mask, ~(1u << hw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
          if (old & (1u << hw)) {  // claimed (another wave may have been faster)
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(alive_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)alive_mask, 0u));
            const bool give = alive && (rank & 1);
            const unsigned long long gm = __ballot(give);
            const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(gm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)gm, 0u));
            uint32_t* mb = reinterpret_cast<uint32_t*>(reinterpret_cast<WaveLds*>(ha->wave_blocks) + hw);
            if (give) {
              auto put = [&](int f, float v) { mb[f * 32 + slot] = __builtin_bit_cast(uint32_t, v); };
              mb[HELP_PIX * 32 + slot] = pix_idx;
              put(HELP_D0, d[0]); put(HELP_D1, d[1]); put(HELP_D2, d[2]);
              put(HELP_NEAR, near); put(HELP_FAR, far); put(HELP_FAR_M, far_m);
              put(HELP_T, t); put(HELP_TC, tc);
              put(HELP_WS, ws); put(HELP_DEP, dep); put(HELP_CR, cr); put(HELP_CG, cg); put(HELP_CB, cb);
              mb[HELP_NSAMP * 32 + slot] = (uint32_t)n_ray_samples;
            }
            if (lane == 0) {
              __hip_atomic_fetch_add(&hl->busy, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(&hl->mail[hw], 0x80000000u | ((unsigned)ha->view << 8) | (unsigned)__popcll(gm), __ATOMIC_RELEASE,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);  // (release: after the mailbox writes of every lane -- LDS operations of a wave execute in order)
            }
            alive = alive && !give;
            given = given || give;
          }
        }
      }
    }
    NRF_STAMP(t0);
    // ---- march: ballot/mbcnt compaction of the found samples, k-major, into the wave's LDS slots
    unsigned long long slots = 0ull;
    int cnt = 0, S = 0;
    int budget = P.march_budget;
    bool marching = alive;
    bool ended = false;  // t >= far or sample cap: the ray dies after compositing this round's samples
    // A ray whose transmittance is already low will probably terminate within the next samples: whatever it queues behind its
    // terminating sample is evaluated for nothing (the reference's own schedule wastes 8.6 % that way, an unconditional queue
    // of 8 here 2.4 %).  The per-round queue shrinks with T; per-ray semantics -- and so every pixel -- do not depend on it.
    const float Tq = 1.0f - ws;
    int cap = 8;
    if (P.sample_cap == 1) cap = Tq < 0.02f ? 1 : (Tq < 0.1f ? 2 : (Tq < 0.4f ? 4 : 8));
    else if (P.sample_cap == 2) {  // the samples the ray still needs to reach T < 1e-4 (= 2^-13.3) if every one of them halves T
      int e;
      (void)__builtin_frexpf(Tq, &e);
      cap = min(max(e + 13, 1), 8);
    }
    for (int k = 0; k < 8; ++k) {
      const unsigned long long mm = __ballot(marching);
      if (mm == 0ull || S + __popcll(mm) > SLOTS) break;
      float x = 0.f, y = 0.f, z = 0.f, dt = 0.f;
      bool found = false;
#ifdef NRF_PHASE_TIMING
      const int budget_before = budget;
#endif
      float last_t = tc;  // (last_t == composited t)
      if constexpr (PERTURB) {
        // render_utils.h:585-589 with the per-ray loop's n_step == 1: EVERY sample's search starts at rays_t + the ray's shift,
        // and last_t with it; a search that ran out of budget in the round before (t != tc) goes on where it was
        last_t = tc + mc.dt_min * pcg32_first_float((uint64_t)ray_number(P, pix_idx), (uint64_t)(uint32_t)P.perturb);
        if (t == tc) t = last_t;
      }
      if (marching) {
        const int r = COARSE_LDS ? march_next<true, MARCH>(mc, M.occ_bits, coarse_lds, ctab_lds, o[0], o[1], o[2], d[0], d[1], d[2], rdx,
                                                    rdy, rdz, sx, sy, sz, far_m, t_skip, budget, t, x, y, z, dt)
                                 : march_next<false, MARCH_GENERIC>(mc, M.occ_bits, nullptr, M.cell_bound, o[0], o[1], o[2], d[0], d[1], d[2], rdx,
                                                     rdy, rdz, sx, sy, sz, far_m, t_skip, budget, t, x, y, z, dt);
        found = r == MARCH_FOUND;
        marching = found;
        ended = ended || r == MARCH_EXHAUSTED;
      }
#ifdef NRF_PHASE_TIMING
      {
        int used = budget_before - budget;
        n_lane_trips += (unsigned)used;
        for (int o = 32; o; o >>= 1) used = max(used, __shfl_xor(used, o));
        n_wave_iters += (unsigned)used;
      }
#endif
      const unsigned long long fm = __ballot(found);
      if (found) {
        const int slot = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
        const float tn = t + dt;        // march: t += dt
        const float delta = tn - last_t;  // deltas[1] = t - last_t
        tc = tc + delta;                // composite: t += deltas[1]
        t = tc;                         // next march starts from rays_t
        W->pos[slot] = make_float4(x, y, z, __builtin_bit_cast(float, lane));
        W->aux[slot] = make_float2(dt, tc);
        slots |= (unsigned long long)slot << (8 * k);
        cnt++;
        if (n_ray_samples + cnt >= P.max_steps) { marching = false; ended = true; }
        else if (cnt >= cap) marching = false;
      }
      S += __popcll(fm);
    }
    wave_sync();
    NRF_STAMP(t1);

    if (S > 0) {
      // ---- network on the S queued samples (sample-major MFMA tiles)
      network_dispatch<NET, FAST>(M, wl, lvs, W, lm.gen, S, lane, P.density_scale);
      wave_sync();
    }
    NRF_STAMP(t2);

    // ---- alpha compositing, R/include/nerf-cuda/render_utils.h:699-743
    if (alive) {
      bool terminated = false;
      for (int k = 0; k < 8; ++k) {
        if (k >= cnt) break;
        const int slot = (int)((slots >> (8 * k)) & 0xffull);
        const float4 so = W->out[slot];
        const float2 dtc = W->aux[slot];
        const float alpha = 1.0f - __expf(-so.w * dtc.x);
        const float T = 1 - ws;
        const float wgt = alpha * T;
        ws += wgt;
        dep += wgt * dtc.y;  // depth += weight * t, t = composited t of this sample
        cr += wgt * so.x;
        cg += wgt * so.y;
        cb += wgt * so.z;
        n_comp++;
        // `T < 1e-4` against a double literal: true exactly for T <= 9.99999974737875e-05f
        if (T <= 9.99999974737875e-05f) { terminated = true; break; }
      }
      n_ray_samples += cnt;
      alive = !(terminated || ended);
    }
    wave_sync();
    NRF_STAMP(t3);
    NRF_ACC(c_march, t0, t1);
    NRF_ACC(c_net, t1, t2);
    NRF_ACC(c_comp, t2, t3);
    n_samples += (unsigned)S;
    n_rounds++;
    n_tile_slots += (unsigned)((S + 15) & ~15);
  }

  acc.ws = ws; acc.dep = dep; acc.cr = cr; acc.cg = cg; acc.cb = cb;
  if constexpr (HELP) *given_out = given;
  for (int o = 32; o; o >>= 1) n_comp += __shfl_xor(n_comp, o);
  ts.n_composited += n_comp;
  ts.n_samples += n_samples;
  ts.n_rounds += n_rounds;
  ts.n_tile_slots += n_tile_slots;
#ifdef NRF_PHASE_TIMING
  ts.c_march += c_march; ts.c_net += c_net; ts.c_comp += c_comp;
  ts.n_lane_trips += n_lane_trips; ts.n_wave_iters += n_wave_iters;
#endif
}

// ------------------------------------------------------- the render kernel ----
// 256 threads, >= 4 waves per SIMD (four workgroups per CU, 39.9 KB of LDS each): caps the kernel at
// 128 VGPRs.  Small workgroups matter: a workgroup's LDS and wave slots are only released when its
// slowest tile is done.
// (the generic instance is bound by its LDS rows, not by registers: no 128-VGPR cap there)
template <int NET, bool COARSE_LDS, int MARCH, bool PERTURB = false>
__global__ __launch_bounds__(RENDER_THREADS, NET == NET_GENERIC ? 2 : (NET == NET_WIDE ? 3 : 4)) void render_kernel(const DevModel M, const FrameParams P, const ViewBatch VB,
                                                     float4* __restrict__ rgba, float* __restrict__ depth,
                                                     unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool GEN = NET == NET_GENERIC;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const LdsMap lm = lds_map<NET>(smem, M, wave, RENDER_WAVES);
  uint4* wl = lm.wl;
  LevelParams* lvs = lm.lvs;
  WaveLds* W = lm.W;
  // march tables: coarse occupancy bits + cell-boundary table (staged once per workgroup)
  uint32_t* coarse_lds = reinterpret_cast<uint32_t*>(lm.tables);
  float* ctab_lds = reinterpret_cast<float*>(coarse_lds + M.lds_coarse_words);
  // LDS timeline of a workgroup:
  //  (1) nothing: ray generation and the slab test against the box of occupied cells need no table; if no
  //      ray of the four tiles enters the box (background strips, 60-70 % of a typical frame) the
  //      workgroup stores the background and leaves;
  //  (2) level table, march tables and -- borrowed from the weight area, which is not needed yet -- the
  //      dilated coarse occupancy of the per-ray visibility walk (~40 dependent bit lookups per ray: from
  //      global memory that was 26 % of all wave cycles); direction encoding of the surviving rays;
  //  (3) the weight fragments overwrite the borrowed area, unless the walk left no ray alive.
  // (generic instance: the borrowed area is the activation rows, which the first network pass overwrites)
  uint32_t* dil_lds = reinterpret_cast<uint32_t*>(wl);
  const bool use_dil_lds = COARSE_LDS && M.occ_dilated != nullptr && M.lds_dilated_words > 0;

  // Block -> tile-strip order.  Consecutive blocks go to different XCDs (round-robin dispatch); keeping
  // that order interleaves the image over all 8 XCDs at strip granularity, which balances the load
  // (the object covers a few image bands only; an "XCD owns a contiguous band" remap left half of
  // the XCDs idle: 43 % wave-slot occupancy in profiles/r01/pmc_summary_c.txt).
  // A launch renders VB.n_views cameras of the same model / resolution / shard: view-major block
  // order, so the workgroups of view v+1 fill the wave slots that the tail of view v leaves idle.
  // (Alternating the blocks of 2-8 views, each started at a different height of its frame, was
  // measured 2-10 % slower: the views then compete for L1/L2 with disjoint table regions.)
  const int view = (int)blockIdx.x / VB.blocks_per_view;  // wave-uniform (SALU)
  const ViewParams& V = VB.v[view];
  {
    const OutPlanes op = view_planes(P, rgba, depth, view, VB.view_stride_px);
    rgba = op.rgba;  // (NULL: packed 8-bit output, store_pixel)
    depth = op.depth;
  }
  const int swz = (int)blockIdx.x - view * VB.blocks_per_view;
  const int strips_x = (P.tiles_x + 3) >> 2;
  const int k_local = swz * RENDER_WAVES + wave;
  const bool valid_tile = k_local < P.n_local_tiles;  // wave-uniform; padding waves still take the barriers below
  // partition unit = a strip of 4 horizontally adjacent tiles (one workgroup): strip s belongs to
  // rank s % shard_count, local strip index s / shard_count (nerfhip.h nrf_options)
  const int strip = (k_local >> 2) * P.shard_count + P.shard_index;
  const int tx = (strip % strips_x) * 4 + (k_local & 3), ty = strip / strips_x;
  const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
  const bool in_img = valid_tile && px < P.W && py < P.H;
  {
    // (0) the strip's 32x8 pixels against the view's region of interest: outside it every ray misses the box
    // of occupied cells, so the pixel is the background -- no ray is generated (blockIdx-uniform: all four
    // waves of the workgroup take this exit together, before any barrier)
    const int sx0 = (strip % strips_x) * 32, sy0 = ty * 8;
    if (sx0 > V.roi[2] || sx0 + 31 < V.roi[0] || sy0 > V.roi[3] || sy0 + 7 < V.roi[1]) {
      if (valid_tile) store_pixel<-1, true>(P, rgba, depth, k_local, lane, px, py, in_img, make_float4(P.bg_color, P.bg_color, P.bg_color, 0.f), 0.f);
      return;
    }
  }

  NRF_STAMP(t_begin);
#ifdef NRF_PHASE_TIMING
  __shared__ unsigned long long wg_first, wg_last;
  __shared__ unsigned wg_done;
  if (threadIdx.x == 0) {  // ordered before their use by the barrier of step (1)
    wg_first = ~0ull;
    wg_last = 0ull;
    wg_done = 0u;
  }
#endif
  // ---- ray generation + aabb
  const float o[3] = {V.org[0], V.org[1], V.org[2]};
  float d[3];
  ray_dir(V.R, V.cam, px, py, d);
  float near, far;
  near_far(M.aabb, o, d, P.min_near, near, far);
  const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
  const MarchConst mc = march_const(M, P.dt_gamma);
  const int sx = __builtin_signbitf(d[0]) ? 0 : 1;  // copysignf(1, d) > 0: the far face of the cell
  const int sy = __builtin_signbitf(d[1]) ? 0 : 1;
  const int sz = __builtin_signbitf(d[2]) ? 0 : 1;

  // Per-ray semantics (DESIGN.md "Schedule-free compositing"): exactly the reference loop at
  // n_step == 1 -- after every emitted sample the march restarts from the composited t
  // (rays_t = t0 + (t1 - t0), nerf_render.cu:332 -> render_utils.h:716,742), a ray ends when
  // T < 1e-4, when t >= far, or after max_steps samples.  How samples of different rays are
  // batched into rounds therefore cannot change the picture, and the batching below is chosen
  // for the hardware: every round fills up to 64 LDS sample slots (<= 8 per ray), and a lane
  // may spend at most P.march_budget cell trips per round, so that one ray crossing empty space
  // never stalls the other 63 (it simply contributes no sample until it finds one).
  float t = near;    // march position
  float tc = near;   // composited t (t at the last emitted sample)
  bool alive = in_img && (near < far);
  // No sample can lie outside the (inflated) box of occupied cells: rays that miss it are done,
  // the others stop marching where they leave it.  NaN-safe: a 0*inf in the slab test fails `<`.
  float far_m = far;
  float t_skip = near;  // march trips before t_skip test a cell that is known to be empty; near: no such trip
  float t_in, t_out;
  box_interval(M.occ_box, o, rdx, rdy, rdz, t_in, t_out);
  const bool nan = !(t_in == t_in) || !(t_out == t_out);
  {
    const bool hits = (M.occ_box[0] <= M.occ_box[3]) && (t_in <= t_out) && (t_out > near);
    if (t_out < far_m) far_m = t_out;
    if (nan) far_m = far;  // degenerate direction: fall back to the plain aabb range
    alive = alive && (hits || nan);
  }
  const bool wg_live = __syncthreads_or(alive ? 1 : 0) != 0;  // (1): does any ray of the workgroup enter the box?
  if (wg_live) {
    // ---- (2) tables
    if (use_dil_lds)
      for (uint32_t i = threadIdx.x; i < M.lds_dilated_words; i += blockDim.x) dil_lds[i] = M.occ_dilated[i];
    if (COARSE_LDS) {
      for (uint32_t i = threadIdx.x; i < M.lds_coarse_words; i += blockDim.x) coarse_lds[i] = M.occ_coarse[i];
      for (uint32_t i = threadIdx.x; i < M.lds_ctab_floats; i += blockDim.x) ctab_lds[i] = M.cell_bound[i];
    }
    if (threadIdx.x < 16) lvs[threadIdx.x] = M.lv[threadIdx.x];
    __syncthreads();
    // finer, still exact: walk the dilated coarse occupancy between the box entry and exit, one walk per
    // cascade over the ray's stretch inside that cascade's cube (a trip at a position of level k looks up
    // cascade k's grid, and such a position lies inside cube k)
    if (M.occ_dilated != nullptr && alive && !nan) {
      const int Hc = (int)(M.H >> 2);
      const float t_box0 = fmaxf(t_in, near);
      bool any = false;
      float first = far_m, last = t_box0;
      const uint32_t n_casc = MARCH == MARCH_UNIT ? 1u : M.cascade;  // MARCH_UNIT: one cascade, mip_bound 1
      for (uint32_t k = 0; k < n_casc; ++k) {
        const float mb = (n_casc > 1) ? fminf(ldexpf(1.0f, (int)k), M.bound) : fminf(1.0f, M.bound);
        float c_in = t_box0, c_out = far_m;
        if (n_casc > 1) {  // the ray inside cube k
          const float cube[6] = {-mb, -mb, -mb, mb, mb, mb};
          float a, b;
          box_interval(cube, o, rdx, rdy, rdz, a, b);
          if (a == a && b == b) {
            c_in = fmaxf(c_in, a);
            c_out = fminf(c_out, b);
          }
        }
        if (!(c_in < c_out)) continue;
        float t_last, t_first;
        const uint32_t* dl_g = M.occ_dilated + (size_t)k * M.dilated_level_words;
        const uint32_t* dl_l = dil_lds + (size_t)k * M.dilated_level_words;
        const bool vis = use_dil_lds ? coarse_visibility(dl_l, Hc, mb, o, d, rdx, rdy, rdz, c_in, c_out, t_first, t_last)
                                     : coarse_visibility(dl_g, Hc, mb, o, d, rdx, rdy, rdz, c_in, c_out, t_first, t_last);
        if (vis) {
          any = true;
          first = fminf(first, t_first);
          last = fmaxf(last, t_last);
        }
      }
      alive = any;
      if (last < far_m) far_m = last;
      if (any) t_skip = first;
#if NRF_MARCH_FF
      if (any && P.march_ff != 0) {  // (the generic march: any grid size / bound; its table sits in LDS or in global memory)
        const float* tab = COARSE_LDS ? ctab_lds : M.cell_bound;
        if (MARCH == MARCH_UNIT) t = fast_forward_to_barrier(mc, tab, 1.0f, o, d, rdx, rdy, rdz, t, t_skip, far_m);
        else if (MARCH == MARCH_POW2 || mc.C > 1) t = fast_forward_to_barrier_pow2(mc, tab, o, d, rdx, rdy, rdz, t, t_skip, far_m);
        else t = fast_forward_to_barrier(mc, tab, fminf(1.0f, mc.bound), o, d, rdx, rdy, rdz, t, t_skip, far_m);
      }
#endif
    }
    if (alive) {  // direction encoding: only rays that will evaluate the network need it
      float u0 = 0.5f * d[0]; u0 = u0 + 0.5f;  // linear_transformer(0.5, 0.5), nerf_render.cu:313-314
      float u1 = 0.5f * d[1]; u1 = u1 + 0.5f;
      float u2 = 0.5f * d[2]; u2 = u2 + 0.5f;
      if constexpr (GEN) {
        lm.gen.rayd[3 * lane] = u0;  // encoded per pass, for the pass's samples (gen_network_from_lds)
        lm.gen.rayd[3 * lane + 1] = u1;
        lm.gen.rayd[3 * lane + 2] = u2;
      } else {
        half_t e[16];
        encode_dir16(M, u0, u1, u2, e);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          half2_t h;
          h.x = e[2 * j];
          h.y = e[2 * j + 1];
          W->dirf[lane][j] = h2_bits(h);
        }
        if constexpr (NET == NET_WIDE) {  // the entries beyond the first sixteen are evaluated per sample (dir_entries8)
          lm.gen.rayd[3 * lane] = u0;
          lm.gen.rayd[3 * lane + 1] = u1;
          lm.gen.rayd[3 * lane + 2] = u2;
        }
      }
    }
    // ---- (3) the weight fragments replace the dilated bitfield
    if (__syncthreads_or(alive ? 1 : 0) != 0) {
      if constexpr (!GEN) {
        stage_fragments<NET>(M, wl);
        __syncthreads();
      }
    }
  }
  if (!valid_tile) return;  // no barrier after this point
  NRF_STAMP(t_setup_done);
  TileAcc acc;
  TileStats ts;
  // (pix_idx as in the persistent kernel: what ray_number reads for the perturb branch)
  const uint32_t pix_idx = P.tile_major ? (uint32_t)k_local * 64u + (uint32_t)lane : (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
  tile_rounds<NET, COARSE_LDS, MARCH, false, false, PERTURB>(M, P, mc, lm, coarse_lds, ctab_lds, lane, o, d, rdx, rdy, rdz, sx, sy, sz, far_m,
                                                             t_skip, t, tc, alive, acc, ts, 0, nullptr, pix_idx);
  const float ws = acc.ws, dep = acc.dep, cr = acc.cr, cg = acc.cg, cb = acc.cb;
  const unsigned n_samples = ts.n_samples, n_rounds = ts.n_rounds, n_tile_slots = ts.n_tile_slots;
  const unsigned n_composited = ts.n_composited;
#ifdef NRF_PHASE_TIMING
  const unsigned long long c_march = ts.c_march, c_net = ts.c_net, c_comp = ts.c_comp;
  const unsigned n_lane_trips = ts.n_lane_trips, n_wave_iters = ts.n_wave_iters;
#endif

  // ---- get_image_and_depth, R/include/nerf-cuda/render_utils.h:257-264 (depth 0 when the ray missed the aabb)
  {
    const float bgw = (1 - ws) * P.bg_color;
    const float span = far - near;
    const float dn = span > 0.0f ? fmaxf(dep - near, 0.0f) / span : 0.0f;
    store_pixel(P, rgba, depth, k_local, lane, px, py, in_img, make_float4(cr + bgw, cg + bgw, cb + bgw, ws), dn);
  }
  counters += (blockIdx.x % COUNTER_SLOTS) * 16;  // see COUNTER_SLOTS
  if (lane == 0 && n_rounds != 0) {           // waves that never sampled (background) add nothing
    atomicAdd(&counters[0], (unsigned long long)n_samples);
    atomicAdd(&counters[1], (unsigned long long)n_rounds);
    atomicAdd(&counters[11], (unsigned long long)n_tile_slots);
    atomicAdd(&counters[7], (unsigned long long)n_composited);
  }
#ifdef NRF_PHASE_TIMING
  if (lane == 0) {
    NRF_STAMP(t_end);
    // wave slots a workgroup holds until its slowest tile is done: RENDER_WAVES x (last end - first begin) against
    // the sum of the waves' own spans (counters[5])
    atomicMin(&wg_first, t_begin);
    atomicMax(&wg_last, t_end);
    __threadfence_block();
    const unsigned valid_waves = min((unsigned)RENDER_WAVES, (unsigned)(P.n_local_tiles - (k_local - wave)));
    if (atomicAdd(&wg_done, 1u) + 1u == valid_waves) {
      atomicAdd(&counters[12], (unsigned long long)valid_waves * (wg_last - wg_first));
      atomicAdd(&counters[13], 1ull);
    }
    atomicAdd(&counters[2], c_march);
    atomicAdd(&counters[3], c_net);
    atomicAdd(&counters[4], c_comp);
    atomicAdd(&counters[5], t_end - t_begin);
    atomicAdd(&counters[6], 1ull);
    atomicAdd(&counters[9], (unsigned long long)n_wave_iters);
    atomicAdd(&counters[10], t_setup_done - t_begin);
  }
#endif
#ifdef NRF_PHASE_TIMING
  {
    unsigned lt = n_lane_trips;
    for (int o = 32; o; o >>= 1) lt += __shfl_xor(lt, o);
    if (lane == 0) atomicAdd(&counters[8], (unsigned long long)lt);
  }
#endif
}

// Direction encoding of one live ray into the wave's LDS block (what the network phase reads per sample).
template <int NET>
__device__ __forceinline__ void encode_ray_dir(const DevModel& M, const LdsMap& lm, int lane, const float (&d)[3]) {
  float u0 = 0.5f * d[0]; u0 = u0 + 0.5f;  // linear_transformer(0.5, 0.5), nerf_render.cu:313-314
  float u1 = 0.5f * d[1]; u1 = u1 + 0.5f;
  float u2 = 0.5f * d[2]; u2 = u2 + 0.5f;
  if constexpr (NET == NET_GENERIC) {
    lm.gen.rayd[3 * lane] = u0;  // encoded per pass, for the pass's samples (gen_network_from_lds)
    lm.gen.rayd[3 * lane + 1] = u1;
    lm.gen.rayd[3 * lane + 2] = u2;
  } else if constexpr (NET == NET_WIDE_SH) {
    // every SH coefficient of the ray, once (tcnn's padded row: leading ones, then degree^2 values: gen_encode_dir); the
    // first sixteen entries are also what K step 0 of the first rgb layer reads from dirf, as in every other instance
    half_t* row = lm.gen.dir + (size_t)lane * SH_ROW_HALVES;
    gen_encode_dir(M, *M.gen, u0, u1, u2, row);
    const uint32_t* r32 = reinterpret_cast<const uint32_t*>(row);
#pragma unroll
    for (int j = 0; j < 8; ++j) lm.W->dirf[lane][j] = r32[j];
  } else {
    half_t e[16];
    encode_dir16(M, u0, u1, u2, e);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      half2_t h;
      h.x = e[2 * j];
      h.y = e[2 * j + 1];
      lm.W->dirf[lane][j] = h2_bits(h);
    }
    if constexpr (NET == NET_WIDE) {  // the entries beyond the first sixteen are evaluated per sample (dir_entries8)
      lm.gen.rayd[3 * lane] = u0;
      lm.gen.rayd[3 * lane + 1] = u1;
      lm.gen.rayd[3 * lane + 2] = u2;
    }
  }
}

// ---------------------------------------------------- the persistent form ----
// The same tile program as render_kernel with the scheduling turned inside out: ONE workgroup of persist_waves waves per
// CU stays for the whole launch, stages the weight fragments, the level table and every march table (coarse + dilated
// occupancy, cell boundaries) into LDS once, and then every WAVE on its own pulls 8x8 tiles from the work queues
// (below) until they are empty.  Against render_kernel this removes (a) the wave slots a 4-tile workgroup holds until
// its slowest tile is done (7-8 % of the slot time, profiles/r02/phase_timing_wg_hold.txt), (b) the per-strip staging
// of 29 KB (44 KB of tables alone with five cascades) and its three barriers, and (c) the borrowing of the weight area
// by the dilated table.  The queues only hold the strip rows a view's region of interest touches; the rest of the
// frame is background and is filled by a static sweep once a wave finds the queues empty (no atomics: same-address
// device atomics cost ~12 ns each, see COUNTER_SLOTS).
// Exit: a wave leaves its loop when every queue has handed out its last position (the counters only grow, and a
// workgroup visits every class once); no wave waits for another one after the staging barrier, except -- for the
// length of one device atomic -- for the wave of its workgroup that is fetching the next strip.
// waves of the persistent workgroup: what the instance's registers allow per SIMD (x 4 SIMDs) -- hot: <= 128 VGPRs, 4 per
// SIMD; wide and the 128-neuron form: <= 168, 3 (round 5: at 4 per SIMD NET_W128 spilled a register); generic: 3 per SIMD when
// the LDS rows of 12 waves fit beside the march tables, else 2.  NET_WIDE with the generic march runs 8 (WIDE_GENERIC_MARCH_WAVES:
// at 12 it spilled 1-6 registers) -- no shipped instance has scratch (tests/test_abi_cpu.py).
__host__ __device__ constexpr int persist_waves(int net) {
  return net == NET_WIDE_SH ? 8 : ((net == NET_GENERIC || net == NET_WIDE || net == NET_W128 || net == NET_ACT) ? 12 : 16);
}
// (WIDE_GENERIC_MARCH_WAVES: nrf_launch.h -- the host sizes the workgroup with it as well)
constexpr int LDS_CLOCK_BYTES = 16;  // wave 0's entry stamps (core-clock counter, 100 MHz counter), see shader_clock_mhz
constexpr int LDS_QUEUE_BYTES = (MAX_VIEWS + 2) * 4 + 80 + LDS_CLOCK_BYTES;  // q_begin of every view + the total; the workgroup's block counter; HelpLds; the clock stamps
static_assert(sizeof(HelpLds) + LDS_CLOCK_BYTES <= 80 + LDS_CLOCK_BYTES, "HelpLds and the clock stamps live behind the scheduler word");

// The kernel's by-value arguments as they lie in the kernarg segment (the tile loop re-reads them per tile through a
// pointer the compiler cannot see through: otherwise every field of the three structs is hoisted out of the loop and
// kept in -- that is: spilled from -- SGPRs for the whole launch, 177 v_readlane in the march and network loops).
struct PersistArgs {
  DevModel M;
  FrameParams P;
  ViewBatch VB;
};
// (kernel arguments are placed like the members of a struct: each at the next multiple of its alignment)
static_assert(offsetof(PersistArgs, P) == (sizeof(DevModel) + alignof(FrameParams) - 1) / alignof(FrameParams) * alignof(FrameParams) &&
                  offsetof(PersistArgs, VB) % alignof(ViewBatch) == 0,
              "PersistArgs mirrors the kernarg segment of render_persistent_kernel");

// WLDS (generic instance): the layers' weight fragments are staged in LDS as well, instead of streamed from L2 per pass.
// U8: the instance that writes the reference's 8-bit Image layout (OUT_U8, store_tile_u8)
// FAST: nrf_options::fast_interp (opt-in single-rounding interpolation; register-resident instance only) -- instances of
// their own, so that the shipped default symbols keep the bit-exact arithmetic (tests/test_abi_cpu.py checks their ISA)
template <int NET, int MARCH, int WAVES = persist_waves(NET), bool WLDS = false, bool U8 = false, bool FAST = false>
__global__ __launch_bounds__(64 * WAVES, 1) void render_persistent_kernel(const DevModel M0, const FrameParams P0, const ViewBatch VB0,
                                                                              float4* __restrict__ rgba0, float* __restrict__ depth0,
                                                                              unsigned long long* __restrict__ counters,
                                                                              unsigned* __restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  typedef const PersistArgs __attribute__((address_space(4))) * KArgPtr;
  KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const DevModel& M = M0;
  const FrameParams& P = P0;
  const ViewBatch& VB = VB0;
  constexpr int PERSIST_WAVES = WAVES;
  constexpr bool GEN = NET == NET_GENERIC;
  constexpr int OUT8 = U8 ? 1 : 0;
  LdsMap lm = lds_map<NET>(smem, M, wave, PERSIST_WAVES);
  uint32_t* coarse_lds = reinterpret_cast<uint32_t*>(lm.tables);
  float* ctab_lds = reinterpret_cast<float*>(coarse_lds + M.lds_coarse_words);
  uint32_t* dil_lds = reinterpret_cast<uint32_t*>(ctab_lds + M.lds_ctab_floats);
  int* q_lds = reinterpret_cast<int*>(dil_lds + M.lds_dilated_words);
  unsigned* sched = reinterpret_cast<unsigned*>(q_lds + MAX_VIEWS + 1);
  HelpLds* hl = reinterpret_cast<HelpLds*>(sched + 1);
  static_assert(WAVES <= 16, "HelpLds::mail / idle_mask hold 16 waves");
  // ---- staged once per workgroup (= once per CU and launch)
  if constexpr (!GEN) stage_fragments<NET>(M, lm.wl);
  if constexpr (GEN && WLDS) {  // the generic instance's fragments, behind everything else (16-byte aligned)
    uint4* w_lds = reinterpret_cast<uint4*>(smem + ((reinterpret_cast<unsigned char*>(hl + 1) + LDS_CLOCK_BYTES - smem + 15) & ~(size_t)15));
    for (uint32_t i = threadIdx.x; i < M.gen_frag_bytes / 16u; i += blockDim.x) w_lds[i] = M.wfrag[i];
    lm.gen.wfrag = w_lds;
  }
  for (uint32_t i = threadIdx.x; i < M.lds_coarse_words; i += blockDim.x) coarse_lds[i] = M.occ_coarse[i];
  for (uint32_t i = threadIdx.x; i < M.lds_ctab_floats; i += blockDim.x) ctab_lds[i] = M.cell_bound[i];
  for (uint32_t i = threadIdx.x; i < M.lds_dilated_words; i += blockDim.x) dil_lds[i] = M.occ_dilated[i];
  if (threadIdx.x < 16) lm.lvs[threadIdx.x] = M.lv[threadIdx.x];
  if (threadIdx.x <= MAX_VIEWS) {
    const int v = (int)threadIdx.x;
    q_lds[v] = v < VB.n_views ? VB.v[v].q_begin : VB.q_total;
  }
  if (threadIdx.x == 0) *sched = (0xfffffeu << 5) | 4u;  // no strip yet: the first wave to ask fetches one
  if (threadIdx.x == 0) {
    hl->idle_mask = 0u;
    hl->busy = PERSIST_WAVES;
  }
  if (threadIdx.x < 16) hl->mail[threadIdx.x] = 0u;
  __syncthreads();
#ifndef NRF_PHASE_TIMING
  // The shader clock this launch ran at (nrf_stats::shader_clock_mhz; what gather rates per clock are priced with): wave 0 of
  // every workgroup stamps the core-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) here and when
  // it leaves for good; the entry pair waits in LDS (no register is held for it), the two differences are summed over the
  // workgroups in the statistics counters: clock = sum d(memtime) / sum d(memrealtime) x 100 MHz.  Two stamps per launch and
  // workgroup: nothing executes in the tile loop for it.
  unsigned* clk_lds = reinterpret_cast<unsigned*>(hl + 1);
  if (wave == 0) {
    unsigned long long t0, r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    if (lane == 0) {
      clk_lds[0] = (unsigned)t0;
      clk_lds[1] = (unsigned)(t0 >> 32);
      clk_lds[2] = (unsigned)r0;
      clk_lds[3] = (unsigned)(r0 >> 32);
    }
  }
#endif

  const MarchConst mc = march_const(M, P.dt_gamma);
  TileStats ts;
  // ---- the work queues.  Unsharded frames: one queue per XCD class -- class c owns the strip columns c, c + 8, ...
  // of every view, so that the strips an XCD's L2 serves are vertical neighbours (16 KB of L1 per CU and 4 MB of L2 per
  // XCD are where the table lives; with ONE queue, strips landed on random XCDs and the L2 missed twice as often as
  // under render_kernel's static strip -> XCD mapping).  A workgroup starts with the class of its own XCD and moves on
  // to the next class when a queue runs dry, so the XCDs balance at the end.  Sharded frames (a rank's strips are
  // every N-th one): the same with the rank's own strips of a row -- its j-th strip of every row belongs to class
  // j % 8, and the strips of a column are N apart in the rank's numbering whatever the row.
  // A queue entry is one strip (4 tiles); 4 x 4-tile blocks were 2 % slower in 16-view launches and 13 % slower for one
  // view (the 16 tiles of a block at the object's centre are half of a CU's share); groups of 2 or 4 adjacent columns
  // per class measured like single columns, groups of 8 were 2 % slower.
  //   sched (LDS, one word per workgroup) = classes moved past << 29 | queue position << 5 | tiles taken
  const unsigned n_cls = (unsigned)VB0.n_classes, cls_cols = (unsigned)VB0.class_cols, n_units = (unsigned)VB0.q_total;
  const unsigned cls0 = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) % n_cls;  // HW_REG_XCC_ID[3:0]
  constexpr unsigned POS_DONE = 0xffffffu;
  NRF_STAMP(t_loop_begin);
#ifdef NRF_PHASE_TIMING
  unsigned long long c_sched = 0;
  const unsigned long long t_rt_begin = stamp_rt();
#endif
  while (true) {
    NRF_STAMP(t_sched0);
    unsigned pos, bt, moved;
    while (true) {
      unsigned old = 0u;
      if (lane == 0) old = atomicAdd(sched, 1u);
      old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
      moved = old >> 29;
      pos = (old >> 5) & POS_DONE;
      bt = old & 31u;
      if (pos == POS_DONE || bt < 4u) break;
      if (bt == 4u) {  // this wave refills: tile 0 of the new strip is its own
        unsigned nb = POS_DONE;
        for (; moved < n_cls; ++moved) {
          const unsigned cls = (cls0 + moved) % n_cls;
          const unsigned total = n_units * ((cls_cols - cls + n_cls - 1u) / n_cls);
          unsigned got = 0u;
          if (lane == 0) got = atomicAdd(queue + cls, 1u);
          got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
          if (got < total) { nb = got; break; }
        }
        if (nb == POS_DONE) moved = 0u;
        else if (P0.plan_order != nullptr) {  // the launch was planned: the class's nb-th pull renders this position
          unsigned off = 0u;
          const unsigned cls = (cls0 + moved) % n_cls;
          for (unsigned c2 = 0; c2 < cls; ++c2) off += n_units * ((cls_cols - c2 + n_cls - 1u) / n_cls);
          nb = P0.plan_order[off + nb];
        }
        if (lane == 0) __hip_atomic_store(sched, (moved << 29) | (nb << 5) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        pos = nb;
        bt = 0u;
        break;
      }
      // another wave is refilling: wait until the strip changes, then try again
      while ((__hip_atomic_load(sched, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> 5) == (old >> 5)) __builtin_amdgcn_s_sleep(2);
    }
    NRF_STAMP(t_sched1);
    NRF_ACC(c_sched, t_sched0, t_sched1);
    if (pos == POS_DONE) break;
    asm volatile("" : "+s"(ka));  // this tile's argument loads are this tile's (see PersistArgs)
    const DevModel& M = *(const DevModel*)&ka->M;
    const FrameParams& P = *(const FrameParams*)&ka->P;
    const ViewBatch& VB = *(const ViewBatch*)&ka->VB;
    // queue position -> (unit, column), unit -> view.  A unit is a strip row of a view's region of interest; the views'
    // units are numbered one after the other.
    const unsigned cls = (cls0 + moved) % n_cls, ncols = (cls_cols - cls + n_cls - 1u) / n_cls;
    const int u = (int)(pos / ncols), j = (int)(pos - (unsigned)u * ncols);
    static_assert(MAX_VIEWS == 128, "two ballots cover the views");
    const int view = __popcll(__ballot(lane < VB.n_views && u >= q_lds[lane])) +
                     __popcll(__ballot(lane + 64 < VB.n_views && u >= q_lds[lane + 64])) - 1;  // wave-uniform
    const ViewParams& V = VB.v[view];
    const int ul = u - V.q_begin;
    // Centre-out: a view's units are visited in the order c, c + 1, c - 1, c + 2, ... from the middle of the region of
    // interest, where the rays cross the most of the object, so that a view's last tiles tend to be light ones.
    const bool co = P.centre_out != 0;
    auto centre_out = [co](int i, int n) { const int off = (i + 1) >> 1; return co ? (n - 1) / 2 + ((i & 1) ? off : -off) : i; };
    // unit -> strip row of the frame; entry j of the class -> the (cls + n_cls * j)-th of this rank's strips in that row
    const int sxn = (P.tiles_x + 3) >> 2, N = P.shard_count;
    const int row = V.q_row0 + centre_out(ul, V.q_rows);
    const int s_row = row * sxn;                                    // global strips [s_row, s_row + sxn) form the row
    const int ls_first = s_row > P.shard_index ? (s_row - P.shard_index + N - 1) / N : 0;
    const int ls = ls_first + (int)cls + (int)n_cls * j;
    if (ls * N + P.shard_index >= s_row + sxn) continue;            // this row holds fewer of the rank's strips
    const int k_local = ls * 4 + (int)bt;
    if (k_local >= V.k_hi || k_local >= P.n_local_tiles) continue;  // padding of the last strip
    const OutPlanes op = view_planes<OUT8>(P, rgba0, depth0, view, VB.view_stride_px);
    float4* rgba = op.rgba;  // (NULL: packed 8-bit output, store_pixel)
    float* depth = op.depth;
    const int strips_x = (P.tiles_x + 3) >> 2;
    const int strip = (k_local >> 2) * P.shard_count + P.shard_index;
    const int tx = (strip % strips_x) * 4 + (k_local & 3), ty = strip / strips_x;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    const bool in_img = px < P.W && py < P.H;
    const float4 background = make_float4(P.bg_color, P.bg_color, P.bg_color, 0.f);
    // the tile's own 8x8 pixels against the region of interest (render_kernel tests the strip's 32x8)
    if (tx * 8 > V.roi[2] || tx * 8 + 7 < V.roi[0] || ty * 8 > V.roi[3] || ty * 8 + 7 < V.roi[1]) {
      store_pixel<OUT8, true>(P, rgba, depth, k_local, lane, px, py, in_img, background, 0.f);
      if constexpr (U8) if (tx < P.tiles_x) tile_written(P, view, ty, lane);  // (not the padding tiles of a ragged strip)
      continue;
    }
    NRF_STAMP(t_begin);
#ifdef NRF_PHASE_TIMING
    const unsigned samples_before = ts.n_samples;
#endif
    // ---- ray generation + aabb (as render_kernel)
    const float o[3] = {V.org[0], V.org[1], V.org[2]};
    float d[3];
    ray_dir(V.R, V.cam, px, py, d);
    float near, far;
    near_far(M.aabb, o, d, P.min_near, near, far);
    const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
    const int sx = __builtin_signbitf(d[0]) ? 0 : 1;
    const int sy = __builtin_signbitf(d[1]) ? 0 : 1;
    const int sz = __builtin_signbitf(d[2]) ? 0 : 1;
    float t = near, tc = near;
    bool alive = in_img && (near < far);
    float far_m = far, t_skip = near, t_in, t_out;
    box_interval(M.occ_box, o, rdx, rdy, rdz, t_in, t_out);
    const bool nan = !(t_in == t_in) || !(t_out == t_out);
    {
      const bool hits = (M.occ_box[0] <= M.occ_box[3]) && (t_in <= t_out) && (t_out > near);
      if (t_out < far_m) far_m = t_out;
      if (nan) far_m = far;
      alive = alive && (hits || nan);
    }
    if (M.occ_dilated != nullptr && alive && !nan) {  // the visibility walk, on this workgroup's own copy of the table
      const int Hc = (int)(M.H >> 2);
      const float t_box0 = fmaxf(t_in, near);
      bool any = false;
      float first = far_m, last = t_box0;
      const uint32_t n_casc = MARCH == MARCH_UNIT ? 1u : M.cascade;
      for (uint32_t k = 0; k < n_casc; ++k) {
        const float mb = (n_casc > 1) ? fminf(ldexpf(1.0f, (int)k), M.bound) : fminf(1.0f, M.bound);
        float c_in = t_box0, c_out = far_m;
        if (n_casc > 1) {
          const float cube[6] = {-mb, -mb, -mb, mb, mb, mb};
          float a, b;
          box_interval(cube, o, rdx, rdy, rdz, a, b);
          if (a == a && b == b) {
            c_in = fmaxf(c_in, a);
            c_out = fminf(c_out, b);
          }
        }
        if (!(c_in < c_out)) continue;
        float t_last, t_first;
        if (coarse_visibility(dil_lds + (size_t)k * M.dilated_level_words, Hc, mb, o, d, rdx, rdy, rdz, c_in, c_out, t_first, t_last)) {
          any = true;
          first = fminf(first, t_first);
          last = fmaxf(last, t_last);
        }
      }
      alive = any;
      if (last < far_m) far_m = last;
      if (any) t_skip = first;
#if NRF_MARCH_FF
      if (any && P.march_ff != 0) {
        if (MARCH == MARCH_UNIT) t = fast_forward_to_barrier(mc, ctab_lds, 1.0f, o, d, rdx, rdy, rdz, t, t_skip, far_m);
        else if (MARCH == MARCH_POW2 || mc.C > 1) t = fast_forward_to_barrier_pow2(mc, ctab_lds, o, d, rdx, rdy, rdz, t, t_skip, far_m);
        else t = fast_forward_to_barrier(mc, ctab_lds, fminf(1.0f, mc.bound), o, d, rdx, rdy, rdz, t, t_skip, far_m);
      }
#endif
    }
    TileAcc acc;
    bool given = false;  // tail splitting: this lane's ray went to a helper wave, which stores its pixel
    const uint32_t pix_idx = P.tile_major ? (uint32_t)k_local * 64u + (uint32_t)lane : (uint32_t)py * (uint32_t)P.W + (uint32_t)px;
    if (__ballot(alive) != 0ull) {
      if (alive) encode_ray_dir<NET>(M, lm, lane, d);  // direction encoding of the rays that will evaluate the network
      wave_sync();
      NRF_STAMP(t_setup_done);
      const HelpArgs ha = {hl, lm.W - wave, view};
      tile_rounds<NET, true, MARCH, true, FAST>(M, P, mc, lm, coarse_lds, ctab_lds, lane, o, d, rdx, rdy, rdz, sx, sy, sz, far_m, t_skip, t, tc,
                                                    alive, acc, ts, 0, &ha, pix_idx, near, far, &given);
#ifdef NRF_PHASE_TIMING
      if (lane == 0) {
        NRF_STAMP(t_end);
        atomicAdd(&counters[(blockIdx.x % COUNTER_SLOTS) * 16 + 5], t_end - t_begin);
        atomicAdd(&counters[(blockIdx.x % COUNTER_SLOTS) * 16 + 6], 1ull);
        atomicAdd(&counters[(blockIdx.x % COUNTER_SLOTS) * 16 + 10], t_setup_done - t_begin);
      }
#endif
    }
    // ---- get_image_and_depth, R/include/nerf-cuda/render_utils.h:257-264
    const float bgw = (1 - acc.ws) * P.bg_color;
    const float span = far - near;
    float dn = span > 0.0f ? fmaxf(acc.dep - near, 0.0f) / span : 0.0f;
#ifdef NRF_PHASE_TIMING
    if (P.march_budget == 4095) {  // diagnostic: the depth plane carries the tile's cost (cycles / 1e6) and start time instead
      NRF_STAMP(t_tile_end);
      const unsigned hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all 32 bits
      dn = (lane & 7) == 1 ? (float)(t_tile_end - t_begin) * 1e-6f
         : (lane & 7) == 2 ? (float)(ts.n_samples - samples_before) * 1e-3f
         : (lane & 7) == 3 ? (float)bt
         : (lane & 7) == 4 ? (float)((hw_id >> 4) & 3u)   // SIMD_ID
         : (lane & 7) == 5 ? (float)wave
         : (lane & 7) == 6 ? (float)(hw_id & 15u)          // WAVE_ID (slot)
         : (float)(t_begin - t_loop_begin) * 1e-6f;
    }
#endif
    const unsigned long long given_mask = __ballot(given);
    if (given_mask == 0ull) {
      store_pixel<OUT8>(P, rgba, depth, k_local, lane, px, py, in_img, make_float4(acc.cr + bgw, acc.cg + bgw, acc.cb + bgw, acc.ws), dn);
    } else if (!given && (in_img || P.tile_major)) {  // a split tile: the rays this wave kept, lane by lane
      const bool pad = !in_img;  // (padding pixels of a shard's tile-major buffer are zero)
      store_ray_pixel<OUT8>(P, rgba, depth, pix_idx, pad ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(acc.cr + bgw, acc.cg + bgw, acc.cb + bgw, acc.ws),
                            pad ? 0.f : dn);
    }
    if constexpr (U8) if (tx < P.tiles_x) tile_written(P, view, ty, lane, 64u - (unsigned)__popcll(given_mask));
  }

#ifdef NRF_PHASE_TIMING
  if (lane == 0) {  // how long this wave was in the tile loop (sum, count, maximum) and how much of it waiting for a tile
    NRF_STAMP(t_loop_end);
    unsigned long long* cs = counters + (blockIdx.x % COUNTER_SLOTS) * 16;
    atomicAdd(&cs[12], t_loop_end - t_loop_begin);
    atomicAdd(&cs[13], 1ull);
    atomicMax(&cs[14], t_loop_end - t_loop_begin);
    atomicAdd(&cs[15], c_sched);
    // absolute entry / exit stamps of every wave (nrf_debug_wave_times)
    unsigned long long* wt = counters + COUNTER_SLOTS * 16 + 16 + 2 * ((size_t)blockIdx.x * PERSIST_WAVES + wave);
    wt[0] = t_rt_begin;
    wt[1] = stamp_rt();  // (replaced below by the end of the wave's helping)
  }
  unsigned n_helped = 0;
#endif
  // ---- the tiles the queue does not hold are background: a static sweep, one tile per wave and step
  // (skip_outside: the caller fills those rows of the frame itself, nrf_api.hip host frames)
  if (!P.skip_outside) {
    const float4 background = make_float4(P.bg_color, P.bg_color, P.bg_color, 0.f);
    const int n_waves = (int)gridDim.x * PERSIST_WAVES;
    const int strips_x = (P.tiles_x + 3) >> 2;
    for (int view = 0; view < VB.n_views; ++view) {
      const ViewParams& V = VB.v[view];
      const OutPlanes op = view_planes<OUT8>(P, rgba0, depth0, view, VB.view_stride_px);
      float4* rgba = op.rgba;
      float* depth = op.depth;
      const int outside = P.n_local_tiles - (min(V.k_hi, P.n_local_tiles) - V.k_lo);  // tiles before k_lo and from k_hi on
      for (int i = (int)blockIdx.x * PERSIST_WAVES + wave; i < outside; i += n_waves) {
        const int k_local = i < V.k_lo ? i : i + (min(V.k_hi, P.n_local_tiles) - V.k_lo);
        const int strip = (k_local >> 2) * P.shard_count + P.shard_index;
        const int tx = (strip % strips_x) * 4 + (k_local & 3), ty = strip / strips_x;
        const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
        store_pixel<OUT8, true>(P, rgba, depth, k_local, lane, px, py, px < P.W && py < P.H, background, 0.f);
      }
    }
  }
  // ---- tail splitting: this wave has nothing left of its own -- it takes rays off the waves of its workgroup that are
  // still rendering (HelpLds above), until none of them is
  while (P0.tail_split != 0) {
    if (lane == 0) {
      __hip_atomic_fetch_add(&hl->busy, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_or(&hl->idle_mask, 1u << wave, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned mail = 0u;
    while (true) {
      mail = __hip_atomic_load(&hl->mail[wave], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (mail != 0u) break;
      if (__hip_atomic_load(&hl->busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= 0) break;  // nobody is left who could send any
      __builtin_amdgcn_s_sleep(8);
    }
    mail = (unsigned)__builtin_amdgcn_readfirstlane((int)mail);
    if (mail == 0u) break;
    const int n_rays = (int)(mail & 0xffu), view = (int)((mail >> 8) & 0xffu);
    const uint32_t* mb = reinterpret_cast<const uint32_t*>(lm.W);
    const bool mine = lane < n_rays;
    const int ml = mine ? lane : 0;
    auto getf = [&](int f) { return __builtin_bit_cast(float, mb[f * 32 + ml]); };
    const uint32_t pix_idx = mb[HELP_PIX * 32 + ml];
    const float d[3] = {getf(HELP_D0), getf(HELP_D1), getf(HELP_D2)};
    const float near = getf(HELP_NEAR), far = getf(HELP_FAR), far_m = getf(HELP_FAR_M);
    const float t = getf(HELP_T), tc = getf(HELP_TC);
    TileAcc acc;
    acc.ws = getf(HELP_WS); acc.dep = getf(HELP_DEP); acc.cr = getf(HELP_CR); acc.cg = getf(HELP_CG); acc.cb = getf(HELP_CB);
    const int n_ray_samples = (int)mb[HELP_NSAMP * 32 + ml];
    wave_sync();  // every lane has read its ray: the block is this wave's own again
    if (lane == 0) __hip_atomic_store(&hl->mail[wave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" : "+s"(ka));
    const DevModel& M = *(const DevModel*)&ka->M;
    const FrameParams& P = *(const FrameParams*)&ka->P;
    const ViewBatch& VB = *(const ViewBatch*)&ka->VB;
    const ViewParams& V = VB.v[view];
    const float o[3] = {V.org[0], V.org[1], V.org[2]};
    const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
    const int sx = __builtin_signbitf(d[0]) ? 0 : 1;
    const int sy = __builtin_signbitf(d[1]) ? 0 : 1;
    const int sz = __builtin_signbitf(d[2]) ? 0 : 1;
    if (mine) encode_ray_dir<NET>(M, lm, lane, d);
    wave_sync();
    bool given = false;
    const HelpArgs ha = {hl, lm.W - wave, view};
    // (t_skip: the occupancy lookups it would skip are of cells known to be empty -- looking them up changes nothing)
    tile_rounds<NET, true, MARCH, true, FAST>(M, P, mc, lm, coarse_lds, ctab_lds, lane, o, d, rdx, rdy, rdz, sx, sy, sz, far_m, -3.402823466e+38f, t, tc,
                                                  mine, acc, ts, n_ray_samples, &ha, pix_idx, near, far, &given);
    const OutPlanes op = view_planes<OUT8>(P, rgba0, depth0, view, VB.view_stride_px);
    const bool store = mine && !given;
    if (store) {  // get_image_and_depth, as in the tile loop
      const float bgw = (1 - acc.ws) * P.bg_color;
      const float span = far - near;
      const float dn = span > 0.0f ? fmaxf(acc.dep - near, 0.0f) / span : 0.0f;
      store_ray_pixel<OUT8>(P, op.rgba, op.depth, pix_idx, make_float4(acc.cr + bgw, acc.cg + bgw, acc.cb + bgw, acc.ws), dn);
    }
    if constexpr (U8) {  // (8-bit planes are row-major: the strip row of the rays' tile follows from a pixel index)
      const int row = __builtin_amdgcn_readfirstlane((int)(pix_idx / (uint32_t)P.W)) >> 3;
      tile_written(P, view, row, lane, (unsigned)__popcll(__ballot(store)));
    }
#ifdef NRF_PHASE_TIMING
    n_helped += (unsigned)n_rays;
#endif
  }
#ifdef NRF_PHASE_TIMING
  if (lane == 0) {  // when the wave left for good, and how many rays it took off others: nrf_debug_wave_times slot 1, counters[2 .. ] unchanged
    const unsigned long long t_help_end = stamp_rt();
    unsigned long long* wt = counters + COUNTER_SLOTS * 16 + 16 + 2 * ((size_t)blockIdx.x * PERSIST_WAVES + wave);
    wt[1] = (t_help_end << 8) | (unsigned long long)(n_helped > 255u ? 255u : n_helped);  // low byte: rays helped with (capped)
  }
#endif
  counters += (blockIdx.x % COUNTER_SLOTS) * 16;
#ifndef NRF_PHASE_TIMING
  if (wave == 0) {  // (see the entry stamps)
    unsigned long long t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (lane == 0) {
      const unsigned long long t0 = (unsigned long long)clk_lds[0] | ((unsigned long long)clk_lds[1] << 32);
      const unsigned long long r0 = (unsigned long long)clk_lds[2] | ((unsigned long long)clk_lds[3] << 32);
      atomicAdd(&counters[12], t1 - t0);
      atomicAdd(&counters[13], r1 - r0);
    }
  }
#endif
  if (lane == 0 && ts.n_rounds != 0) {
    atomicAdd(&counters[0], (unsigned long long)ts.n_samples);
    atomicAdd(&counters[1], (unsigned long long)ts.n_rounds);
    atomicAdd(&counters[11], (unsigned long long)ts.n_tile_slots);
    atomicAdd(&counters[7], (unsigned long long)ts.n_composited);
#ifdef NRF_PHASE_TIMING
    atomicAdd(&counters[2], ts.c_march);
    atomicAdd(&counters[3], ts.c_net);
    atomicAdd(&counters[4], ts.c_comp);
    atomicAdd(&counters[9], (unsigned long long)ts.n_wave_iters);
#endif
  }
#ifdef NRF_PHASE_TIMING
  {
    unsigned lt = ts.n_lane_trips;
    for (int o = 32; o; o >>= 1) lt += __shfl_xor(lt, o);
    if (lane == 0) atomicAdd(&counters[8], (unsigned long long)lt);
  }
#endif
}

// ---- launch plumbing shared by the instance translation units ----
// Dynamic LDS above 64 KiB has to be announced per kernel (gfx950 has 160 KiB per CU).
template <typename K>
static hipError_t allow_lds(K kernel, int bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// What launch_render (nrf_kernels.hip) has decided for a launch of the persistent form; the instance families live in
// nrf_kernels_hot.hip (base.json shape), nrf_kernels_width.hip (16 / 32 / 128 neurons, other depths), nrf_kernels_wide.hip
// (Frequency / SH directions beyond 16 values), nrf_kernels_generic.hip; render_kernel instances in nrf_kernels_strip.hip.
struct PersistLaunch {
  const DevModel* M;
  const FrameParams* P;
  const ViewBatch* VB;
  void *rgba, *depth, *counters;
  unsigned* queue;
  hipStream_t st;
  int lds, wgs, waves;
  bool unit, pow2;  // march form: one cascade with mip_bound 1 / several cascades with a power-of-two bound (else generic)
};
hipError_t launch_persistent_hot(const PersistLaunch& L);
hipError_t launch_persistent_width(const PersistLaunch& L);
hipError_t launch_persistent_wide(const PersistLaunch& L);
hipError_t launch_persistent_generic(const PersistLaunch& L);
hipError_t launch_persistent_grid(const PersistLaunch& L);
struct StripLaunch {
  const DevModel* M;
  const FrameParams* P;
  const ViewBatch* VB;
  void *rgba, *depth, *counters;
  hipStream_t st;
  int lds, blocks;
  bool lds_tab, unit, pow2;
  bool perturb;  // nrf_options.perturb > 0: the PERTURB instances (tables in global memory, generic march)
};
hipError_t launch_strip(const StripLaunch& L);
void preload_hot();
void preload_width();
void preload_wide();
void preload_generic();
void preload_strip();
void preload_grid();

// one instance of the persistent form: WV waves per workgroup, WL = generic weights in LDS, 8-bit output / fast_interp chosen at run time
#define NRF_LAUNCH_PERSISTENT_F(G, U, WV, WL, O8, FI)                                                                    \
  do {                                                                                                                   \
    if (L.waves != (WV)) return hipErrorInvalidConfiguration; /* the host sized the workgroup's LDS for another instance */ \
    hipError_t e_ = allow_lds(render_persistent_kernel<G, U, WV, WL, O8, FI>, L.lds);                                    \
    if (e_ != hipSuccess) return e_;                                                                                     \
    hipLaunchKernelGGL((render_persistent_kernel<G, U, WV, WL, O8, FI>), dim3(L.wgs), dim3(64 * WV), L.lds, L.st, *L.M,  \
                       *L.P, *L.VB, (float4*)L.rgba, (float*)L.depth, (unsigned long long*)L.counters, L.queue);         \
  } while (0)
#define NRF_LAUNCH_PERSISTENT_W(G, U, WV, WL)                                                                            \
  do {                                                                                                                   \
    if (L.P->out_mode == OUT_U8) NRF_LAUNCH_PERSISTENT_F(G, U, WV, WL, true, false);                                     \
    else NRF_LAUNCH_PERSISTENT_F(G, U, WV, WL, false, false);                                                            \
  } while (0)
#define NRF_LAUNCH_PERSISTENT(G, U) NRF_LAUNCH_PERSISTENT_W(G, U, persist_waves(G), false)

}  // namespace nrf
