// nrf_kernels.hip -- gfx950 kernels of the render hot path and their launchers.
//
// The stage kernels (the render kernel's device functions one stage at a time, for the parity tests: include/nerfhip.h
// "stage entry points"), the planning kernels, the pixel-format kernels and every launcher.  The render kernel itself is a
// template in nrf_render.h; its instances are compiled in nrf_kernels_{hot,width,wide,generic,strip}.hip.

#include "nrf_render.h"

namespace nrf {

// ------------------------------------------------------------ stage kernels ----
// Hot instance: the same per-level specialisation render_kernel uses (uni_modes of the level's
// group of four), so the bit-exact encode test covers the hot path's index arithmetic.
template <bool FAST>
__global__ __launch_bounds__(256) void encode_grid_kernel(const DevModel M, const float* __restrict__ pos01, uint32_t n,
                                                          uint32_t* __restrict__ out) {
  __shared__ LevelParams lvs[16];
  if (threadIdx.x < 16) lvs[threadIdx.x] = M.lv[threadIdx.x];
  __syncthreads();
  // one thread per (sample, level): out[sample][level] as packed half2
  const uint64_t total = (uint64_t)n * 16u;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t s = (uint32_t)(i >> 4), level = (uint32_t)(i & 15u);
    const float px = pos01[3 * (size_t)s], py = pos01[3 * (size_t)s + 1], pz = pos01[3 * (size_t)s + 2];
    const uint32_t uni = (M.uni_modes >> (2 * (level >> 2))) & 3u;
    if ((M.quad_mask >> level) & 1u) {  // the render kernel gathers this level from its cell-major quad copy: so does this entry point
      uint32_t v[8];
      float fr[3];
      if ((M.quad_far >> (level >> 2)) & 1u) level_gather_quad_far(M.grid, lvs[level], px, py, pz, v, fr);
      else level_gather_quad(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
      out[i] = level_interp<FAST>(v, fr);
    } else if (uni == 2u) out[i] = encode_level<2, FAST>(M.grid, M.grid_bytes, lvs[level], px, py, pz);
    else if (uni == 1u) out[i] = encode_level<1, FAST>(M.grid, M.grid_bytes, lvs[level], px, py, pz);
    else out[i] = encode_level<0, FAST>(M.grid, M.grid_bytes, lvs[level], px, py, pz);
  }
}

// The cell-major quad copy of a level (level_gather_quad): one thread per cell (x, y, z), x, y < res, z <= res, copies the entries
// grid_index (T/.../grid.h:100-117) names for the corners (x | x + 1, y | y + 1, z) -- fast_hash & (size - 1) on a hashed level,
// (x + y res + z res^2) % size on a dense one -- into one 16-byte entry.  Runs once per level at nrf_load_model.
__global__ __launch_bounds__(256) void build_quads_kernel(const uint32_t* __restrict__ table, uint32_t res, uint32_t size, uint32_t hashed,
                                                          uint4* __restrict__ quads) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
  if (x >= res) return;
  uint32_t e[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t cx = x + (k & 1u), cy = y + (k >> 1);
    const uint32_t idx = hashed ? ((cx ^ (cy * 2654435761u) ^ (z * 805459861u)) & (size - 1u)) : ((cx + cy * res + z * res * res) % size);
    e[k] = table[idx];
  }
  quads[((size_t)z * res + y) * res + x] = make_uint4(e[0], e[1], e[2], e[3]);
}

// Generic instance: one thread per (sample, level) writes F halves of the row [feat_w]; level 0's thread also
// writes the zero padding (grid.h:959-969).
__global__ __launch_bounds__(256) void gen_encode_grid_kernel(const DevModel M, const float* __restrict__ pos01, uint32_t n,
                                                              half_t* __restrict__ out) {
  __shared__ LevelParams lvs[16];
  if (threadIdx.x < 16) lvs[threadIdx.x] = M.lv[threadIdx.x];
  __syncthreads();
  const GenModel& G = *M.gen;
  const half_t* __restrict__ grid = reinterpret_cast<const half_t*>(M.grid);
  const uint64_t total = (uint64_t)n * G.n_levels;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t s = (uint32_t)(i / G.n_levels), level = (uint32_t)(i - (uint64_t)s * G.n_levels);
    const float px = pos01[3 * (size_t)s], py = pos01[3 * (size_t)s + 1], pz = pos01[3 * (size_t)s + 2];
    half_t* row = out + (size_t)s * G.feat_w;
    half_t r[8];
    if (M.hot_grid != 0u) {
      // the model's frames come from a GRID instance of the render kernel (nrf_render.h grid_features): this entry point then runs
      // THAT instance's gathers and interpolation, so that the bit-exact encode test covers them (tests/test_generic_gpu.py)
      // -- including the specialisation grid_features picks for the level: the render kernel's lane group g holds the levels
      // 4 * jl + g, and step jl takes the dense (1) / hashed (2) form of the index arithmetic when all of its levels are
      float fr[3];
      uint32_t o[4] = {0u, 0u, 0u, 0u};
      const uint32_t uni = (M.uni_modes >> (2 * (level >> 2))) & 3u;
      if (M.hot_grid == 1u) {  // F = 1: one half per (sample, level)
        uint32_t r;
        if (M.grid_nearest) {
          r = uni == 2u ? level_nearest_f1<2>(M.grid, M.grid_bytes, lvs[level], px, py, pz) : level_nearest_f1<0>(M.grid, M.grid_bytes, lvs[level], px, py, pz);
        } else {
          uint32_t v[8];
          if (uni == 2u) level_gather_f1<2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
          else level_gather_f1<0>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
          if (M.grid_smooth) smoothstep_fractions(fr);
          r = level_interp<false>(v, fr);
        }
        reinterpret_cast<unsigned short*>(row)[level] = (unsigned short)(r & 0xffffu);
        if (level == 0)
          for (uint32_t j = G.feat_raw; j < G.feat_w; ++j) row[j] = (half_t)0.0f;
        continue;
      }
      if (M.grid_nearest) {
        if (M.hot_grid == 2u) {
          uint32_t q[1];
          if (uni == 2u) level_nearest<2, 1>(M.grid, M.grid_bytes, lvs[level], px, py, pz, q);
          else level_nearest<0, 1>(M.grid, M.grid_bytes, lvs[level], px, py, pz, q);
          o[0] = q[0];
        } else if (M.hot_grid == 4u) {
          uint32_t q[2];
          if (uni == 2u) level_nearest<2, 2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, q);
          else level_nearest<0, 2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, q);
          o[0] = q[0]; o[1] = q[1];
        } else {
          if (uni == 2u) level_nearest<2, 4>(M.grid, M.grid_bytes, lvs[level], px, py, pz, o);
          else level_nearest<0, 4>(M.grid, M.grid_bytes, lvs[level], px, py, pz, o);
        }
      } else if (M.hot_grid == 2u) {
        uint32_t v[8];
        if (uni == 2u) level_gather<2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        else level_gather<0>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        if (M.grid_smooth) smoothstep_fractions(fr);
        o[0] = level_interp<false>(v, fr);
      } else if (M.hot_grid == 4u) {
        uint32_t v[16], q[2];
        if (uni == 2u) level_gather_wide<2, 2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        else if (uni == 1u) level_gather_wide<1, 2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        else level_gather_wide<0, 2>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        if (M.grid_smooth) smoothstep_fractions(fr);
        level_interp_wide<2>(v, fr, q);
        o[0] = q[0]; o[1] = q[1];
      } else {
        uint32_t v[32];
        if (uni == 2u) level_gather_wide<2, 4>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        else if (uni == 1u) level_gather_wide<1, 4>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        else level_gather_wide<0, 4>(M.grid, M.grid_bytes, lvs[level], px, py, pz, v, fr);
        if (M.grid_smooth) smoothstep_fractions(fr);
        level_interp_wide<4>(v, fr, o);
      }
      for (uint32_t f = 0; f < G.F; f += 2u) *reinterpret_cast<uint32_t*>(row + level * G.F + f) = o[f >> 1];
      if (level == 0)
        for (uint32_t j = G.feat_raw; j < G.feat_w; ++j) row[j] = (half_t)0.0f;
      continue;
    }
    switch (G.F) {
      case 1: { half_t q[1]; gen_level<1>(grid, lvs[level], G.interp, px, py, pz, q); r[0] = q[0]; } break;
      case 2: { half_t q[2]; gen_level<2>(grid, lvs[level], G.interp, px, py, pz, q); r[0] = q[0]; r[1] = q[1]; } break;
      case 4: { half_t q[4]; gen_level<4>(grid, lvs[level], G.interp, px, py, pz, q);
#pragma unroll
                for (int f = 0; f < 4; ++f) r[f] = q[f]; } break;
      default: gen_level<8>(grid, lvs[level], G.interp, px, py, pz, r); break;
    }
    for (uint32_t f = 0; f < G.F; ++f) row[level * G.F + f] = r[f];
    if (level == 0)
      for (uint32_t j = G.feat_raw; j < G.feat_w; ++j) row[j] = (half_t)0.0f;
  }
}

__global__ __launch_bounds__(256) void encode_dir_kernel(const DevModel M, const float* __restrict__ dir01, uint32_t n,
                                                         uint32_t* __restrict__ out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    half_t e[16];
    encode_dir16(M, dir01[3 * (size_t)i], dir01[3 * (size_t)i + 1], dir01[3 * (size_t)i + 2], e);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      half2_t h;
      h.x = e[2 * j];
      h.y = e[2 * j + 1];
      out[(size_t)i * 8 + j] = h2_bits(h);
    }
  }
}

__global__ __launch_bounds__(256) void gen_encode_dir_kernel(const DevModel M, const float* __restrict__ dir01, uint32_t n,
                                                             half_t* __restrict__ out) {
  const GenModel& G = *M.gen;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    gen_encode_dir(M, G, dir01[3 * (size_t)i], dir01[3 * (size_t)i + 1], dir01[3 * (size_t)i + 2], out + (size_t)i * G.dir_w);
}

// Both MLPs on pre-encoded inputs: one wave = 64 samples per trip.
// feat fp16 [n][32], dirfeat fp16 [n][16] -> out fp16 [n][4] = (r, g, b, sigma)
// HBM-bound (104 B and 20 480 FLOP per sample): every lane reads 16 B of its sample's feature row and 8 B
// of its direction row (the natural-K-order copy of the first weight matrix makes that the B fragment),
// a wave reads 1 KiB + 512 B contiguous per 16-sample tile, and the next chunk's rows are in flight
// while the current one goes through the 80 MFMAs.
constexpr int MLP_TILES = 2;  // 16-sample tiles per trip
struct RegFrags {  // the lane's weight fragments, held in registers for the kernel's life (80 VGPRs)
  const half8_t* w;
  __device__ __forceinline__ half8_t operator()(int f) const { return w[f]; }
};
// OCC: workgroups of 4 waves per compute unit the registers are budgeted for (3: 168 VGPRs, 2: 256); NAMED: the three row
// buffers rotate by name (loop unrolled three times) instead of by copying registers -- see below.  The shipped form is
// chosen in launch_mlp_forward (NRF_MLP_FORM for A/B runs).
constexpr int MLP_FORM_DEFAULT = 30;
template <bool REPEAT, int OCC = 3, bool NAMED = false>
__global__ __launch_bounds__(256, OCC) void mlp_forward_kernel(const DevModel M, const uint4* __restrict__ feat,
                                                          const uint2* __restrict__ dirfeat, uint32_t n,
                                                          half_t* __restrict__ out, uint32_t repeat) {
  constexpr int T = MLP_TILES, CH = 16 * T;
  const int lane = lane_id(), g = lane >> 4, c = lane & 15;
  half8_t wreg[N_FRAGS_ALL];
#pragma unroll
  for (int f = 0; f < N_FRAGS_ALL; ++f) wreg[f] = __builtin_bit_cast(half8_t, M.wfrag[f * 64 + lane]);
  const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n_chunks = (n + CH - 1) / CH;
  // Three row buffers in rotation: while chunk i is evaluated out of one, chunk i + 1 waits in the second and the loads of chunk
  // i + 2 fly into the third.  The rotation is by NAME (the loop is unrolled three times, each copy with its own roles), not by
  // copying registers: a copy `next = next2` at the loop's end makes the wave wait for loads it has only just issued
  // (s_waitcnt vmcnt(0) at every latch: round 4's form -- the prefetch reached one chunk ahead, not two).
  uint4 fb[3][T];
  uint2 db[3][T];
  if (n == 0) return;
  // rows past the end (the last chunk's padding, the prefetches beyond the last chunk) read row n - 1 instead: no
  // predication, no zero fill (85 v_mov per trip before) -- their results are never stored
  auto load_chunk = [&](uint32_t chunk, uint4 (&f)[T], uint2 (&d)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint64_t s64 = (uint64_t)chunk * CH + 16u * t + c;
      const uint32_t s = s64 < n ? (uint32_t)s64 : n - 1u;
      f[t] = feat[(size_t)s * 4 + g];     // halves 8g..8g+7 of the row
      d[t] = dirfeat[(size_t)s * 4 + g];  // entries 4g..4g+3
    }
  };
  // one chunk: its rows are in (fv, dv); the loads of the chunk two steps ahead go to (fl, dl)
  auto step = [&](uint32_t chunk, uint4 (&fv)[T], uint2 (&dv)[T], uint4 (&fl)[T], uint2 (&dl)[T]) {
    load_chunk(chunk + 2 * n_waves, fl, dl);  // in flight during the MFMAs of this chunk and the next
    half8_t f[T];
    half4_t df[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f[t] = __builtin_bit_cast(half8_t, fv[t]);
      df[t] = __builtin_bit_cast(half4_t, dv[t]);
    }
    MlpOut<T> o;
    // REPEAT (nrf_mlp_forward_repeat): the same rows `repeat` times from registers -- the rate of the MFMA chain
    // with its re-packing, without the HBM stream; the empty asm keeps the evaluations from being merged
    for (uint32_t r = 0; r < (REPEAT ? repeat : 1u); ++r) {
      if (REPEAT) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          asm volatile("" : "+v"(fv[t].x), "+v"(fv[t].y), "+v"(fv[t].z), "+v"(fv[t].w), "+v"(dv[t].x), "+v"(dv[t].y));
          f[t] = __builtin_bit_cast(half8_t, fv[t]);
          df[t] = __builtin_bit_cast(half4_t, dv[t]);
        }
      }
      mlp_tiles<T, FRAG_D0_NATURAL>(RegFrags{wreg}, f, df, o, M.rgb_output_activation == NRF_ACT_SIGMOID);
    }
    // One 8-byte store per sample: (r, g, b) of every tile are in lane row 0, sigma of tile g in lane row g -- move tile g's
    // colours to lane row g as well (v_permlane16_swap) and let the first T lane rows write 128 contiguous bytes each
    // (three partial stores of 4 + 2 + 2 bytes per sample before).
    static_assert(T == 1 || T == 2, "lane rows that hold a tile's outputs");
    uint32_t rg_row = o.rg[0], bx_row = o.bx[0];
    if constexpr (T == 2) {
      rg_row = __builtin_amdgcn_permlane16_swap(o.rg[0], o.rg[1], false, false)[0];
      bx_row = __builtin_amdgcn_permlane16_swap(o.bx[0], o.bx[1], false, false)[0];
    }
    if (g < T) {
      const uint32_t s = chunk * CH + 16u * g + c;
      const uint32_t sigma_bits = (uint32_t)__builtin_bit_cast(unsigned short, o.sigma);
      if (s < n) *reinterpret_cast<uint2*>(out + 4 * (size_t)s) = make_uint2(rg_row, (bx_row & 0xffffu) | (sigma_bits << 16));
    }
  };
  load_chunk(wave_global, fb[0], db[0]);
  load_chunk(wave_global + n_waves, fb[1], db[1]);
  if constexpr (!NAMED) {  // rotation by copies (rounds 1-4)
    for (uint32_t chunk = wave_global; chunk < n_chunks; chunk += n_waves) {
      step(chunk, fb[0], db[0], fb[2], db[2]);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        fb[0][t] = fb[1][t];
        db[0][t] = db[1][t];
        fb[1][t] = fb[2][t];
        db[1][t] = db[2][t];
      }
    }
    return;
  }
  for (uint32_t chunk = wave_global; chunk < n_chunks;) {
    step(chunk, fb[0], db[0], fb[2], db[2]);
    chunk += n_waves;
    if (chunk >= n_chunks) break;
    step(chunk, fb[1], db[1], fb[0], db[0]);
    chunk += n_waves;
    if (chunk >= n_chunks) break;
    step(chunk, fb[2], db[2], fb[1], db[1]);
    chunk += n_waves;
  }
}

// Generic instance of the stage above: feat fp16 [n][feat_w], dirfeat fp16 [n][dir_w] -> out fp16 [n][4].
// One wave = 32 samples per trip: rows copied into the wave's LDS regions, then the same gen_mlps as render_kernel.
__global__ __launch_bounds__(256) void gen_mlp_forward_kernel(const DevModel M, const half_t* __restrict__ feat,
                                                              const half_t* __restrict__ dirfeat, uint32_t n,
                                                              uint2* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const GenModel& G = *M.gen;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id(), g = lane >> 4, c = lane & 15;
  const LdsMap lm = lds_map<NET_GENERIC>(smem, M, wave, 4);
  const uint32_t wave_global = blockIdx.x * 4 + wave, n_waves = gridDim.x * 4;
  const uint32_t n_chunks = (n + GEN_SAMPLES - 1) / GEN_SAMPLES;
  for (uint32_t chunk = wave_global; chunk < n_chunks; chunk += n_waves) {
    const uint32_t first = chunk * GEN_SAMPLES;
    for (uint32_t r = 0; r < (uint32_t)GEN_SAMPLES; ++r) {  // rows of the chunk, 64 lanes across the columns
      const uint32_t s = first + r;
      half_t* xr = lm.gen.X + (size_t)r * G.act_stride;
      half_t* dr = lm.gen.dir + (size_t)r * G.dir_stride;
      for (uint32_t j = lane; j < G.feat_k; j += 64u) xr[j] = (s < n && j < G.feat_w) ? feat[(size_t)s * G.feat_w + j] : (half_t)0.0f;
      for (uint32_t j = lane; j < G.dir_w; j += 64u) dr[j] = s < n ? dirfeat[(size_t)s * G.dir_w + j] : (half_t)0.0f;
    }
    gen_wave_sync();
    int ray[GEN_TILES];
#pragma unroll
    for (int t = 0; t < GEN_TILES; ++t) ray[t] = 16 * t + c;
    float4_t o[GEN_TILES];
    gen_mlps<false>(M, G, lm.gen, lane, ray, o);
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < GEN_TILES; ++t) {
        const uint32_t s = first + 16u * t + c;
        if (s < n) out[s] = make_uint2(pack_h2(o[t][0], o[t][1]), pack_h2(o[t][2], o[t][3]));
      }
    }
    gen_wave_sync();
  }
}

// Whole network on raw march output through the SAME code path as render_kernel.
// DENSITY_ONLY (generic or hot): sigma only, rgb untouched (density-grid generation evaluates positions without directions).
template <int NET>
__global__ __launch_bounds__(256, 2) void network_kernel(const DevModel M, const float* __restrict__ xyz,
                                                      const float* __restrict__ dir, uint32_t n, float* __restrict__ sigma,
                                                      float* __restrict__ rgb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool GEN = NET == NET_GENERIC;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id();
  const LdsMap lm = lds_map<NET>(smem, M, wave, 4);
  uint4* wl = lm.wl;
  LevelParams* lvs = lm.lvs;
  if constexpr (!GEN) stage_fragments<NET>(M, wl);
  if (threadIdx.x < 16) lvs[threadIdx.x] = M.lv[threadIdx.x];
  __syncthreads();
  WaveLds* W = lm.W;
  const uint32_t wave_global = blockIdx.x * 4 + wave;
  const uint32_t n_waves = gridDim.x * 4;
  const uint32_t n_chunks = (n + 63u) >> 6;
  for (uint32_t chunk = wave_global; chunk < n_chunks; chunk += n_waves) {
    const uint32_t i = (chunk << 6) + lane;
    const int S = (int)min(64u, n - (chunk << 6));
    if (i < n) {
      W->pos[lane] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
                                 __builtin_bit_cast(float, lane));
      float u0 = 0.5f * dir[3 * (size_t)i]; u0 = u0 + 0.5f;
      float u1 = 0.5f * dir[3 * (size_t)i + 1]; u1 = u1 + 0.5f;
      float u2 = 0.5f * dir[3 * (size_t)i + 2]; u2 = u2 + 0.5f;
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
        if constexpr (NET == NET_WIDE) {
          lm.gen.rayd[3 * lane] = u0;
          lm.gen.rayd[3 * lane + 1] = u1;
          lm.gen.rayd[3 * lane + 2] = u2;
        }
      }
    }
    wave_sync();
    network_dispatch<NET>(M, wl, lvs, W, lm.gen, S, lane, 1.0f);
    wave_sync();
    if (i < n) {
      const float4 so = W->out[lane];
      sigma[i] = so.w;
      rgb[3 * (size_t)i] = so.x;
      rgb[3 * (size_t)i + 1] = so.y;
      rgb[3 * (size_t)i + 2] = so.z;
    }
    wave_sync();
  }
}

__global__ __launch_bounds__(256) void generate_rays_kernel(const DevModel M, const FrameParams P, float* __restrict__ rays_o,
                                                            float* __restrict__ rays_d, float* __restrict__ nears,
                                                            float* __restrict__ fars) {
  const int n = P.W * P.H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int px = i % P.W, py = i / P.W;
    const float o[3] = {P.org[0], P.org[1], P.org[2]};
    float d[3], nr, fr;
    ray_dir(P.R, P.cam, px, py, d);
    near_far(M.aabb, o, d, P.min_near, nr, fr);
    if (rays_o) { rays_o[3 * (size_t)i] = o[0]; rays_o[3 * (size_t)i + 1] = o[1]; rays_o[3 * (size_t)i + 2] = o[2]; }
    if (rays_d) { rays_d[3 * (size_t)i] = d[0]; rays_d[3 * (size_t)i + 1] = d[1]; rays_d[3 * (size_t)i + 2] = d[2]; }
    if (nears) nears[i] = nr;
    if (fars) fars[i] = fr;
  }
}

template <bool COARSE>
__global__ __launch_bounds__(256) void march_kernel(const DevModel M, float dt_gamma, const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d, const float* __restrict__ rays_t,
                                                    const float* __restrict__ fars, uint32_t n, uint32_t n_step,
                                                    float* __restrict__ xyzs, float* __restrict__ dirs,
                                                    float* __restrict__ deltas, uint32_t perturb) {
  const MarchConst mc = march_const(M, dt_gamma);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float ox = rays_o[3 * (size_t)i], oy = rays_o[3 * (size_t)i + 1], oz = rays_o[3 * (size_t)i + 2];
    const float dx = rays_d[3 * (size_t)i], dy = rays_d[3 * (size_t)i + 1], dz = rays_d[3 * (size_t)i + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const int sx = __builtin_signbitf(dx) ? 0 : 1, sy = __builtin_signbitf(dy) ? 0 : 1, sz = __builtin_signbitf(dz) ? 0 : 1;
    const float far = fars[i];
    float t = rays_t[i];  // kernel_march_rays with an explicit n_step: render_utils.h:591-653
    if (perturb) t = t + mc.dt_min * pcg32_first_float((uint64_t)i, (uint64_t)perturb);  // :585-589, n = the ray's place in the call
    float last_t = t;
    bool marching = true;
    for (uint32_t k = 0; k < n_step; ++k) {
      const size_t s = (size_t)i * n_step + k;
      float x = 0.f, y = 0.f, z = 0.f, dt = 0.f;
      bool found = false;
      if (marching) {
        int budget = 0x7fffffff;
        found = march_next<COARSE, MARCH_GENERIC>(mc, M.occ_bits, M.occ_coarse, M.cell_bound, ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, sx, sy, sz,
                                   far, -3.402823466e+38f, budget, t, x, y, z, dt) == MARCH_FOUND;
      }
      marching = found;
      if (found) t += dt;
      // unused slots are zero-filled (deviation D-1)
      xyzs[3 * s] = found ? x : 0.f;
      xyzs[3 * s + 1] = found ? y : 0.f;
      xyzs[3 * s + 2] = found ? z : 0.f;
      dirs[3 * s] = found ? dx : 0.f;
      dirs[3 * s + 1] = found ? dy : 0.f;
      dirs[3 * s + 2] = found ? dz : 0.f;
      deltas[2 * s] = found ? dt : 0.f;
      deltas[2 * s + 1] = found ? t - last_t : 0.f;
      if (found) last_t = t;
    }
  }
}

__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                        const float* __restrict__ deltas, uint32_t n, uint32_t n_step,
                                                        float* __restrict__ rays_t, float* __restrict__ state) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float* st = state + 5 * (size_t)i;
    float ws = st[0], dep = st[1], cr = st[2], cg = st[3], cb = st[4];
    float t = rays_t[i];
    uint32_t step = 0;
    while (step < n_step) {
      const size_t s = (size_t)i * n_step + step;
      const float dt = deltas[2 * s];
      if (dt == 0) break;
      const float alpha = 1.0f - __expf(-sigmas[s] * dt);
      const float T = 1 - ws;
      const float wgt = alpha * T;
      ws += wgt;
      t += deltas[2 * s + 1];
      dep += wgt * t;
      cr += wgt * rgbs[3 * s];
      cg += wgt * rgbs[3 * s + 1];
      cb += wgt * rgbs[3 * s + 2];
      if (T <= 9.99999974737875e-05f) break;
      step++;
    }
    rays_t[i] = step < n_step ? -1.0f : t;
    st[0] = ws; st[1] = dep; st[2] = cr; st[3] = cg; st[4] = cb;
  }
}

// ---- density grid from the network (NerfRender::generate_density_grid, R/src/nerf_render.cu:388-429) ----
// Cell positions of one cascade: init_xyzs (render_utils.h:91-108: -1.f + 2.f/(H-1)*id per axis, x-major cell order)
// scaled by dd_scale's k = bound_c - bound_c/H (nerf_render.cu:410-413); a constant direction (the density does not
// depend on it).
__global__ __launch_bounds__(256) void density_positions_kernel(uint32_t H, float k, float* __restrict__ xyz, float* __restrict__ dir) {
  const uint32_t n = H * H * H;
  const float step = 2.f / (float)(H - 1);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t id[3] = {i / (H * H), (i % (H * H)) / H, i % H};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = step * (float)id[a];
      v = -1.f + v;
      xyz[3 * (size_t)i + a] = k * v;
      dir[3 * (size_t)i + a] = a == 2 ? 1.0f : 0.0f;
    }
  }
}

// dd_scale (k = 0.001691) + dg_update (render_utils.h:120-128), n_iterations times from the reference's initial value
// 1/64 (nerf_render.cu:393): g = max(g * decay, k * sigma) for g >= 0.
__global__ __launch_bounds__(256) void density_update_kernel(const float* __restrict__ sigma, uint32_t n, float decay, int n_iterations,
                                                             float* __restrict__ grid) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float tmp = 0.001691f * sigma[i];
    float g = 1.0f / 64;
    for (int it = 0; it < n_iterations; ++it) {
      if (g >= 0) {
        const float gd = g * decay;
        g = gd > tmp ? gd : tmp;
      }
    }
    grid[i] = g;
  }
}

// gathered [shard][view][tiles_per_shard][64][C] -> row-major [view][H][W][C]
__global__ __launch_bounds__(256) void untile_kernel(const float* __restrict__ gathered, int shard_count, int tiles_per_shard,
                                                     int C, int W, int H, int tiles_x, int n_views, float* __restrict__ out) {
  const size_t frame = (size_t)W * H, total = frame * (size_t)n_views;
  const int strips_x = (tiles_x + 3) >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int view = (int)(i / frame);
    const size_t p = i - (size_t)view * frame;
    const int px = (int)(p % W), py = (int)(p / W);
    const int tx = px >> 3, ty = py >> 3;
    const int strip = ty * strips_x + (tx >> 2);
    const int shard = strip % shard_count, k = (strip / shard_count) * 4 + (tx & 3);
    const int l = (py & 7) * 8 + (px & 7);
    const float* src = gathered + ((((size_t)shard * n_views + view) * tiles_per_shard + k) * 64 + l) * C;
    for (int ch = 0; ch < C; ++ch) out[i * C + ch] = src[ch];
  }
}

__global__ __launch_bounds__(256) void quantize_kernel(const float4* __restrict__ rgba, const float* __restrict__ depth, int n,
                                                       unsigned char* __restrict__ rgb8, unsigned char* __restrict__ depth8) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 v = rgba[i];
    rgb8[3 * (size_t)i] = quant_u8(v.x);
    rgb8[3 * (size_t)i + 1] = quant_u8(v.y);
    rgb8[3 * (size_t)i + 2] = quant_u8(v.z);
    depth8[i] = quant_u8(depth[i]);
  }
}

// ------------------------------------------------------------ queue planning ----
// One view (a render_frame call) is ~8 strips per wave of the persistent kernel, and a heavy 8x8 tile alone is longer than a
// wave's fair share of the frame: in which ORDER the queues hand the strips out decides how long the launch's tail is.
// Two small kernels ahead of the render price and sort the launch's strips:
//   plan_price_kernel  one thread per tile: its centre ray walks the dilated coarse occupancy (a DDA, as the render kernel's visibility walk) between its entry into and its
//                      exit from the box of occupied cells, against the dilated coarse occupancy table (LDS); the points in
//                      set cells, times their spacing over the march's step there, estimate the ray's march steps.  A strip
//                      takes the maximum of its four tiles.  (It also zeroes the call's statistics counters and queue words:
//                      the memset that used to precede the render.)
//   plan_sort_kernel   one workgroup: counting sort of every class's queue positions by that price, dearest first
//                      (PLAN_BINS bins relative to the launch's dearest strip; a bin roughly keeps the centre-out order).
// The price is GEOMETRIC on purpose: with the MEASURED cost of every tile as its price (a diagnostic experiment, round 4) a lone
// 1080p view took 0.98 ms instead of 0.87 (no plan: 0.90) -- tiles of equal cost lie all over the picture, a cost-sorted
// queue renders distant strips side by side and the table's cache locality is gone; the path length through the occupancy
// varies smoothly over the picture, so sorting by it is roughly dearest-first AND spatially coherent.
// The render kernel then reads position i of a class through this permutation.  Only the ORDER of the work depends on the
// estimate: every strip is rendered exactly as before, and the frames are bit-identical with and without a plan
// (tests/test_persistent_gpu.py).
//   plan buffer (unsigned words): [1] max price (float bits; reset by the sort), [4 .. 4 + cap) price per position
//   (class-major: class c's positions follow those of the classes before it), [4 + cap .. 4 + 2 cap) the order
constexpr int PLAN_BINS = 64, PLAN_THREADS = 256, PLAN_SORT_THREADS = 1024;
struct PlanClasses {
  unsigned off[9];  // where each class's positions begin (class-major numbering); [8] = all positions
  unsigned n_cls;
  __device__ void init(const ViewBatch& VB) {
    n_cls = (unsigned)VB.n_classes;
    const unsigned cls_cols = (unsigned)VB.class_cols, n_units = (unsigned)VB.q_total;
    unsigned o = 0u;
    for (unsigned c = 0; c < 8u; ++c) {
      off[c] = o;
      if (c < n_cls) o += n_units * ((cls_cols - c + n_cls - 1u) / n_cls);
    }
    off[8] = o;
  }
  __device__ __forceinline__ unsigned class_of(unsigned g, unsigned& begin) const {
    unsigned cls = 0;
#pragma unroll
    for (unsigned c = 1; c < 8u; ++c) cls += (c < n_cls && g >= off[c]) ? 1u : 0u;  // (an empty class shares its successor's offset)
    begin = off[cls];
    return cls;
  }
};

__global__ __launch_bounds__(PLAN_THREADS) void plan_price_kernel(const DevModel M, const FrameParams P, const ViewBatch VB,
                                                                  unsigned* __restrict__ plan, unsigned* __restrict__ zero, unsigned zero_words) {
  __shared__ PlanClasses pc;
  __shared__ float s_max[PLAN_THREADS / 64];
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dil[];  // the dilated table, every cascade
  const unsigned T = blockIdx.x * PLAN_THREADS + threadIdx.x, n_threads = gridDim.x * PLAN_THREADS;
  for (unsigned i = T; i < zero_words; i += n_threads) zero[i] = 0u;
  const unsigned dil_words = M.dilated_level_words * M.cascade;
  for (unsigned i = threadIdx.x; i < dil_words; i += PLAN_THREADS) s_dil[i] = M.occ_dilated[i];
  if (threadIdx.x == 0) pc.init(VB);
  __syncthreads();
  const unsigned n_cls = pc.n_cls, cls_cols = (unsigned)VB.class_cols, n_pos = pc.off[8];
  float* est = reinterpret_cast<float*>(plan + 4);
  const unsigned g = T >> 2, bt = T & 3u;
  float cost = 0.0f;
  if (g < n_pos) {
    unsigned begin;
    const unsigned cls = pc.class_of(g, begin);
    const unsigned pos = g - begin, ncols = (cls_cols - cls + n_cls - 1u) / n_cls;
    const int u = (int)(pos / ncols), j = (int)(pos - (unsigned)u * ncols);
    int lo = 0, hi = VB.n_views - 1;  // the last view whose first unit is <= u (render_persistent_kernel's ballot count)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (VB.v[mid].q_begin <= u) lo = mid; else hi = mid - 1;
    }
    const ViewParams& V = VB.v[lo];
    // position -> tile, as render_persistent_kernel maps it
    const int ul = u - V.q_begin;
    const int off2 = (ul + 1) >> 1;
    const int row = V.q_row0 + (P.centre_out != 0 ? (V.q_rows - 1) / 2 + ((ul & 1) ? off2 : -off2) : ul);
    const int sxn = (P.tiles_x + 3) >> 2, N = P.shard_count;
    const int s_row = row * sxn;
    const int ls_first = s_row > P.shard_index ? (s_row - P.shard_index + N - 1) / N : 0;
    const int ls = ls_first + (int)cls + (int)n_cls * j;
    const int k_local = ls * 4 + (int)bt;
    bool valid = ul < V.q_rows && ls * N + P.shard_index < s_row + sxn && k_local < V.k_hi && k_local < P.n_local_tiles;
    const int strip = (k_local >> 2) * N + P.shard_index;
    const int tx = (strip % sxn) * 4 + (k_local & 3), ty = strip / sxn;
    valid = valid && !(tx * 8 > V.roi[2] || tx * 8 + 7 < V.roi[0] || ty * 8 > V.roi[3] || ty * 8 + 7 < V.roi[1]) && tx < P.tiles_x;
    if (valid) {
      const int px = min(tx * 8 + 4, P.W - 1), py = min(ty * 8 + 4, P.H - 1);
      const float o[3] = {V.org[0], V.org[1], V.org[2]};
      float d[3], near, far, t_in, t_out;
      ray_dir(V.R, V.cam, px, py, d);
      near_far(M.aabb, o, d, P.min_near, near, far);
      box_interval(M.occ_box, o, 1 / d[0], 1 / d[1], 1 / d[2], t_in, t_out);
      const float t0 = fmaxf(t_in, near), t1 = fminf(far, t_out);
      if (near < far && M.occ_box[0] <= M.occ_box[3] && t0 < t1) {
        // a DDA over the dilated coarse cells (the render kernel's own visibility walk, one per cascade on that cascade's stretch
        // of the ray): the path inside set cells over the march's step there.  (Round 3 sampled 32 points along the ray: on a
        // 2-3 unit stretch those are 0.06-0.1 apart, a coarse cell is 0.06 wide -- thin structures were priced at zero, and
        // tiles as dear as 0.6 Mcycles started in the last third of a lone frame: profiles/r04/tile_cost_map_before.txt)
        const int Hc = (int)(M.H >> 2);
        const float dt_min = 2 * 1.7320508075688772f / 1024, dt_max = 2 * M.bound / (float)M.H;
        const float rd[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
        for (uint32_t k = 0; k < M.cascade; ++k) {
          const float mb = M.cascade > 1 ? fminf(ldexpf(1.0f, (int)k), M.bound) : fminf(1.0f, M.bound);
          float c_in = t0, c_out = t1;
          if (M.cascade > 1) {
            const float cube[6] = {-mb, -mb, -mb, mb, mb, mb};
            float a, b;
            box_interval(cube, o, rd[0], rd[1], rd[2], a, b);
            if (a == a && b == b) {
              c_in = fmaxf(c_in, a);
              c_out = fminf(c_out, b);
            }
          }
          if (!(c_in < c_out)) continue;
          const uint32_t* dil = s_dil + (size_t)k * M.dilated_level_words;
          const float cs = 2.0f * mb / (float)Hc, rcs = (float)Hc / (2.0f * mb);
          int ci[3], step[3];
          float tmax[3], tdelta[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float pa = o[a] + c_in * d[a];
            int cc = (int)floorf((pa + mb) * rcs);
            cc = cc < 0 ? 0 : (cc > Hc - 1 ? Hc - 1 : cc);
            ci[a] = cc;
            step[a] = d[a] >= 0.0f ? 1 : -1;
            const float edge = (float)(cc + (d[a] >= 0.0f ? 1 : 0)) * cs - mb;
            const bool flat = !(fabsf(rd[a]) <= 3.0e38f);
            tmax[a] = flat ? 3.0e38f : (edge - o[a]) * rd[a];
            tdelta[a] = flat ? 3.0e38f : cs * fabsf(rd[a]);
          }
          float t = c_in;
          for (int guard = 0; guard < 3 * Hc + 3; ++guard) {
            const uint32_t cc = ((uint32_t)ci[0] * Hc + (uint32_t)ci[1]) * Hc + (uint32_t)ci[2];
            const float t_exit = fminf(fminf(tmax[0], fminf(tmax[1], tmax[2])), c_out);
            if (((dil[cc >> 5] >> (cc & 31u)) & 1u) && t_exit > t) cost += (t_exit - t) / clampf(0.5f * (t + t_exit) * P.dt_gamma, dt_min, dt_max);
            if (!(t_exit < c_out)) break;
            t = t_exit;
            if (tmax[0] <= t_exit) { ci[0] += step[0]; tmax[0] += tdelta[0]; }
            if (tmax[1] <= t_exit) { ci[1] += step[1]; tmax[1] += tdelta[1]; }
            if (tmax[2] <= t_exit) { ci[2] += step[2]; tmax[2] += tdelta[2]; }
            if ((unsigned)ci[0] >= (unsigned)Hc || (unsigned)ci[1] >= (unsigned)Hc || (unsigned)ci[2] >= (unsigned)Hc) break;
          }
        }
      }
    }
    if (!(cost >= 0.0f) || cost > 1.0e6f) cost = cost > 1.0e6f ? 1.0e6f : 0.0f;  // (NaN: a degenerate camera -- any order is a valid order)
  }
  cost = fmaxf(cost, __shfl_xor(cost, 1));
  cost = fmaxf(cost, __shfl_xor(cost, 2));  // the strip's price: its dearest tile
  if (bt == 0u && g < n_pos) est[g] = cost;
  float wmax = cost;
  for (int sh = 4; sh < 64; sh <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, sh));
  if ((threadIdx.x & 63u) == 0u) s_max[threadIdx.x >> 6] = wmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = s_max[0];
    for (int w = 1; w < PLAN_THREADS / 64; ++w) m = fmaxf(m, s_max[w]);
    if (m > 0.0f) atomicMax(plan + 1, __float_as_uint(m));  // (prices are >= 0: their bit patterns order like the values)
  }
}

__global__ __launch_bounds__(PLAN_SORT_THREADS) void plan_sort_kernel(const ViewBatch VB, unsigned* __restrict__ plan, unsigned cap) {
  __shared__ PlanClasses pc;
  __shared__ unsigned s_bins[8 * PLAN_BINS];
  extern __shared__ __attribute__((aligned(16))) uint8_t s_bin[];  // a bin per position
  if (threadIdx.x == 0) pc.init(VB);
  for (unsigned i = threadIdx.x; i < 8u * PLAN_BINS; i += PLAN_SORT_THREADS) s_bins[i] = 0u;
  __syncthreads();
  const unsigned n_pos = pc.off[8];
  const float* est = reinterpret_cast<const float*>(plan + 4);
  unsigned* order = plan + 4 + cap;
  const float top = __uint_as_float(plan[1]);
  const float scale = top > 0.0f ? (float)PLAN_BINS / top : 0.0f;
  const float4* est4 = reinterpret_cast<const float4*>(est);
  for (unsigned q4 = threadIdx.x; q4 * 4u < n_pos; q4 += PLAN_SORT_THREADS) {
    const float4 v = est4[q4];
    const float c4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned i = q4 * 4u + (unsigned)e;
      if (i < n_pos) {
        unsigned begin;
        const unsigned cls = pc.class_of(i, begin);
        const int q = (int)(c4[e] * scale);
        const unsigned b = (unsigned)(PLAN_BINS - 1 - (q < 0 ? 0 : (q > PLAN_BINS - 1 ? PLAN_BINS - 1 : q)));  // dearest: bin 0
        s_bin[i] = (uint8_t)b;
        atomicAdd(&s_bins[cls * PLAN_BINS + b], 1u);
      }
    }
  }
  __syncthreads();
  static_assert(PLAN_BINS == 64 && PLAN_SORT_THREADS >= 8 * 64, "one wave scans one class's bins");
  if (threadIdx.x < 8u * PLAN_BINS) {  // exclusive prefix inside each class: wave w = class w, lane = bin
    const unsigned n = s_bins[threadIdx.x];
    unsigned run = n;
    for (int sh = 1; sh < 64; sh <<= 1) {
      const unsigned up = __shfl_up(run, sh);
      if ((int)(threadIdx.x & 63u) >= sh) run += up;
    }
    s_bins[threadIdx.x] = run - n;
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < n_pos; i += PLAN_SORT_THREADS) {  // a thread's positions in queue order: a bin roughly keeps its strips' order
    unsigned begin;
    const unsigned cls = pc.class_of(i, begin);
    const unsigned slot = atomicAdd(&s_bins[cls * PLAN_BINS + s_bin[i]], 1u);
    order[begin + slot] = i - begin;
  }
  if (threadIdx.x == 0) plan[1] = 0u;  // ready for the next call that uses this buffer
}

// ---------------------------------------------------------------- launchers ----
static inline int grid_for(uint64_t n, int block = 256, int cap = 256 * 8) {
  uint64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (uint64_t)cap) g = cap;
  return (int)g;
}

static int gen_lds_bytes(const DevModel& M, int waves) { return LDS_LEVEL_BYTES + waves * ((int)sizeof(WaveLds) + (int)M.gen_wave_bytes); }

hipError_t launch_render(const DevModel& M, const FrameParams& Pin, const ViewBatch& VBin, void* rgba, void* depth, void* counters,
                         hipStream_t st, bool first_launch, unsigned* plan, unsigned plan_cap) {
  ViewBatch VB = VBin;
  FrameParams P = Pin;
  P.plan_order = nullptr;
  // the call's statistics counters and queue words are zero before its first launch (plan_price_kernel does it for a planned launch)
  auto clear_for_first = [&]() { return first_launch ? hipMemsetAsync(counters, 0, COUNTER_BYTES + RENDER_QUEUE_BYTES, st) : hipSuccess; };
  VB.blocks_per_view = (P.n_local_tiles + RENDER_WAVES - 1) / RENDER_WAVES;
  if (VB.blocks_per_view <= 0 || VB.n_views <= 0) return clear_for_first();  // a shard without a strip (tiny frames, many ranks)
  if (VB.n_views > MAX_VIEWS) return hipErrorInvalidValue;
  const int blocks = VB.blocks_per_view * VB.n_views;
  const bool perturb = P.perturb != 0;  // (render_kernel's PERTURB instances: per-strip workgroups, tables in global memory)
  const bool lds_tab = M.lds_coarse_words > 0 && !perturb;
  if (M.persistent && lds_tab) {
    // work queues: per view the strip rows its region of interest touches (sharded: the local strips of those rows)
    const int strips_x = (P.tiles_x + 3) >> 2, N = P.shard_count, idx = P.shard_index;
    const int k_end = (P.n_local_tiles + 3) & ~3;
    int q = 0;
    for (int v = 0; v < VB.n_views; ++v) {
      ViewParams& V = VB.v[v];
      V.k_lo = V.k_hi = 0;
      int rows = 0, row0 = 0;
      if (V.roi[2] >= V.roi[0] && V.roi[3] >= V.roi[1]) {
        const int ty0 = std::max(V.roi[1] >> 3, 0), ty1 = std::min(V.roi[3] >> 3, P.tiles_y - 1);
        if (ty1 >= ty0) {
          rows = ty1 - ty0 + 1;
          row0 = ty0;
          const int s0 = ty0 * strips_x, s1 = (ty1 + 1) * strips_x;  // global strips [s0, s1)
          const int ls0 = s0 > idx ? (s0 - idx + N - 1) / N : 0, ls1 = s1 > idx ? (s1 - idx + N - 1) / N : 0;
          V.k_lo = std::min(4 * ls0, k_end);
          V.k_hi = std::min(4 * ls1, k_end);
        }
      }
      V.q_begin = q;
      V.q_rows = rows;
      V.q_row0 = row0;
      q += rows;  // units: the strip rows the region of interest touches
    }
    VB.q_total = q;
    VB.n_classes = P.queue_classes >= 1 && P.queue_classes <= 8 ? P.queue_classes : 8;
    VB.class_cols = (strips_x + N - 1) / N;  // a row holds at most this many of the rank's strips
    if ((long long)q * VB.class_cols >= 0xffffff) return hipErrorInvalidValue;  // 24-bit queue positions
    const int waves = (int)M.persist_waves;
    const int lds = (M.wide_sh ? render_persistent_lds_widesh_bytes() : M.hot_width ? render_persistent_lds_width_bytes((int)M.hot_width)
                     : M.hot_grid ? render_persistent_lds_fixed_bytes(0u, 0u, 0u, 16)  // (the hot instance's workgroup)
                                 : render_persistent_lds_fixed_bytes(M.generic, M.wide, M.gen_wave_bytes, waves)) +
                    4 * (int)(M.lds_coarse_words + M.lds_ctab_floats + M.lds_dilated_words) +
                    (M.gen_weights_lds ? 16 + (int)M.gen_frag_bytes : 0);
    const long long tiles = (long long)P.n_local_tiles * VB.n_views;
    const int wgs = (int)std::max(1LL, std::min((long long)M.n_cus, (tiles + waves - 1) / waves));
    unsigned* queue = reinterpret_cast<unsigned*>((unsigned long long*)counters + COUNTER_SLOTS * 16);
    hipError_t e = hipSuccess;
    const long long n_pos = (long long)q * VB.class_cols;
    const size_t dil_bytes = (size_t)4 * M.dilated_level_words * M.cascade;
    if (first_launch && plan != nullptr && M.occ_dilated != nullptr && n_pos > 0 && n_pos <= (long long)plan_cap && n_pos <= 60 * 1024 &&
        dil_bytes <= 60 * 1024) {
      const int blocks = (int)((n_pos * 4 + PLAN_THREADS - 1) / PLAN_THREADS);
      hipLaunchKernelGGL(plan_price_kernel, dim3(blocks), dim3(PLAN_THREADS), (dil_bytes + 15) & ~(size_t)15, st, M, P, VB, plan,
                         (unsigned*)counters, (unsigned)((COUNTER_BYTES + RENDER_QUEUE_BYTES) / 4));
      hipLaunchKernelGGL(plan_sort_kernel, dim3(1), dim3(PLAN_SORT_THREADS), ((size_t)n_pos + 15) & ~(size_t)15, st, VB, plan, plan_cap);
      P.plan_order = plan + 4 + plan_cap;
    } else {
      e = first_launch ? clear_for_first() : hipMemsetAsync(queue, 0, RENDER_QUEUE_BYTES, st);
    }
    if (e != hipSuccess) return e;
    const int form = march_form(M.H, M.cascade, M.bound);
    const bool unit = form == MARCH_FORM_UNIT, pow2 = form == MARCH_FORM_POW2;
    const PersistLaunch L{&M, &P, &VB, rgba, depth, counters, queue, st, lds, wgs, waves, unit, pow2};
    if (M.wide_sh || M.wide) e = launch_persistent_wide(L);   // Frequency / SH directions beyond 16 values (nrf_kernels_wide.hip)
    else if (M.hot_width) e = launch_persistent_width(L);     // 16 / 32 / 128 neurons, other depths (nrf_kernels_width.hip)
    else if (M.hot_grid) e = launch_persistent_grid(L);       // other grids in front of base.json's MLPs (nrf_kernels_grid.hip)
    else if (M.generic) e = launch_persistent_generic(L);     // nrf_kernels_generic.hip
    else e = launch_persistent_hot(L);                        // the base.json shape (nrf_kernels_hot.hip)
    if (e != hipSuccess) return e;
    return hipGetLastError();
  }
  {
    const hipError_t e0 = clear_for_first();
    if (e0 != hipSuccess) return e0;
  }
  const int fixed = M.generic ? gen_lds_bytes(M, RENDER_WAVES)
                              : (M.wide ? LDS_FIXED_BYTES + (LDS_WFRAG_WIDE_BYTES - LDS_WFRAG_BYTES) + RENDER_WAVES * LDS_RAYD_BYTES
                                        : LDS_FIXED_BYTES);
  const int lds = fixed + (lds_tab ? 4 * (int)(M.lds_coarse_words + M.lds_ctab_floats) : 0);
  // hot instances: compile-time activations, march tables in LDS; a power-of-two grid with either one cascade and
  // mip_bound == 1 (MARCH_UNIT) or several cascades and a power-of-two bound (MARCH_POW2)
  const int form = march_form(M.H, M.cascade, M.bound);
  const bool unit = lds_tab && form == MARCH_FORM_UNIT, pow2 = lds_tab && form == MARCH_FORM_POW2;
  const hipError_t es = launch_strip(StripLaunch{&M, &P, &VB, rgba, depth, counters, st, lds, blocks, lds_tab, unit, pow2, perturb});
  if (es != hipSuccess) return es;
  return hipGetLastError();
}

hipError_t launch_build_quads(const void* table, uint32_t res, uint32_t size, bool hashed, void* quads, hipStream_t st) {
  if (res < 2 || res > 65535u) return hipErrorInvalidValue;  // (grid dimensions y, z)
  const dim3 grid((res + 255u) / 256u, res, res + 1u);
  hipLaunchKernelGGL(build_quads_kernel, grid, dim3(256), 0, st, (const uint32_t*)table, res, size, hashed ? 1u : 0u, (uint4*)quads);
  return hipGetLastError();
}

hipError_t launch_encode_grid(const DevModel& M, const void* pos01, uint32_t n, void* out, hipStream_t st, bool fast_interp) {
  if (!n) return hipSuccess;
  if (M.generic)
    hipLaunchKernelGGL(gen_encode_grid_kernel, dim3(grid_for((uint64_t)n * M.n_levels)), dim3(256), 0, st, M, (const float*)pos01, n,
                       (half_t*)out);
  else if (fast_interp)
    hipLaunchKernelGGL(encode_grid_kernel<true>, dim3(grid_for((uint64_t)n * 16)), dim3(256), 0, st, M, (const float*)pos01, n,
                       (uint32_t*)out);
  else
    hipLaunchKernelGGL(encode_grid_kernel<false>, dim3(grid_for((uint64_t)n * 16)), dim3(256), 0, st, M, (const float*)pos01, n,
                       (uint32_t*)out);
  return hipGetLastError();
}

hipError_t launch_encode_dir(const DevModel& M, const void* dir01, uint32_t n, void* out, hipStream_t st) {
  if (!n) return hipSuccess;
  if (M.generic)
    hipLaunchKernelGGL(gen_encode_dir_kernel, dim3(grid_for(n)), dim3(256), 0, st, M, (const float*)dir01, n, (half_t*)out);
  else
    hipLaunchKernelGGL(encode_dir_kernel, dim3(grid_for(n)), dim3(256), 0, st, M, (const float*)dir01, n, (uint32_t*)out);
  return hipGetLastError();
}

hipError_t launch_mlp_forward(const DevModel& M, const void* feat, const void* dirfeat, uint32_t n, void* out, uint32_t repeat,
                              hipStream_t st) {
  if (!n) return hipSuccess;
  if (M.generic) {
    const int lds = gen_lds_bytes(M, 4);
    hipError_t e = allow_lds(gen_mlp_forward_kernel, lds);
    if (e != hipSuccess) return e;
    const uint64_t chunks = ((uint64_t)n + GEN_SAMPLES - 1) / GEN_SAMPLES;
    for (uint32_t r = 0; r < (repeat ? repeat : 1u); ++r)  // (the repeat count is a measurement aid of the hot instance)
      hipLaunchKernelGGL(gen_mlp_forward_kernel, dim3(grid_for(chunks, 4, 256 * 4)), dim3(256), lds, st, M, (const half_t*)feat,
                         (const half_t*)dirfeat, n, (uint2*)out);
    return hipGetLastError();
  }
  const uint64_t chunks = ((uint64_t)n + 16 * MLP_TILES - 1) / (16 * MLP_TILES);
  // the kernel's form: workgroups per CU its registers are budgeted for x how the prefetch buffers rotate (mlp_forward_kernel).
  // NRF_MLP_FORM = 30 (3 per CU, copies: rounds 1-4) | 20 | 21 (2 per CU, by name) -- A/B runs; default: MLP_FORM_DEFAULT.
  // (3 per CU by name needs 176 registers: built, it spilled 8-11 of them -- not shipped)
  static const int form = [] {
    const char* e = std::getenv("NRF_MLP_FORM");
    const int f = e ? std::atoi(e) : MLP_FORM_DEFAULT;
    return (f == 30 || f == 20 || f == 21) ? f : MLP_FORM_DEFAULT;
  }();
#define NRF_LAUNCH_MLP(R, OCC, NAMED)                                                                                     \
  hipLaunchKernelGGL((mlp_forward_kernel<R, OCC, NAMED>), dim3(grid_for(chunks, 4, 256 * OCC)), dim3(256), 0, st, M,      \
                     (const uint4*)feat, (const uint2*)dirfeat, n, (half_t*)out, repeat)
#define NRF_LAUNCH_MLP_FORM(R)                                                                                            \
  do {                                                                                                                    \
    if (form == 20) NRF_LAUNCH_MLP(R, 2, false);                                                                          \
    else if (form == 21) NRF_LAUNCH_MLP(R, 2, true);                                                                      \
    else NRF_LAUNCH_MLP(R, 3, false);                                                                                     \
  } while (0)
  if (repeat > 1) NRF_LAUNCH_MLP_FORM(true); else NRF_LAUNCH_MLP_FORM(false);
#undef NRF_LAUNCH_MLP_FORM
#undef NRF_LAUNCH_MLP
  return hipGetLastError();
}

hipError_t launch_network(const DevModel& M, const void* xyz, const void* dir, uint32_t n, void* sigma, void* rgb, hipStream_t st) {
  if (!n) return hipSuccess;
  const uint64_t chunks = ((uint64_t)n + 63) / 64;
  if (M.generic) {
    const int lds = gen_lds_bytes(M, 4);
    hipError_t e = allow_lds(network_kernel<NET_GENERIC>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(network_kernel<NET_GENERIC>, dim3(grid_for(chunks, 4, 256 * 4)), dim3(256), lds, st, M,
                       (const float*)xyz, (const float*)dir, n, (float*)sigma, (float*)rgb);
  } else if (M.wide) {
    const int lds = LDS_TOTAL_BYTES + (LDS_WFRAG_WIDE_BYTES - LDS_WFRAG_BYTES) + 4 * LDS_RAYD_BYTES;
    hipLaunchKernelGGL(network_kernel<NET_WIDE>, dim3(grid_for(chunks, 4, 256 * 4)), dim3(256), lds, st, M,
                       (const float*)xyz, (const float*)dir, n, (float*)sigma, (float*)rgb);
  } else {
    hipLaunchKernelGGL(network_kernel<NET_HOT>, dim3(grid_for(chunks, 4, 256 * 4)), dim3(256), LDS_TOTAL_BYTES, st, M,
                       (const float*)xyz, (const float*)dir, n, (float*)sigma, (float*)rgb);
  }
  return hipGetLastError();
}

hipError_t launch_density_positions(uint32_t H, float k, void* xyz, void* dir, hipStream_t st) {
  hipLaunchKernelGGL(density_positions_kernel, dim3(grid_for((uint64_t)H * H * H)), dim3(256), 0, st, H, k, (float*)xyz, (float*)dir);
  return hipGetLastError();
}

hipError_t launch_density_update(const void* sigma, uint32_t n, float decay, int n_iterations, void* grid, hipStream_t st) {
  hipLaunchKernelGGL(density_update_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float*)sigma, n, decay, n_iterations,
                     (float*)grid);
  return hipGetLastError();
}

hipError_t launch_generate_rays(const DevModel& M, const FrameParams& P, void* rays_o, void* rays_d, void* nears, void* fars,
                                hipStream_t st) {
  hipLaunchKernelGGL(generate_rays_kernel, dim3(grid_for((uint64_t)P.W * P.H)), dim3(256), 0, st, M, P, (float*)rays_o,
                     (float*)rays_d, (float*)nears, (float*)fars);
  return hipGetLastError();
}

hipError_t launch_march(const DevModel& M, float dt_gamma, const void* rays_o, const void* rays_d, const void* rays_t,
                        const void* fars, uint32_t n, uint32_t n_step, void* xyzs, void* dirs, void* deltas, hipStream_t st, uint32_t perturb) {
  if (!n) return hipSuccess;
  if (M.coarse_shift)
    hipLaunchKernelGGL(march_kernel<true>, dim3(grid_for(n)), dim3(256), 0, st, M, dt_gamma, (const float*)rays_o,
                       (const float*)rays_d, (const float*)rays_t, (const float*)fars, n, n_step, (float*)xyzs, (float*)dirs,
                       (float*)deltas, perturb);
  else
    hipLaunchKernelGGL(march_kernel<false>, dim3(grid_for(n)), dim3(256), 0, st, M, dt_gamma, (const float*)rays_o,
                       (const float*)rays_d, (const float*)rays_t, (const float*)fars, n, n_step, (float*)xyzs, (float*)dirs,
                       (float*)deltas, perturb);
  return hipGetLastError();
}

hipError_t launch_composite(const void* sigmas, const void* rgbs, const void* deltas, uint32_t n, uint32_t n_step, void* rays_t,
                            void* state, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(composite_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float*)sigmas, (const float*)rgbs,
                     (const float*)deltas, n, n_step, (float*)rays_t, (float*)state);
  return hipGetLastError();
}

hipError_t launch_untile(const void* gathered, int shard_count, int tiles_per_shard, int C, int W, int H, int n_views, void* out,
                         hipStream_t st) {
  hipLaunchKernelGGL(untile_kernel, dim3(grid_for((uint64_t)W * H * n_views)), dim3(256), 0, st, (const float*)gathered, shard_count,
                     tiles_per_shard, C, W, H, (W + 7) / 8, n_views, (float*)out);
  return hipGetLastError();
}

// The reference's output format packed per pixel: r | g << 8 | b << 16 | depth << 24 (any buffer layout)
__global__ __launch_bounds__(256) void quantize_rgbd8_kernel(const float4* __restrict__ rgba, const float* __restrict__ depth,
                                                             size_t n, uint32_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = rgba[i];
    out[i] = (uint32_t)quant_u8(v.x) | ((uint32_t)quant_u8(v.y) << 8) | ((uint32_t)quant_u8(v.z) << 16) |
             ((uint32_t)quant_u8(depth[i]) << 24);
  }
}

hipError_t launch_quantize_rgbd8(const void* rgba, const void* depth, uint64_t n, void* out, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(quantize_rgbd8_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float4*)rgba, (const float*)depth, (size_t)n,
                     (uint32_t*)out);
  return hipGetLastError();
}

// gathered packed pixels [shard][view][tiles_per_shard][64] (r | g << 8 | b << 16 | depth << 24, what the members of a device
// group render with nrf_bind_output_rgbd8) -> the reference's host Image layout: rgb u8 [view][H][W][3], depth u8 [view][H][W].
// One thread per 4 horizontally adjacent pixels (they lie in one tile row: 16 contiguous bytes of the shard) when the
// width is a multiple of 4: three dword stores of rgb and one of depth; else one thread per pixel with byte stores.
__global__ __launch_bounds__(256) void untile_rgbd8_u8_kernel(const uint32_t* __restrict__ gathered, int shard_count, int tiles_per_shard,
                                                              int W, int H, int tiles_x, int n_views, unsigned char* __restrict__ rgb8,
                                                              unsigned char* __restrict__ depth8) {
  const size_t frame = (size_t)W * H;
  const int strips_x = (tiles_x + 3) >> 2;
  auto src_of = [&](int view, int px, int py) {
    const int tx = px >> 3, ty = py >> 3;
    const int strip = ty * strips_x + (tx >> 2);
    const int shard = strip % shard_count, k = (strip / shard_count) * 4 + (tx & 3);
    return gathered + (((size_t)shard * n_views + view) * tiles_per_shard + k) * 64 + (py & 7) * 8 + (px & 7);
  };
  if ((W & 3) == 0) {
    const size_t quads = frame / 4, total = quads * (size_t)n_views;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int view = (int)(i / quads);
      const size_t p = (i - (size_t)view * quads) * 4;
      const uint4 v = *reinterpret_cast<const uint4*>(src_of(view, (int)(p % W), (int)(p / W)));
      uint32_t* o = reinterpret_cast<uint32_t*>(rgb8 + ((size_t)view * frame + p) * 3);
      o[0] = (v.x & 0xffffffu) | (v.y << 24);
      o[1] = ((v.y & 0xffffffu) >> 8) | (v.z << 16);
      o[2] = ((v.z & 0xffffffu) >> 16) | (v.w << 8);
      *reinterpret_cast<uint32_t*>(depth8 + (size_t)view * frame + p) = (v.x >> 24) | ((v.y >> 24) << 8) | ((v.z >> 24) << 16) | ((v.w >> 24) << 24);
    }
  } else {
    const size_t total = frame * (size_t)n_views;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int view = (int)(i / frame);
      const size_t p = i - (size_t)view * frame;
      const uint32_t v = *src_of(view, (int)(p % W), (int)(p / W));
      rgb8[3 * i] = (unsigned char)(v & 0xffu);
      rgb8[3 * i + 1] = (unsigned char)((v >> 8) & 0xffu);
      rgb8[3 * i + 2] = (unsigned char)((v >> 16) & 0xffu);
      depth8[i] = (unsigned char)(v >> 24);
    }
  }
}

hipError_t launch_untile_rgbd8_u8(const void* gathered, int shard_count, int tiles_per_shard, int W, int H, int n_views, void* rgb8,
                                  void* depth8, hipStream_t st) {
  const uint64_t work = (uint64_t)W * H * n_views / ((W & 3) == 0 ? 4 : 1);
  hipLaunchKernelGGL(untile_rgbd8_u8_kernel, dim3(grid_for(work)), dim3(256), 0, st, (const uint32_t*)gathered, shard_count,
                     tiles_per_shard, W, H, (W + 7) / 8, n_views, (unsigned char*)rgb8, (unsigned char*)depth8);
  return hipGetLastError();
}

hipError_t launch_quantize(const void* rgba, const void* depth, int n, void* rgb8, void* depth8, hipStream_t st) {
  hipLaunchKernelGGL(quantize_kernel, dim3(grid_for((uint64_t)n)), dim3(256), 0, st, (const float4*)rgba, (const float*)depth, n,
                     (unsigned char*)rgb8, (unsigned char*)depth8);
  return hipGetLastError();
}

// Every translation unit is a code object of its own, and the HIP runtime loads a code object when one of its kernels is first
// launched (a millisecond or more: bench.py's first single-view frame measured 2 ms instead of 0.88).  `all` = every family;
// otherwise the families a base.json-shaped model launches (this unit's planning / stage kernels, the hot family).
void preload_kernels(bool all) {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(plan_price_kernel));
  preload_hot();
  if (all) {
    preload_width();
    preload_wide();
    preload_generic();
    preload_strip();
    preload_grid();
  }
}

int render_lds_bytes() { return LDS_FIXED_BYTES; }
// LDS of the persistent workgroup of `waves` waves without its march tables (lds_map<NET> + the queue words)
int render_persistent_lds_fixed_bytes(uint32_t generic, uint32_t wide, uint32_t gen_wave_bytes, int waves) {
  if (generic) return LDS_LEVEL_BYTES + waves * ((int)sizeof(WaveLds) + (int)gen_wave_bytes) + LDS_QUEUE_BYTES;
  if (wide) return LDS_WFRAG_WIDE_BYTES + LDS_LEVEL_BYTES + waves * ((int)sizeof(WaveLds) + LDS_RAYD_BYTES) + LDS_QUEUE_BYTES;
  return LDS_WFRAG_BYTES + LDS_LEVEL_BYTES + waves * (int)sizeof(WaveLds) + LDS_QUEUE_BYTES;
}
// ... of the 16-wave workgroup of a width instance (NET_W16 / NET_W32 / NET_W128)
static int width_net(int width) {
  return width == 16 ? NET_W16 : (width == 32 ? NET_W32 : (width == 64 ? NET_DEPTH : (width == (int)HOT_WIDTH_ACT ? NET_ACT : NET_W128)));
}
int render_persistent_lds_width_bytes(int width) {
  const int net = width_net(width);
  return net_wfrag_bytes(net) + LDS_LEVEL_BYTES + persist_waves(net) * (int)sizeof(WaveLds) + LDS_QUEUE_BYTES;
}
int render_persist_waves_for(uint32_t generic, uint32_t wide, uint32_t wide_sh, uint32_t hot_width, uint32_t hot_grid, int form) {
  if (wide_sh) return persist_waves(NET_WIDE_SH);
  if (hot_grid) return persist_waves(hot_grid == 1 ? NET_GRID1 : (hot_grid == 2 ? NET_GRID2 : (hot_grid == 4 ? NET_GRID4 : NET_GRID8)));
  if (hot_width) return persist_waves(width_net((int)hot_width));
  if (wide && !generic && form == MARCH_FORM_GENERIC) return WIDE_GENERIC_MARCH_WAVES;  // (nrf_kernels_wide.hip)
  return persist_waves(generic ? NET_GENERIC : (wide ? NET_WIDE : NET_HOT));
}
int render_persistent_lds_widesh_bytes() {  // the 8-wave workgroup of NET_WIDE_SH without its march tables
  return net_wfrag_bytes(NET_WIDE_SH) + LDS_LEVEL_BYTES + persist_waves(NET_WIDE_SH) * ((int)sizeof(WaveLds) + LDS_SHROW_BYTES) + LDS_QUEUE_BYTES;
}
int render_width_frags(int width) { return width == 16 ? MlpShape<16>::N : (width == 32 ? MlpShape<32>::N : (width == 128 ? MlpShape<128>::N : N_FRAGS)); }
int render_wide_lds_fixed_bytes() { return LDS_FIXED_BYTES + (LDS_WFRAG_WIDE_BYTES - LDS_WFRAG_BYTES) + RENDER_WAVES * LDS_RAYD_BYTES; }
int render_lds_table_max_bytes() { return LDS_MARCH_TABLE_MAX; }
int render_gen_lds_fixed_bytes(uint32_t gen_wave_bytes) { return LDS_LEVEL_BYTES + RENDER_WAVES * ((int)sizeof(WaveLds) + (int)gen_wave_bytes); }


}  // namespace nrf
