// mlp_probe.hip -- variants of the register-resident MLP chain (nrf_device.h mlp_tiles) timed in isolation: what does
// the fused-MLP kernel lose to LDS weight reads, to the tile count per pass, to waves per SIMD?  Every variant evaluates
// the same 80 MFMAs per 64 samples (20 480 FLOP per sample) `rep` times per chunk from registers; TFLOP/s against the
// 2.5 PFLOP/s dense fp16 peak.  Weights and inputs are random; only the rate matters here (parity: tests/).
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I../../include -o mlp_probe mlp_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../nerf-cuda_amd/csrc/nrf_device.h"

using namespace nrf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct RegFrags {
  const half8_t* w;  // N_FRAGS_ALL fragments of this lane, in registers
  __device__ __forceinline__ half8_t operator()(int f) const { return w[f]; }
};

// NT tiles per pass, weights from LDS (REG == false) or hoisted into registers (REG == true)
template <int NT, bool REG, int MIN_BLOCKS>
__global__ __launch_bounds__(256, MIN_BLOCKS) void probe(const uint4* __restrict__ wfrag, const uint16_t* __restrict__ lut, const uint4* __restrict__ feat,
                                                          const uint2* __restrict__ dirfeat, uint32_t n_chunks, uint32_t rep,
                                                          uint2* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* wl = reinterpret_cast<uint4*>(smem);
  for (int i = threadIdx.x; i < N_FRAGS_ALL * 64; i += blockDim.x) wl[i] = wfrag[i];
  __syncthreads();
  const int lane = lane_id(), g = lane >> 4, c = lane & 15;
  half8_t wreg[N_FRAGS_ALL];
  if (REG) {
#pragma unroll
    for (int f = 0; f < N_FRAGS_ALL; ++f) wreg[f] = frag_load(wl, f, lane);
  }
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
    uint4 fv[NT];
    uint2 dv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint32_t s = (chunk * NT + t) * 16u + c;
      fv[t] = feat[(size_t)s * 4 + g];
      dv[t] = dirfeat[(size_t)s * 4 + g];
    }
    MlpOut<NT> o;
    for (uint32_t r = 0; r < rep; ++r) {
      half8_t f[NT];
      half4_t df[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        asm volatile("" : "+v"(fv[t].x), "+v"(fv[t].y), "+v"(fv[t].z), "+v"(fv[t].w), "+v"(dv[t].x), "+v"(dv[t].y));
        f[t] = __builtin_bit_cast(half8_t, fv[t]);
        df[t] = __builtin_bit_cast(half4_t, dv[t]);
      }
      if constexpr (REG) mlp_tiles<NT, FRAG_D0_NATURAL>(RegFrags{wreg}, f, df, o);
      else mlp_tiles<NT, FRAG_D0_NATURAL>(LdsFrags{wl, lane}, f, df, o);
    }
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) out[(chunk * NT + t) * 16u + c] = make_uint2(o.rg[t], o.bx[t]);
    }
    if (g < NT) reinterpret_cast<half_t*>(out)[4 * ((chunk * NT + g) * 16u + c) + 3] = o.sigma;
  }
}

// The same MLP shape on v_mfma_f32_32x32x16_f16 (one tile = 32 samples, lane = (half, sample)): 24 MFMAs of 32 cycles
// per 32 samples instead of 40 of 16 (the 16-row output layers are padded to 32 rows: 24 576 FLOP issued per sample
// for the 20 480 counted), the same 104 conversions / ReLUs per 32 samples.  The data flow is the real one (every
// layer's D fragment re-packed in-lane as the next layer's B fragments); the K permutations that would make the values
// right live in the weight fragments, which are random here -- only the rate is measured.
typedef float float16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void pack16(const float16_t& a, half8_t& lo, half8_t& hi) {
  const half8_t z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) { lo[i] = (half_t)a[i]; hi[i] = (half_t)a[8 + i]; }
  lo = __builtin_elementwise_max(lo, z);
  hi = __builtin_elementwise_max(hi, z);
}
template <int MIN_BLOCKS>
__global__ __launch_bounds__(256, MIN_BLOCKS) void probe32(const uint4* __restrict__ wfrag, const uint4* __restrict__ feat,
                                                           const uint2* __restrict__ dirfeat, uint32_t n_chunks, uint32_t rep,
                                                           uint2* __restrict__ out) {
  const int lane = lane_id();
  half8_t w[24];
#pragma unroll
  for (int f = 0; f < 24; ++f) w[f] = __builtin_bit_cast(half8_t, wfrag[f * 64 + lane]);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  const float16_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t chunk = wave; chunk < n_chunks; chunk += n_waves) {
    const uint32_t s = chunk * 32u + (lane & 31);
    uint4 f0 = feat[(size_t)s * 4 + (lane >> 5)], f1 = feat[(size_t)s * 4 + 2 + (lane >> 5)];
    uint4 dv = make_uint4(dirfeat[(size_t)s * 4 + (lane >> 5)].x, dirfeat[(size_t)s * 4 + (lane >> 5)].y,
                          dirfeat[(size_t)s * 4 + 2 + (lane >> 5)].x, dirfeat[(size_t)s * 4 + 2 + (lane >> 5)].y);
    float16_t o = zero;
    for (uint32_t r = 0; r < rep; ++r) {
      asm volatile("" : "+v"(f0.x), "+v"(f0.y), "+v"(f0.z), "+v"(f0.w), "+v"(f1.x), "+v"(f1.y), "+v"(f1.z), "+v"(f1.w), "+v"(dv.x), "+v"(dv.y));
      const half8_t b0 = __builtin_bit_cast(half8_t, f0), b1 = __builtin_bit_cast(half8_t, f1);
      float16_t acc[2];
      half8_t hb[4];
      // density 32 -> 64
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * m], b0, zero, 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * m + 1], b1, acc[m], 0, 0, 0);
      }
      pack16(acc[0], hb[0], hb[1]);
      pack16(acc[1], hb[2], hb[3]);
      // density 64 -> 16 (padded to 32 rows)
      float16_t d = zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[4 + k], hb[k], d, 0, 0, 0);
      half8_t rin0;
#pragma unroll
      for (int i = 0; i < 8; ++i) rin0[i] = (half_t)d[i];
      const half8_t rin1 = __builtin_bit_cast(half8_t, dv);
      // rgb 32 -> 64
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[8 + 2 * m], rin0, zero, 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[8 + 2 * m + 1], rin1, acc[m], 0, 0, 0);
      }
      pack16(acc[0], hb[0], hb[1]);
      pack16(acc[1], hb[2], hb[3]);
      // rgb 64 -> 64
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[m] = zero;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[12 + 4 * m + k], hb[k], acc[m], 0, 0, 0);
      }
      half8_t hc[4];
      pack16(acc[0], hc[0], hc[1]);
      pack16(acc[1], hc[2], hc[3]);
      // rgb 64 -> 16 (padded)
      o = zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[20 + k], hc[k], o, 0, 0, 0);
      f0.x ^= (uint32_t)(o[0] == 12345.0f);
    }
    if ((lane >> 5) == 0) out[s] = make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], expf(o[3])));
  }
}
template <int MIN_BLOCKS>
void run32(const char* name, const uint4* w, const uint4* feat, const uint2* dirf, uint32_t n, uint32_t rep, uint2* out, int blocks_per_cu) {
  const uint32_t n_chunks = n / 32;
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((probe32<MIN_BLOCKS>), dim3(blocks), dim3(256), 0, 0, w, feat, dirf, n_chunks, 1u, out);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe32<MIN_BLOCKS>), dim3(blocks), dim3(256), 0, 0, w, feat, dirf, n_chunks, rep, out);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double tf = (double)n * rep * 20480.0 / (ms * 1e-3) / 1e12;  // the USEFUL flops, as everywhere else
  std::printf("%-52s %8.3f ms  %8.1f TFLOP/s  %.3f of 2.5 PF\n", name, ms, tf, tf / 2500.0);
}

template <int NT, bool REG, int MIN_BLOCKS>
void run(const char* name, const uint4* w, const uint16_t* lut, const uint4* feat, const uint2* dirf, uint32_t n, uint32_t rep, uint2* out, int blocks_per_cu) {
  const uint32_t n_chunks = n / (16 * NT);
  const int lds = N_FRAGS_ALL * 64 * 16;
  const int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL((probe<NT, REG, MIN_BLOCKS>), dim3(blocks), dim3(256), lds, 0, w, lut, feat, dirf, n_chunks, 1u, out);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((probe<NT, REG, MIN_BLOCKS>), dim3(blocks), dim3(256), lds, 0, w, lut, feat, dirf, n_chunks, rep, out);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double tf = (double)n * rep * 20480.0 / (ms * 1e-3) / 1e12;
  std::printf("%-52s %8.3f ms  %8.1f TFLOP/s  %.3f of 2.5 PF\n", name, ms, tf, tf / 2500.0);
}

int main() {
  const uint32_t n = 1u << 22, rep = 16;
  std::vector<_Float16> hw((size_t)N_FRAGS_ALL * 64 * 8), hf((size_t)n * 32), hd((size_t)n * 16);
  srand(1);
  for (auto& v : hw) v = (_Float16)((rand() % 2001 - 1000) * 1e-4f);
  for (auto& v : hf) v = (_Float16)((rand() % 2001 - 1000) * 5e-4f);
  for (auto& v : hd) v = (_Float16)((rand() % 2001 - 1000) * 5e-4f);
  std::vector<uint16_t> hl(65536);
  for (uint32_t b = 0; b < 65536u; ++b) {
    const uint16_t hb = (uint16_t)b;
    const _Float16 v = (_Float16)expf((float)__builtin_bit_cast(_Float16, hb));
    hl[b] = __builtin_bit_cast(uint16_t, v);
  }
  void *w, *f, *d, *o, *l;
  CK(hipMalloc(&l, 65536 * 2));
  CK(hipMemcpy(l, hl.data(), 65536 * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&f, hf.size() * 2)); CK(hipMalloc(&d, hd.size() * 2)); CK(hipMalloc(&o, (size_t)n * 8));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(f, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d, hd.data(), hd.size() * 2, hipMemcpyHostToDevice));
#define RUN(NT, REG, MB, BPC) run<NT, REG, MB>("NT=" #NT " reg=" #REG " launch_bounds(256," #MB ") blocks/CU=" #BPC, (const uint4*)w, (const uint16_t*)l, (const uint4*)f, (const uint2*)d, n, rep, (uint2*)o, BPC)
  RUN(2, false, 4, 4);
  RUN(2, false, 3, 3);
  RUN(2, false, 2, 2);
  RUN(2, true, 2, 2);
  RUN(2, true, 3, 3);
  RUN(1, false, 4, 4);
  RUN(1, false, 6, 6);
  RUN(1, true, 3, 3);
  RUN(4, false, 2, 2);
  RUN(4, false, 3, 3);
  RUN(4, true, 2, 2);
  RUN(4, true, 1, 1);
  RUN(2, false, 1, 1);
  RUN(4, false, 1, 1);
  run32<2>("32x32x16 chain, registers, launch_bounds(256,2), 2/CU", (const uint4*)w, (const uint4*)f, (const uint2*)d, n, rep, (uint2*)o, 2);
  run32<3>("32x32x16 chain, registers, launch_bounds(256,3), 3/CU", (const uint4*)w, (const uint4*)f, (const uint2*)d, n, rep, (uint2*)o, 3);
  run32<4>("32x32x16 chain, registers, launch_bounds(256,4), 4/CU", (const uint4*)w, (const uint4*)f, (const uint2*)d, n, rep, (uint2*)o, 4);
  return 0;
}
