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
  return 0;
}
