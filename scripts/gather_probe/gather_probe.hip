// gather_probe.hip -- what does the texture path charge for the hash-grid gathers of render_kernel, and would fetching the
// (x, x+1) corner pair as ONE aligned 8-byte load when it happens to be aligned (x even for hashed levels: index(x+1) =
// index(x) ^ 1) relieve it?  Random indices into a 24 MiB table (L2 / Infinity Cache resident, like the real one), 4 waves
// per SIMD, per lane and trip:
//   A  8 x buffer_load_dword                        (today's kernel: 8 corners)
//   B  4 x buffer_load_dwordx2 (aligned)            (lower bound of the pair scheme: every pair aligned)
//   C  4 x buffer_load_dwordx2 (aligned) + 4 x buffer_load_dword under an exec mask that is on for half of the lanes
//   D  4 x buffer_load_dword                        (half the gathers: what a perfect 2x saving would give)
// Reports ms per launch and ns-equivalent TA pressure; build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(const uint32_t* __restrict__ table, uint32_t bytes, int iters, uint32_t* __restrict__ out) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(table), 0, bytes, 0x00020000);
  const uint32_t mask = bytes - 1u;
  uint32_t seed = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    uint32_t off[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      seed = mix(seed + c);
      off[c] = seed & mask;
    }
    if (MODE == 0) {
      uint32_t v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[c] & ~3u, 0, 0);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc ^= v[c];
    } else if (MODE == 1) {
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      u2 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[c] & ~7u, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc ^= v[c].x ^ v[c].y;
#pragma unroll
      for (int c = 4; c < 8; ++c) acc ^= off[c];
    } else if (MODE == 2) {
      typedef uint32_t u2 __attribute__((ext_vector_type(2)));
      u2 v[4];
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[c] & ~7u, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (off[c] & 4u) w[c] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[4 + c] & ~3u, 0, 0);  // half of the lanes
#pragma unroll
      for (int c = 0; c < 4; ++c) acc ^= v[c].x ^ v[c].y ^ w[c];
    } else {
      uint32_t v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[c] & ~3u, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc ^= v[c];
#pragma unroll
      for (int c = 4; c < 8; ++c) acc ^= off[c];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const uint32_t* table, uint32_t bytes, uint32_t* out) {
  const int blocks = 256 * 4 * 2, iters = 2000;
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, table, bytes, 10, out);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, table, bytes, iters, out);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double groups = (double)blocks * 4 * iters;  // wave-trips
  std::printf("%-64s %8.3f ms   %7.1f ns per wave-trip per CU-slot   %6.1f Gtrips/s (lane-trips)\n", name, ms, ms * 1e6 / (groups / 256.0),
              groups * 64 / (ms * 1e-3) / 1e9);
}

int main() {
  const uint32_t bytes = 32u << 20;  // power of two for the mask; 32 MiB
  uint32_t *table, *out;
  CK(hipMalloc(&table, bytes));
  CK(hipMemset(table, 1, bytes));
  CK(hipMalloc(&out, 256 * 4 * 2 * 256 * 4));
  run<0>("A: 8 x dword gather", table, bytes, out);
  run<1>("B: 4 x aligned dwordx2 gather", table, bytes, out);
  run<2>("C: 4 x aligned dwordx2 + 4 x dword under a half-on exec mask", table, bytes, out);
  run<3>("D: 4 x dword gather", table, bytes, out);
  run<0>("A again", table, bytes, out);
  return 0;
}
