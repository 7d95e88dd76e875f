// Reproducer attempt for the packed-fp32 hazard of DESIGN.md: does v_pk_mul_f32 see a stale source in
// lanes 48-63 when the source was written just before by a transcendental (quarter-rate) instruction?
//   hipcc --offload-arch=gfx950 -O2 -o pk_hazard pk_hazard.hip && ./pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad_lane, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float x = in[tid];
  unsigned bad = 0;
  for (int i = 0; i < iters; ++i) {
    float got, want;
    // reference: rcp, generous wait, plain multiply
    asm volatile("v_rcp_f32 %0, %1\n\ts_nop 7\n\ts_nop 7\n\tv_mul_f32 %0, %0, %0" : "=&v"(want) : "v"(x));
    if (VARIANT == 0)  // trans result consumed by the packed op right away
      asm volatile("v_mov_b32 v11, 1.0\n\tv_rcp_f32 v10, %1\n\tv_pk_mul_f32 v[12:13], v[10:11], v[10:11]\n\ts_nop 7\n\tv_mov_b32 %0, v12"
                   : "=v"(got) : "v"(x) : "v10", "v11", "v12", "v13");
    else if (VARIANT == 1)  // one independent VALU instruction in between
      asm volatile("v_mov_b32 v11, 1.0\n\tv_rcp_f32 v10, %1\n\tv_mov_b32 v13, 0\n\tv_pk_mul_f32 v[12:13], v[10:11], v[10:11]\n\ts_nop 7\n\tv_mov_b32 %0, v12"
                   : "=v"(got) : "v"(x) : "v10", "v11", "v12", "v13");
    else if (VARIANT == 2)  // s_nop 1 in between
      asm volatile("v_mov_b32 v11, 1.0\n\tv_rcp_f32 v10, %1\n\ts_nop 1\n\tv_pk_mul_f32 v[12:13], v[10:11], v[10:11]\n\ts_nop 7\n\tv_mov_b32 %0, v12"
                   : "=v"(got) : "v"(x) : "v10", "v11", "v12", "v13");
    else if (VARIANT == 3)  // same pattern with a plain (non-packed) consumer
      asm volatile("v_rcp_f32 v10, %1\n\tv_mul_f32 v12, v10, v10\n\ts_nop 7\n\tv_mov_b32 %0, v12"
                   : "=v"(got) : "v"(x) : "v10", "v12");
    else  // packed consumer of a NON-trans producer
      asm volatile("v_mov_b32 v11, 1.0\n\tv_add_f32 v10, %1, %1\n\tv_pk_mul_f32 v[12:13], v[10:11], v[10:11]\n\ts_nop 7\n\tv_mov_b32 %0, v12\n\t"
                   "v_add_f32 v10, %1, %1\n\ts_nop 7\n\tv_mul_f32 v10, v10, v10\n\ts_nop 7\n\tv_mov_b32 %0, v12"
                   : "=v"(got) : "v"(x) : "v10", "v11", "v12", "v13");
    if (VARIANT == 4) {
      float t = x + x;
      want = t * t;
    }
    if (__float_as_uint(got) != __float_as_uint(want)) bad++;
    x = x * 1.0000001f + 1e-7f;
  }
  if (bad) atomicAdd(&bad_lane[threadIdx.x & 63], bad);
}

template <int V>
void run(const char* what, const float* d_in, unsigned* d_bad, int blocks, int iters) {
  hipMemset(d_bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d_in, d_bad, iters);
  hipDeviceSynchronize();
  unsigned h[64];
  hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += h[l];
  printf("%-62s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu %lu %lu %lu\n", what, q[0], q[1], q[2], q[3]);
}

int main() {
  const int blocks = 256 * 16, n = blocks * 256, iters = 20000;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0.5f + (float)(i % 9973) / 9973.0f;
  float* d_in;
  unsigned* d_bad;
  hipMalloc(&d_in, n * sizeof(float));
  hipMalloc(&d_bad, 64 * sizeof(unsigned));
  hipMemcpy(d_in, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  run<3>("v_rcp -> v_mul_f32 (plain consumer), back to back", d_in, d_bad, blocks, iters);
  run<0>("v_rcp -> v_pk_mul_f32, back to back", d_in, d_bad, blocks, iters);
  run<1>("v_rcp -> 1 VALU -> v_pk_mul_f32", d_in, d_bad, blocks, iters);
  run<2>("v_rcp -> s_nop 1 -> v_pk_mul_f32", d_in, d_bad, blocks, iters);
  run<4>("v_add -> v_pk_mul_f32 (non-trans producer), back to back", d_in, d_bad, blocks, iters);
  return 0;
}
