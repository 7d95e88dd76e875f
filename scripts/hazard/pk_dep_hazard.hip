// pk_dep_hazard.hip -- directed probes for the packed-fp32 corruption of DESIGN.md ("lanes 48-63 of one 8x8 tile wrong,
// only with two or more waves per SIMD", seen in the SLP-vectorised build and nowhere since v_pk_*_f32 was banned).
// scripts/hazard/pk_f32_producers.py lists what feeds the packed instructions of that build; besides the documented
// trans-use case (ruled out by pk_hazard.hip) these producer -> consumer shapes have no entry in the CDNA3/4 hazard
// table and rely on hardware interlocks:
//   P1  v_pk_add_f32 -> one independent VALU -> plain VALU reading the HIGH result register   (encode_dir16: x2 | y pair)
//   P2  v_pk_mul_f32 -> v_pk_add_f32 reading both results, back to back                       (86 such pairs per kernel)
//   P3  v_cvt_f32_f64 (quarter rate) -> v_pk_add_f32, back to back                            (ray_dir: i - cx | j - cy)
//   P4  s_mov_b32 of an SGPR pair's high half -> v_pk_add_f32 with that pair as a source, one VALU in between
// Each probe runs the shape and an unhurried reference (s_nop-padded scalar instructions) on changing inputs and counts
// bitwise mismatches per lane quarter; 16 waves per CU (4 per SIMD) all issue it, plus -- second pass -- a v_exp_f32
// "noise" instruction in every trip so that quarter-rate instructions of OTHER waves interleave.
// One bounded run (about a second); this is a directed test, not a soak.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o pk_dep_hazard pk_dep_hazard.hip && ./pk_dep_hazard   (no contraction: the
//   references are separate multiplies and adds, like the packed instructions)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <int P, bool NOISE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, unsigned* __restrict__ bad_lane, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float x = in[tid], y = in[tid] * 0.37f + 0.11f;
  unsigned bad = 0;
  float noise = x;
  for (int i = 0; i < iters; ++i) {
    float got0 = 0.f, got1 = 0.f, want0 = 0.f, want1 = 0.f;
    if (NOISE) asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(noise));
    if (P == 1) {
      // (x + a, y + b) packed, then the high half is copied one slot later (encode_dir16's v_mov_b32 v11, v5)
      asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, -1.0\n\tv_mov_b32 v13, 0.5\n\ts_nop 4\n\t"
                   "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\tv_add_f32 v16, %2, %2\n\tv_mov_b32 %1, v15\n\tv_mov_b32 %0, v14\n\ts_nop 7"
                   : "=&v"(got0), "=&v"(got1) : "v"(x), "v"(y) : "v10", "v11", "v12", "v13", "v14", "v15", "v16");
      want0 = x + -1.0f;
      want1 = y + 0.5f;
    } else if (P == 2) {
      asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, 1.5\n\tv_mov_b32 v13, 0.75\n\ts_nop 4\n\t"
                   "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\tv_pk_add_f32 v[16:17], v[14:15], v[10:11]\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17"
                   : "=&v"(got0), "=&v"(got1) : "v"(x), "v"(y) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
      want0 = x * 1.5f + x;
      want1 = y * 0.75f + y;
    } else if (P == 3) {
      double dx = (double)x, dy = (double)y;
      asm volatile("v_mov_b32 v12, -1.0\n\tv_mov_b32 v13, 0.5\n\ts_nop 4\n\tv_cvt_f32_f64 v10, %2\n\tv_cvt_f32_f64 v11, %3\n\t"
                   "v_pk_add_f32 v[14:15], v[10:11], v[12:13]\n\ts_nop 7\n\tv_mov_b32 %0, v14\n\tv_mov_b32 %1, v15"
                   : "=&v"(got0), "=&v"(got1) : "v"(dx), "v"(dy) : "v10", "v11", "v12", "v13", "v14", "v15");
      want0 = (float)dx + -1.0f;
      want1 = (float)dy + 0.5f;
    } else {
      asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\ts_mov_b32 s20, 0x40000000\n\ts_nop 4\n\t"
                   "s_mov_b32 s21, -1.0\n\tv_add_f32 v16, %2, %2\n\tv_pk_add_f32 v[14:15], v[10:11], s[20:21]\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15"
                   : "=&v"(got0), "=&v"(got1) : "v"(x), "v"(y) : "v10", "v11", "v14", "v15", "v16", "s20", "s21");
      want0 = x + 2.0f;
      want1 = y + -1.0f;
    }
    if (__float_as_uint(got0) != __float_as_uint(want0) || __float_as_uint(got1) != __float_as_uint(want1)) bad++;
    x = x * 1.0000001f + 1e-7f;
    y = y * 0.9999999f + 3e-7f;
  }
  if (NOISE && noise == 12345.0f) bad += 1u << 30;
  if (bad) atomicAdd(&bad_lane[threadIdx.x & 63], bad);
}

template <int P, bool NOISE>
void run(const char* what, const float* d_in, unsigned* d_bad, int blocks, int iters) {
  hipMemset(d_bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL((probe<P, NOISE>), dim3(blocks), dim3(256), 0, 0, d_in, d_bad, iters);
  hipDeviceSynchronize();
  unsigned h[64];
  hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) q[l / 16] += h[l];
  std::printf("%-78s mismatches in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu %lu %lu %lu   (%.1e trials per lane quarter)\n", what, q[0], q[1],
              q[2], q[3], (double)blocks * 256 / 4 * iters);
}

int main() {
  const int blocks = 256 * 4, n = blocks * 256, iters = 40000;  // 16 waves per CU: four per SIMD
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0.5f + (float)(i % 9973) / 9973.0f;
  float* d_in;
  unsigned* d_bad;
  hipMalloc(&d_in, n * sizeof(float));
  hipMalloc(&d_bad, 64 * sizeof(unsigned));
  hipMemcpy(d_in, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  run<1, false>("P1 v_pk_add_f32 -> 1 VALU -> v_mov_b32 of the high result", d_in, d_bad, blocks, iters);
  run<2, false>("P2 v_pk_mul_f32 -> v_pk_add_f32 (dependent, back to back)", d_in, d_bad, blocks, iters);
  run<3, false>("P3 v_cvt_f32_f64 x2 -> v_pk_add_f32 (back to back)", d_in, d_bad, blocks, iters);
  run<4, false>("P4 s_mov_b32 s21 -> 1 VALU -> v_pk_add_f32 with s[20:21]", d_in, d_bad, blocks, iters);
  run<1, true>("P1 with v_exp/v_rcp of every wave interleaved", d_in, d_bad, blocks, iters);
  run<2, true>("P2 with v_exp/v_rcp of every wave interleaved", d_in, d_bad, blocks, iters);
  run<3, true>("P3 with v_exp/v_rcp of every wave interleaved", d_in, d_bad, blocks, iters);
  run<4, true>("P4 with v_exp/v_rcp of every wave interleaved", d_in, d_bad, blocks, iters);
  return 0;
}
