// issue_rate.hip -- how many cycles does ONE SIMD of gfx950 need per wave64 instruction, as a function of how many
// waves share it?  (MI355X_MICROARCH.md prices a wave64 v_fma_f32 at 4 cycles for a lone wave and at
// 2 when several waves share the SIMD; DESIGN.md's "VALU issue 93 %" assumed 4.)
//
// Every wave runs `iters` trips of a block of 64 independent instructions (8 accumulators x 8) of one kind and stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.  LDS per workgroup limits the waves per SIMD to
// at most w = 1, 2, 4, 8 (a workgroup = 4 waves, one per SIMD; 160 KiB / w of LDS each), the grid fills every CU
// three times over.  Reported: shader cycles per wave-instruction as one wave sees them, and -- with no assumption
// about co-residency -- SIMD-cycles per wave-instruction over the whole kernel = kernel time x clock x 1024 SIMDs /
// all wave-instructions executed (the reciprocal of the chip's sustained issue rate per SIMD).
//   build: hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip       run: ./issue_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum { OP_FMA = 0, OP_CVT_PK, OP_PK_ADD_F16, OP_PK_MAX_F16, OP_MUL_LO_U32, OP_EXP, OP_PK_MUL_F32, OP_MFMA16, OP_MFMA16_PLUS_VALU, OP_LDS_READ_B128,
       OP_FMA_MIX, OP_XOR, OP_PK_MAX_I16, OP_MAX_I32, OP_CVT_F16, OP_CVT_PKRTZ, OP_PERM, OP_MED3, OP_MAX_F32, OP_BITOP3, OP_ADD_U32, OP_AND, OP_CNDMASK, OP_LSHL, OP_CVT_I32, OP_FRACT, OP_LSHL_ADD, OP_ADD3, OP_MUL_F32, OP_SUB_F32, OP_MOV, OP_CVT_F32_F16, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_cvt_pk_f16_f32", "v_pk_add_f16", "v_pk_max_f16", "v_mul_lo_u32", "v_exp_f32",
                                    "v_pk_mul_f32", "v_mfma_f32_16x16x32_f16", "mfma + 4 v_cvt_pk_f16_f32 (per 5 instr)", "ds_read_b128",
                                    "v_fma_mix_f32", "v_xor_b32", "v_pk_max_i16", "v_max_i32", "v_cvt_f16_f32", "v_cvt_pkrtz_f16_f32", "v_perm_b32",
                                    "v_med3_f32", "v_max_f32", "v_bitop3_b32", "v_add_u32", "v_and_b32", "v_cndmask_b32", "v_lshlrev_b32", "v_cvt_i32_f32", "v_fract_f32", "v_lshl_add_u32", "v_add3_u32", "v_mul_f32", "v_sub_f32", "v_mov_b32", "v_cvt_f32_f16"};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define R8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  extern __shared__ unsigned char lds[];
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 1e-7f;
  half8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(0.001f * (threadIdx.x + j)); hb[j] = (_Float16)(0.002f * j); }
  float4v m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
  unsigned addr = (threadIdx.x & 63u) * 16u;
  lds[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (OP == OP_FMA) {
      asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_CVT_PK) {
      asm volatile(R8("v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n"
                      "v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_PK_ADD_F16) {
      asm volatile(R8("v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n"
                      "v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_PK_MAX_F16) {
      asm volatile(R8("v_pk_max_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n"
                      "v_pk_max_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_PK_MAX_I16) {
      asm volatile(R8("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n"
                      "v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_MAX_I32) {
      asm volatile(R8("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n"
                      "v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_CVT_F16) {
      asm volatile(R8("v_cvt_f16_f32 %0, %0\n v_cvt_f16_f32 %1, %1\n v_cvt_f16_f32 %2, %2\n v_cvt_f16_f32 %3, %3\n"
                      "v_cvt_f16_f32 %4, %4\n v_cvt_f16_f32 %5, %5\n v_cvt_f16_f32 %6, %6\n v_cvt_f16_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_CVT_PKRTZ) {
      asm volatile(R8("v_cvt_pkrtz_f16_f32 %0, %0, %8\n v_cvt_pkrtz_f16_f32 %1, %1, %8\n v_cvt_pkrtz_f16_f32 %2, %2, %8\n v_cvt_pkrtz_f16_f32 %3, %3, %8\n"
                      "v_cvt_pkrtz_f16_f32 %4, %4, %8\n v_cvt_pkrtz_f16_f32 %5, %5, %8\n v_cvt_pkrtz_f16_f32 %6, %6, %8\n v_cvt_pkrtz_f16_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_PERM) {
      asm volatile(R8("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                      "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_MED3) {
      asm volatile(R8("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                      "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_MAX_F32) {
      asm volatile(R8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                      "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_BITOP3) {
      asm volatile(R8("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
                      "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_ADD_U32) {
      asm volatile(R8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                      "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_AND) {
      asm volatile(R8("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                      "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_CNDMASK) {
      asm volatile(R8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                      "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_LSHL) {
      asm volatile(R8("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n"
                      "v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_CVT_I32) {
      asm volatile(R8("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                      "v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_FRACT) {
      asm volatile(R8("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n"
                      "v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_LSHL_ADD) {
      asm volatile(R8("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n"
                      "v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_ADD3) {
      asm volatile(R8("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n"
                      "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_MUL_F32) {
      asm volatile(R8("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                      "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_SUB_F32) {
      asm volatile(R8("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                      "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_MOV) {
      asm volatile(R8("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3\n"
                      "v_mov_b32 %4, %4\n v_mov_b32 %5, %5\n v_mov_b32 %6, %6\n v_mov_b32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_CVT_F32_F16) {
      asm volatile(R8("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n"
                      "v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (OP == OP_MUL_LO_U32) {
      asm volatile(R8("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                      "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_XOR) {
      asm volatile(R8("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                      "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_EXP) {
      asm volatile(R8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                      "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_FMA_MIX) {
      asm volatile(R8("v_fma_mix_f32 %0, %0, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %1, %8, %9 op_sel_hi:[0,1,0]\n"
                      "v_fma_mix_f32 %2, %2, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %3, %3, %8, %9 op_sel_hi:[0,1,0]\n"
                      "v_fma_mix_f32 %4, %4, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %5, %5, %8, %9 op_sel_hi:[0,1,0]\n"
                      "v_fma_mix_f32 %6, %6, %8, %9 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %7, %7, %8, %9 op_sel_hi:[0,1,0]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (OP == OP_PK_MUL_F32) {
      // 64-bit operand pairs: four accumulators of two floats each
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b, b};
      asm volatile(R8("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                      "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (OP == OP_MFMA16) {
      // 64 MFMAs on four independent accumulators
      for (int j = 0; j < 16; ++j) {
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m1, 0, 0, 0);
        m2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m2, 0, 0, 0);
        m3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m3, 0, 0, 0);
      }
    } else if (OP == OP_MFMA16_PLUS_VALU) {
      // the MLP's mix: one MFMA and four independent VALU conversions, 64 MFMAs + 256 VALU per trip
      for (int j = 0; j < 16; ++j) {
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m0, 0, 0, 0);
        asm volatile("v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m1, 0, 0, 0);
        asm volatile("v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                     : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        m2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m2, 0, 0, 0);
        asm volatile("v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        m3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, m3, 0, 0, 0);
        asm volatile("v_cvt_pk_f16_f32 %0, %0, %4\n v_cvt_pk_f16_f32 %1, %1, %4\n v_cvt_pk_f16_f32 %2, %2, %4\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                     : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      }
    } else if (OP == OP_LDS_READ_B128) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v0, v1, v2, v3;
      asm volatile(R8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                      "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr));
      a0 += v0.x + v1.x + v2.x + v3.x;
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  const float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3];
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63u) == 0) {
    out[2 * wave] = t1 - t0;
    out[2 * wave + 1] = (r1 - r0) | ((unsigned long long)(sink == 12345.0f) << 63);
  }
}

template <int OP>
void run(int w, int iters, unsigned long long* d_out, std::vector<unsigned long long>& h) {
  const int lds = w == 1 ? 160 * 1024 : (160 * 1024) / w - 256;  // exactly w workgroups (4 waves each: one per SIMD) per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int blocks = 256 * w * 3;  // three full rounds
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), lds, 0, d_out, 8);  // warm-up (code object, clocks)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), lds, 0, d_out, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), d_out, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  const int n = blocks * 4;
  for (int i = 0; i < n; ++i) { cyc += (double)h[2 * i]; real += (double)(h[2 * i + 1] & ~(1ull << 63)); }
  cyc /= n; real /= n;
  const double instr = (OP == OP_MFMA16_PLUS_VALU ? 320.0 : 64.0) * iters;
  // chip level, no assumption about which waves were co-resident: kernel time x shader clock x 1024 SIMDs / all wave-instructions
  const double clk_hz = real > 0 ? cyc / real * 1e8 : 2.4e9;
  const double agg = (double)ms * 1e-3 * clk_hz * 1024.0 / (instr * n);
  std::printf("  %-42s w=%d  %7.2f cycles / wave-instr seen by a wave   %6.2f SIMD-cycles / wave-instr over the whole kernel (%.3f ms, %.2f GHz)\n",
              kNames[OP], w, cyc / instr, agg, ms, clk_hz * 1e-9);
}

int main() {
  unsigned long long* d_out;
  CK(hipMalloc(&d_out, 256 * 8 * 3 * 4 * 16));
  std::vector<unsigned long long> h(256 * 8 * 3 * 4 * 2);
  const int iters = 2000;
  for (int w : {1, 2, 4, 8}) {
    std::printf("waves per SIMD: %d\n", w);
    run<OP_FMA>(w, iters, d_out, h);
    run<OP_XOR>(w, iters, d_out, h);
    run<OP_FMA_MIX>(w, iters, d_out, h);
    run<OP_CVT_PK>(w, iters, d_out, h);
    run<OP_PK_ADD_F16>(w, iters, d_out, h);
    run<OP_PK_MAX_F16>(w, iters, d_out, h);
    run<OP_PK_MAX_I16>(w, iters, d_out, h);
    run<OP_MAX_I32>(w, iters, d_out, h);
    run<OP_MED3>(w, iters, d_out, h);
    run<OP_CVT_F16>(w, iters, d_out, h);
    run<OP_CVT_PKRTZ>(w, iters, d_out, h);
    run<OP_PERM>(w, iters, d_out, h);
    run<OP_MAX_F32>(w, iters, d_out, h);
    run<OP_BITOP3>(w, iters, d_out, h);
    run<OP_ADD_U32>(w, iters, d_out, h);
    run<OP_AND>(w, iters, d_out, h);
    run<OP_CNDMASK>(w, iters, d_out, h);
    run<OP_LSHL>(w, iters, d_out, h);
    run<OP_CVT_I32>(w, iters, d_out, h);
    run<OP_FRACT>(w, iters, d_out, h);
    run<OP_LSHL_ADD>(w, iters, d_out, h);
    run<OP_ADD3>(w, iters, d_out, h);
    run<OP_MUL_F32>(w, iters, d_out, h);
    run<OP_SUB_F32>(w, iters, d_out, h);
    run<OP_MOV>(w, iters, d_out, h);
    run<OP_CVT_F32_F16>(w, iters, d_out, h);
    run<OP_MUL_LO_U32>(w, iters, d_out, h);
    run<OP_EXP>(w, iters, d_out, h);
    run<OP_PK_MUL_F32>(w, iters, d_out, h);
    run<OP_MFMA16>(w, iters, d_out, h);
    run<OP_MFMA16_PLUS_VALU>(w, iters, d_out, h);
    run<OP_LDS_READ_B128>(w, iters / 4, d_out, h);
  }
  return 0;
}
