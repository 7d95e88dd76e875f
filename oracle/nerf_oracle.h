/*
 * nerf_oracle.h -- CPU restatement of the reference render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (nerf-cuda_amd/,
 * include/) links, loads or calls this library.  It is used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * reported CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or weights for
 * this path and cannot be built here (CUDA-only), so
 * this oracle is pinned only by the source-derived known answers of
 * SURVEY.md Appendix C (tests/test_oracle_kat.py) -- plus, where a published
 * definition exists that does not pass through this repository's reading of
 * the source, by that: scipy's spherical harmonics for the SH table, the
 * published film curves, numpy's and the CPU's fp16 conversions.
 *
 * Arithmetic contract (what "the reference's algorithm" means here): every
 * fp32 operation individually rounded (no FMA contraction), fp16 storage with
 * round-to-nearest-even, hash-grid interpolation accumulated in fp16
 * (grid.h:236,260), MLP products accumulated in fp32 in ascending-k order and
 * rounded to fp16 after the activation of every layer (the reference
 * accumulates in fp16 inside tensor cores, which is not specified bit-wise).
 */
#ifndef NERF_ORACLE_H_
#define NERF_ORACLE_H_

#include "../include/nerfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nrfo_model nrfo_model;

/* schedule of the march/eval/composite loop */
enum {
  NRFO_SCHED_REFERENCE = 0, /* nerf_render.cu:269-338: one global alive list,
                               n_step = clamp(N/num_alive,1,8)               */
  NRFO_SCHED_TILE64 = 1,    /* independent 8x8 pixel tiles with
                               n_step = clamp(64/alive,1,8)                  */
  NRFO_SCHED_PER_RAY = 2    /* the reference loop with n_step == 1 for every
                               iteration (what it does whenever more than
                               half of the rays are alive): each ray is
                               marched, evaluated and composited one sample
                               at a time.  This is the per-ray semantics the
                               HIP kernel implements; batching cannot change it */
};

const char* nrfo_last_error(void);
int nrfo_create(const nrf_model_desc* d, nrfo_model** out);
void nrfo_destroy(nrfo_model* m);
/* padded widths of the position / direction encodings (fp16 values per sample) */
void nrfo_widths(const nrfo_model* m, uint32_t* feat_width, uint32_t* dir_width);

/* Accumulator arithmetic of the two MLPs.  NRFO_ACC_FP32 (default) is the contract shared with the HIP path.  The
 * FP16 modes emulate the reference's own accumulators -- `wmma::fragment<accumulator,16,16,16,__half>` in every layer,
 * T/src/fully_fused_mlp.cu:69,334,437 -- to MEASURE how far the fp32-accumulate neighbour is from it: the running sum is
 * rounded to fp16 (RNE) after every block of n products (n = 16: one mma_sync per K block, the reference's granularity;
 * n = 1: after every product, the pessimistic bound; 4 / 8: older HMMA K granularities), and the activation is applied
 * to that fp16 value (warp_activation<__half>).  Tensor-core rounding inside a block is unspecified: summed in fp32. */
enum { NRFO_ACC_FP32 = 0, NRFO_ACC_FP16_STEP = 1, NRFO_ACC_FP16_K4 = 4, NRFO_ACC_FP16_K8 = 8, NRFO_ACC_FP16_K16 = 16 };
int nrfo_set_mlp_accumulate(nrfo_model* m, int mode);

/* FMA contraction.  0 (default): every fp32 operation individually rounded -- the contract shared with the HIP path.
 * 1: `a * b + c` evaluated as ONE fused multiply-add wherever the reference's device source has it in one expression
 * (nvcc's default -fmad=true; R/CMakeLists.txt:71-79 sets no -fmad=false): set_rays_d's norm and rotation
 * (render_utils.h:43-47), `ox + t * dx` (:595-597), `x * mip_rbound + 1` (:609-614), `(..) * mip_bound - x` (:643-645),
 * linear_transformer (common_device.cuh:34), pos_fract `input * scale + 0.5f` (T/.../common_device.h:416) and the level scale
 * (grid.h:189), kernel_sh's polynomials (spherical_harmonics.h:66-152), frequency_encoding (frequency.h:88), the sums of
 * kernel_composite_rays (:712-720) and get_image_and_depth (:258-260).  Not fused: products cast before they are added
 * (grid.h:260), host code (nerf_matrix_to_ngp, the offset table).  Exists to MEASURE the distance between the contract and
 * what the reference binary computes; which of two products of a sum a compiler fuses is its choice (here: the left).
 * on == 2: the sensitivity run -- the OTHER choice at every site that has one: of two products the right one is fused, and a
 * product with other uses (`alpha * T` in kernel_composite_rays) is not fused at all; the spread between modes 1 and 2 is the
 * uncertainty of the emulated distance (tests/test_contract_modes.py, bench.py parity.vs_fma_contract).                    */
int nrfo_set_contract(nrfo_model* m, int on);

/* fp16 helpers (round-to-nearest-even, IEEE binary16).  nrfo_f32_to_f16 / nrfo_f16_to_f32 are what the oracle computes
 * with (F16C instructions when built with -mf16c, see nrfo_fp16_backend); the *_soft forms are the bit-level definition. */
uint16_t nrfo_f32_to_f16(float f);
float nrfo_f16_to_f32(uint16_t h);
uint16_t nrfo_f32_to_f16_soft(float f);
float nrfo_f16_to_f32_soft(uint16_t h);
const char* nrfo_fp16_backend(void); /* "f16c" or "software" */
/* one activation of the MLPs on an fp32 pre-activation (T/.../common_device.h:68-114), as mlp_one applies it -- for the known
 * answers of tcnn's ReLU-as-a-product: negative -> -0, NaN -> NaN, below -65504 (an fp16 -inf) -> NaN                        */
float nrfo_activation(uint32_t act, float v);
/* pcg32(initstate, initseq).next_float() (T/dependencies/pcg32/pcg32.h): the number the march's perturb branch draws per ray and
 * call (render_utils.h:585-589)                                                                                              */
float nrfo_pcg32_first_float(uint64_t initstate, uint64_t initseq);
/* conversions in use vs the software definition: all 2^16 halves, every stride-th of the 2^32 floats; returns mismatches */
uint64_t nrfo_fp16_selfcheck(uint32_t stride);

/* render_utils.h:68-77 */
void nrfo_nerf_matrix_to_ngp(const float pose[16], float scale, float out[16]);
/* grid.h:81-117 (3-D) */
uint32_t nrfo_fast_hash3(uint32_t x, uint32_t y, uint32_t z);
uint32_t nrfo_grid_index(const nrfo_model* m, uint32_t level, uint32_t x, uint32_t y,
                         uint32_t z);

/* stage functions; all pointers are host pointers, layouts as in nerfhip.h  */
int nrfo_encode_grid(const nrfo_model* m, const float* pos01, uint32_t n, uint16_t* out);
int nrfo_encode_dir(const nrfo_model* m, const float* dir01, uint32_t n, uint16_t* out);
int nrfo_mlp_forward(const nrfo_model* m, const uint16_t* feat, const uint16_t* dirfeat,
                     uint32_t n, uint16_t* out4);
int nrfo_network(const nrfo_model* m, const float* xyz, const float* dir, uint32_t n,
                 float* sigma, float* rgb);
int nrfo_generate_rays(const nrfo_model* m, const float cam[4], const float pose[16],
                       int W, int H, const nrf_options* o, float* rays_o, float* rays_d,
                       float* nears, float* fars);
int nrfo_march(const nrfo_model* m, const nrf_options* o, const float* rays_o,
               const float* rays_d, const float* rays_t, const float* fars, uint32_t n,
               uint32_t n_step, float* xyzs, float* dirs, float* deltas);
int nrfo_composite(const float* sigmas, const float* rgbs, const float* deltas, uint32_t n,
                   uint32_t n_step, float* rays_t, float* state);
/* the t at which every trip of kernel_march_rays begins, along one ray of an EMPTY volume (the hop arithmetic of
 * render_utils.h:639-651 alone decides them); returns the trip count, stores the first `cap` starts                   */
uint32_t nrfo_march_trip_starts(float bound, uint32_t cascade, uint32_t H, float dt_gamma, const float o[3],
                                const float d[3], float t, float far, float* starts, uint32_t cap);

/* NerfRender::render_frame.  rgba [H][W][4], depth [H][W] row-major.
 * n_threads <= 0 -> all cores.                                              */
int nrfo_render(const nrfo_model* m, const float cam[4], const float pose[16], int W,
                int H, const nrf_options* o, int schedule, int n_threads, float* rgba,
                float* depth, nrf_stats* stats);
/* nrfo_render + per ray ([H][W]) the number of samples its march emitted and a hash of their (dt, t - last_t) bits */
int nrfo_render_rays(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H, const nrf_options* o,
                     int schedule, int n_threads, float* rgba, float* depth, nrf_stats* stats, uint32_t* ray_samples,
                     uint64_t* ray_hash);
/* NRFO_SCHED_PER_RAY runs every ray to its end on its own (no rounds, dynamic schedule: the timed CPU baseline); this is the
 * same schedule through the global round loop of nerf_render.cu:269-338 with n_step fixed to 1 -- the cross-check.        */
int nrfo_render_per_ray_rounds(const nrfo_model* m, const float cam[4], const float pose[16], int W, int H,
                               const nrf_options* o, float* rgba, float* depth, nrf_stats* stats);
/* NerfRender::generate_density_grid (nerf_render.cu:388-429) as nerfhip.h's nrf_generate_density_grid completes it:
 * grid [cascade * H^3] (x-major cells), mean_density = mean(max(g, 0))                                           */
int nrfo_density_grid(const nrfo_model* m, int n_iterations, float decay, float* grid, float* mean_density);
/* nerf_render.cu:352-359 with saturation (DESIGN.md deviation D-2)          */
void nrfo_quantize_u8(const float* rgba, const float* depth, int n_px, uint8_t* rgb,
                      uint8_t* depth_u8);
int nrfo_max_threads(void);

/* render buffer chain: R/src/render_buffer.cu:224-259 (accumulate_kernel) and :261-342, :529-556
 * (tonemap + tonemap_kernel); color spaces / curves are the NRF_CS_* / NRF_TM_* enums.           */
void nrfo_rb_accumulate(const float* frame_rgba, float* accum_rgba, int n, float sample_count, int color_space);
void nrfo_rb_tonemap(const float* accum_rgba, float* surface_rgba, int n, float exposure, const float bg[4],
                     int color_space, int output_color_space, int curve, int clamp_output);

/* render_buffer.cu:413-477: turbo-coloured depth blended over the surface [H][W][4], in place */
void nrfo_rb_overlay_depth(float* surface_rgba, int W, int H, float alpha, const float* depth, float depth_scale, int img_w,
                           int img_h, int fov_axis, float zoom, float center_x, float center_y);

#ifdef __cplusplus
}
#endif
#endif
