// nrf_renderbuffer.hip -- the presentation chain behind the reference's CudaRenderBuffer on gfx950.
//
// What the reference does with a rendered frame (R/src/render_buffer.cu): fold it into a running mean over samples per
// pixel (accumulate, :224-259, :590-609), then develop that mean for display -- background blend, exposure, film curve,
// transfer function (tonemap, :261-342, :529-556, :611-627) -- two launches that each read and write whole planes
// (32 B read + 16 B written per pixel, twice).  Here the chain is ONE streaming pass, `present_kernel`: per pixel it reads
// the frame (and the mean so far, unless this is the first sample: nothing is cleared, nothing is read), writes the new
// mean, develops it in registers and writes the surface -- plus, on request, the surface as packed 8-bit RGBA, which is what
// a display or an encoder takes.  64 B per pixel (48 for the first sample) instead of 96 (+ a 16 B clear).  The two halves
// stay available on their own (nrf_rb_accumulate / nrf_rb_tonemap: the reference's call shape) as instances of the same
// kernel, so the fused pass is bit-identical to the two calls by construction.
// Everything that does not depend on the pixel is settled on the host once per call: 2^exposure, the background in the
// render colour space, the coefficients of the film curve (`FilmCurve`: identity, Reinhard, or a ratio of two quadratics --
// ACES and Hable differ only in six numbers).  A pure stream over 16-byte pixels: one lane per pixel, grid-stride.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nerfhip.h"

namespace {

thread_local std::string g_rb_err;
extern "C" const char* nrf_last_error(void);

// the sRGB transfer pair as the reference states it (R/include/nerf-cuda/common_device.cuh:38-60: note the 0.41666 exponent).
// On the device the power is 2^(e log2 x) on the two transcendental instructions (v_log_f32, v_exp_f32; the argument is
// positive on this branch): within 2e-6 of libm's powf over the range a colour takes -- the general powf made the chain
// ALU-bound (32-52 us per 1080p plane against 20 us for the same stream without it, profiles/r05/rb_bench.txt).  The host
// forms (the background colour, once per call) stay libm's, which is the oracle's.
__host__ __device__ inline float pow_positive(float x, float e) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));
#else
  return powf(x, e);
#endif
}
__host__ __device__ inline float decode_srgb(float v) { return v <= 0.04045f ? v / 12.92f : pow_positive((v + 0.055f) / 1.055f, 2.4f); }
__host__ __device__ inline float encode_srgb(float v) { return v < 0.0031308f ? 12.92f * v : 1.055f * pow_positive(v, 0.41666f) - 0.055f; }

// A film curve, reduced on the host to what the pixel loop needs.  RATIONAL: y = (x^2 n2 + n1 x + n0) / (d2 x^2 + d1 x + d0)
// on max(x, 0) per channel (operation order as written: it is part of the contract with the oracle); LUMA: Reinhard's
// x / (1 + Y) on max(x, 0) with Y the Rec. 709 luminance.
enum : int { CURVE_NONE = 0, CURVE_RATIONAL = 1, CURVE_LUMA = 2 };
struct FilmCurve {
  int kind;
  float n2, n1, n0, d2, d1, d0;
};
FilmCurve film_curve(int tonemap) {  // render_buffer.cu:261-318, the constants of its two rational curves
  FilmCurve f{CURVE_NONE, 0, 0, 0, 0, 0, 0};
  if (tonemap == NRF_TM_ACES) {  // Narkowicz's fit, pre-scaled by 0.6
    f = {CURVE_RATIONAL, 0.6f * 0.6f * 2.51f, 0.6f * 0.03f, 0.0f, 0.6f * 0.6f * 2.43f, 0.6f * 0.59f, 0.14f};
  } else if (tonemap == NRF_TM_HABLE) {  // Hable's filmic operator, normalised so that the white point 11.2 maps to 1
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, white = 11.2f;
    float n2 = A * F - A * E, n1 = C * B * F - B * E, n0 = 0.0f;
    float d2 = A * F, d1 = B * F;
    const float d0 = D * F * F;
    const float at_white_n = n2 * (white * white) + n1 * white + n0;
    const float at_white_d = d2 * (white * white) + d1 * white + d0;
    const float norm = at_white_d / at_white_n;
    f = {CURVE_RATIONAL, 4.0f * n2 * norm, 2.0f * n1 * norm, n0 * norm, 4.0f * d2, 2.0f * d1, d0};
  } else if (tonemap == NRF_TM_REINHARD) {
    f.kind = CURVE_LUMA;
  }
  return f;
}

// one call's pixel-independent state
struct PresentArgs {
  int n;
  float samples_so_far;  // the mean in the accumulate plane is over this many frames (0: the plane's content is ignored)
  float gain;            // 2^exposure
  float4 backdrop;       // the background colour in the RENDER colour space, alpha untouched
  int render_space, display_space;  // NRF_CS_*
  int clamp_display;
  FilmCurve film;
};

// fold one frame sample into the mean over `k` earlier ones: (mean * k + x) / (k + 1)
__device__ __forceinline__ float fold1(float mean, float x, float k) { return (mean * k + x) / (k + 1); }
__device__ __forceinline__ float4 fold_sample(float4 mean, float4 x, const PresentArgs& A) {
  const float k = A.samples_so_far;
  if (A.render_space == NRF_CS_VISPOSNEG) {  // a signed quantity kept as (positive part, negative part) in r, g; b is left alone
    const float signed_mean = fold1(mean.x - mean.y, x.x - x.y, k);
    mean.x = fmaxf(signed_mean, 0.0f);
    mean.y = fmaxf(-signed_mean, 0.0f);
  } else {
    if (A.render_space == NRF_CS_SRGB) {  // the mean is kept in display-referred values
      x.x = encode_srgb(x.x);
      x.y = encode_srgb(x.y);
      x.z = encode_srgb(x.z);
    }
    mean.x = fold1(mean.x, x.x, k);
    mean.y = fold1(mean.y, x.y, k);
    mean.z = fold1(mean.z, x.z, k);
  }
  mean.w = fold1(mean.w, x.w, k);
  return mean;
}

// the mean as it goes to the display: over the backdrop, exposed, through the film curve, in the display's transfer function
__device__ __forceinline__ float4 develop(float4 mean, const PresentArgs& A) {
  const float uncovered = (1 - mean.w) * A.backdrop.w;
  float c[3] = {mean.x + A.backdrop.x * uncovered, mean.y + A.backdrop.y * uncovered, mean.z + A.backdrop.z * uncovered};
  const float alpha = mean.w + uncovered;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (A.render_space == NRF_CS_SRGB) c[k] = decode_srgb(c[k]);
    c[k] *= A.gain;
  }
  if (A.film.kind != CURVE_NONE) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = fmaxf(c[k], 0.f);
    if (A.film.kind == CURVE_LUMA) {
      const float scale = 1.f / ((0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]) + 1.0f);
#pragma unroll
      for (int k = 0; k < 3; ++k) c[k] = c[k] * scale;
    } else {
      const FilmCurve& f = A.film;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float sq = c[k] * c[k];
        c[k] = (sq * f.n2 + f.n1 * c[k] + f.n0) / (f.d2 * sq + f.d1 * c[k] + f.d0);
      }
    }
  }
  float4 out = make_float4(c[0], c[1], c[2], alpha);
  if (A.display_space == NRF_CS_SRGB) {
    out.x = encode_srgb(out.x);
    out.y = encode_srgb(out.y);
    out.z = encode_srgb(out.z);
  }
  if (A.clamp_display) {
    out.x = fminf(fmaxf(out.x, 0.f), 1.f);
    out.y = fminf(fmaxf(out.y, 0.f), 1.f);
    out.z = fminf(fmaxf(out.z, 0.f), 1.f);
    out.w = fminf(fmaxf(out.w, 0.f), 1.f);
  }
  return out;
}

// the library's 8-bit rule (nrf_render.h quant_u8, the reference's (unsigned char)(255.0 * x) made safe: saturating, NaN -> 0)
__device__ __forceinline__ uint32_t to_u8(float v) {
  const double s = 255.0 * (double)v;
  if (!(s > 0.0)) return 0u;
  if (s >= 255.0) return 255u;
  return (uint32_t)s;
}

// FOLD: frame -> mean; DEVELOP: mean -> surface (+ PACK8: the surface as r | g << 8 | b << 16 | a << 24); both: one pass
enum : int { PASS_FOLD = 1, PASS_DEVELOP = 2 };
template <int PASSES, bool PACK8>
__global__ __launch_bounds__(256) void present_kernel(const PresentArgs A, const float4* __restrict__ frame, float4* __restrict__ mean_plane,
                                                      float4* __restrict__ surface, uint32_t* __restrict__ rgba8) {
  const bool first = A.samples_so_far == 0.0f;  // wave-uniform: the mean plane is write-only for the first sample
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
    float4 mean;
    if constexpr ((PASSES & PASS_FOLD) != 0) {
      const float4 x = frame[i];
      mean = fold_sample(first ? make_float4(0.f, 0.f, 0.f, 0.f) : mean_plane[i], x, A);
      mean_plane[i] = mean;
    } else {
      mean = mean_plane[i];
    }
    if constexpr ((PASSES & PASS_DEVELOP) != 0) {
      const float4 out = develop(mean, A);
      surface[i] = out;
      if constexpr (PACK8) rgba8[i] = to_u8(out.x) | (to_u8(out.y) << 8) | (to_u8(out.z) << 16) | (to_u8(out.w) << 24);
    }
  }
}

}  // namespace

struct nrf_render_buffer {
  int device = 0, W = 0, H = 0;
  uint32_t spp = 0;
  int color_space = NRF_CS_LINEAR, curve = NRF_TM_IDENTITY;
  void *frame = nullptr, *depth = nullptr, *accum = nullptr, *surface = nullptr;
  hipStream_t stream = nullptr;
};

// error plumbing shared with nrf_api.hip through nrf_last_error(): this TU keeps its own message
// and exposes it via a tiny hook
extern "C" void nrf_set_last_error_(const char* msg);
namespace {
int rb_fail(int code, const std::string& m) { nrf_set_last_error_(m.c_str()); return code; }
int rb_hip(hipError_t e, const char* what) { return rb_fail(NRF_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define RB_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return rb_hip(_e, #x); } while (0)
void rb_free(nrf_render_buffer* rb) {
  for (void** p : {&rb->frame, &rb->depth, &rb->accum, &rb->surface}) { if (*p) (void)hipFree(*p); *p = nullptr; }
}
int rb_grid(int n) { int g = (n + 255) / 256; return g < 1 ? 1 : (g > 2048 ? 2048 : g); }
}  // namespace

extern "C" {

int nrf_rb_create(int device, nrf_render_buffer** out) {
  if (!out) return rb_fail(NRF_E_INVALID, "null argument");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return rb_fail(NRF_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
  nrf_render_buffer* rb = new nrf_render_buffer;
  rb->device = device;
  RB_TRY(hipSetDevice(device));
  RB_TRY(hipStreamCreateWithFlags(&rb->stream, hipStreamNonBlocking));
  *out = rb;
  return NRF_OK;
}

int nrf_rb_destroy(nrf_render_buffer* rb) {
  if (!rb) return NRF_OK;
  (void)hipSetDevice(rb->device);
  (void)hipDeviceSynchronize();
  rb_free(rb);
  if (rb->stream) (void)hipStreamDestroy(rb->stream);
  delete rb;
  return NRF_OK;
}

int nrf_rb_resize(nrf_render_buffer* rb, int width, int height) {
  if (!rb || width <= 0 || height <= 0) return rb_fail(NRF_E_INVALID, "bad resolution");
  RB_TRY(hipSetDevice(rb->device));
  RB_TRY(hipDeviceSynchronize());
  rb_free(rb);
  const size_t n = (size_t)width * height;
  RB_TRY(hipMalloc(&rb->frame, n * 16));
  RB_TRY(hipMalloc(&rb->depth, n * 4));
  RB_TRY(hipMalloc(&rb->accum, n * 16));
  RB_TRY(hipMalloc(&rb->surface, n * 16));
  RB_TRY(hipMemsetAsync(rb->frame, 0, n * 16, rb->stream));
  RB_TRY(hipMemsetAsync(rb->depth, 0, n * 4, rb->stream));
  RB_TRY(hipMemsetAsync(rb->accum, 0, n * 16, rb->stream));
  RB_TRY(hipMemsetAsync(rb->surface, 0, n * 16, rb->stream));
  RB_TRY(hipStreamSynchronize(rb->stream));
  rb->W = width;
  rb->H = height;
  rb->spp = 0;
  return NRF_OK;
}

int nrf_rb_reset_accumulation(nrf_render_buffer* rb) {
  if (!rb) return rb_fail(NRF_E_INVALID, "null argument");
  rb->spp = 0;
  return NRF_OK;
}
int nrf_rb_spp(nrf_render_buffer* rb, uint32_t* spp) {
  if (!rb || !spp) return rb_fail(NRF_E_INVALID, "null argument");
  *spp = rb->spp;
  return NRF_OK;
}
int nrf_rb_set_color_space(nrf_render_buffer* rb, int cs) {
  if (!rb || cs < 0 || cs > 2) return rb_fail(NRF_E_INVALID, "bad color space");
  rb->color_space = cs;
  return NRF_OK;
}
int nrf_rb_set_tonemap_curve(nrf_render_buffer* rb, int curve) {
  if (!rb || curve < 0 || curve > 3) return rb_fail(NRF_E_INVALID, "bad tonemap curve");
  rb->curve = curve;
  return NRF_OK;
}
int nrf_rb_buffers(nrf_render_buffer* rb, void** frame, void** depth, void** accumulate, void** surface) {
  if (!rb) return rb_fail(NRF_E_INVALID, "null argument");
  if (!rb->frame) return rb_fail(NRF_E_STATE, "resize has not been called");
  if (frame) *frame = rb->frame;
  if (depth) *depth = rb->depth;
  if (accumulate) *accumulate = rb->accum;
  if (surface) *surface = rb->surface;
  return NRF_OK;
}

#define RB_READY()                                                             \
  if (!rb) return rb_fail(NRF_E_INVALID, "null argument");                     \
  if (!rb->frame) return rb_fail(NRF_E_STATE, "resize has not been called");   \
  RB_TRY(hipSetDevice(rb->device));                                            \
  hipStream_t st = stream ? (hipStream_t)stream : rb->stream;                  \
  const int n = rb->W * rb->H;

int nrf_rb_clear_frame(nrf_render_buffer* rb, void* stream) {
  RB_READY();
  RB_TRY(hipMemsetAsync(rb->frame, 0, (size_t)n * 16, st));
  RB_TRY(hipMemsetAsync(rb->depth, 0, (size_t)n * 4, st));
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

}  // extern "C"
namespace {
// the whole launch of one call: grid sized for the plane, every pixel-independent value fixed here
template <int PASSES>
int launch_present(nrf_render_buffer* rb, hipStream_t st, float exposure, const float bg[4], int display_space, void* rgba8) {
  PresentArgs A{};
  A.n = rb->W * rb->H;
  A.samples_so_far = (float)rb->spp;
  A.gain = powf(2.0f, exposure);
  A.render_space = rb->color_space;
  A.display_space = display_space;
  A.clamp_display = 0;
  A.film = film_curve(rb->curve);
  A.backdrop = bg ? make_float4(bg[0], bg[1], bg[2], bg[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (rb->color_space != NRF_CS_SRGB) {  // backgrounds are given display-referred: into the render space once, on the host
    A.backdrop.x = decode_srgb(A.backdrop.x);
    A.backdrop.y = decode_srgb(A.backdrop.y);
    A.backdrop.z = decode_srgb(A.backdrop.z);
  }
  const dim3 grid(rb_grid(A.n)), block(256);
  if (rgba8) hipLaunchKernelGGL((present_kernel<PASSES, true>), grid, block, 0, st, A, (const float4*)rb->frame, (float4*)rb->accum,
                                (float4*)rb->surface, (uint32_t*)rgba8);
  else hipLaunchKernelGGL((present_kernel<PASSES, false>), grid, block, 0, st, A, (const float4*)rb->frame, (float4*)rb->accum,
                          (float4*)rb->surface, (uint32_t*)nullptr);
  RB_TRY(hipGetLastError());
  return NRF_OK;
}
}  // namespace
extern "C" {

int nrf_rb_accumulate(nrf_render_buffer* rb, float exposure, void* stream) {
  (void)exposure;  // unused in the reference as well (render_buffer.cu:595)
  RB_READY();
  (void)n;
  const int rc = launch_present<PASS_FOLD>(rb, st, 0.0f, nullptr, NRF_CS_LINEAR, nullptr);
  if (rc != NRF_OK) return rc;
  ++rb->spp;
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_rb_tonemap(nrf_render_buffer* rb, float exposure, const float bg[4], int output_color_space, void* stream) {
  if (!bg || output_color_space < 0 || output_color_space > 2) return rb_fail(NRF_E_INVALID, "bad argument");
  RB_READY();
  (void)n;
  const int rc = launch_present<PASS_DEVELOP>(rb, st, exposure, bg, output_color_space, nullptr);
  if (rc != NRF_OK) return rc;
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_rb_present(nrf_render_buffer* rb, float exposure, const float bg[4], int output_color_space, void* rgba8, void* stream) {
  if (!bg || output_color_space < 0 || output_color_space > 2) return rb_fail(NRF_E_INVALID, "bad argument");
  RB_READY();
  (void)n;
  const int rc = launch_present<PASS_FOLD | PASS_DEVELOP>(rb, st, exposure, bg, output_color_space, rgba8);
  if (rc != NRF_OK) return rc;
  ++rb->spp;
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

// overlay_depth_kernel, R/src/render_buffer.cu:431-477 with colormap_turbo (:413-429): the polynomial is evaluated
// as the fixed-size Eigen dot products do, ((a0*b0 + a1*b1) + a2*b2) + a3*b3, every operation rounded on its own.
__device__ __forceinline__ float dot4(const float a[4], float b0, float b1, float b2, float b3) {
  return ((a[0] * b0 + a[1] * b1) + a[2] * b2) + a[3] * b3;
}
__device__ __forceinline__ void colormap_turbo(float x, float c[3]) {
  const float kR4[4] = {0.13572138f, 4.61539260f, -42.66032258f, 132.13108234f};
  const float kG4[4] = {0.09140261f, 2.19418839f, 4.84296658f, -14.18503333f};
  const float kB4[4] = {0.10667330f, 12.64194608f, -60.58204836f, 110.36276771f};
  const float kR2[2] = {-152.94239396f, 59.28637943f};
  const float kG2[2] = {4.27729857f, 2.82956604f};
  const float kB2[2] = {-89.90310912f, 27.34824973f};
  x = fminf(fmaxf(x, 0.0f), 1.0f);  // __saturatef (NaN -> 0)
  if (!(x == x)) x = 0.0f;
  const float x2 = x * x, x3 = x2 * x;
  const float v20 = x3 * x, v21 = x3 * x2;
  c[0] = dot4(kR4, 1.0f, x, x2, x3) + (v20 * kR2[0] + v21 * kR2[1]);
  c[1] = dot4(kG4, 1.0f, x, x2, x3) + (v20 * kG2[0] + v21 * kG2[1]);
  c[2] = dot4(kB4, 1.0f, x, x2, x3) + (v20 * kB2[0] + v21 * kB2[1]);
}

__global__ __launch_bounds__(256) void overlay_depth_kernel(int W, int H, float alpha, const float* __restrict__ depth,
                                                            float depth_scale, int img_w, int img_h, int fov_axis, float zoom,
                                                            float center_x, float center_y, float4* __restrict__ surface) {
  const int n = W * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = i % W, y = i / W;
    const float scale = (float)(fov_axis == 0 ? img_w : img_h) / (float)(fov_axis == 0 ? W : H);
    float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
    fx -= (float)W * 0.5f; fx /= zoom; fx += center_x * (float)W;
    fy -= (float)H * 0.5f; fy /= zoom; fy += center_y * (float)H;
    const float u = (fx - (float)W * 0.5f) * scale + (float)img_w * 0.5f;
    const float v = (fy - (float)H * 0.5f) * scale + (float)img_h * 0.5f;
    const int srcx = (int)floorf(u), srcy = (int)floorf(v);
    float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(srcx >= img_w || srcy >= img_h || srcx < 0 || srcy < 0)) {
      float c[3];
      colormap_turbo(depth[(size_t)srcx + (size_t)img_w * srcy] * depth_scale, c);
      color = make_float4(c[0], c[1], c[2], 1.0f);
    }
    const float4 prev = surface[i];
    const float ia = 1.f - alpha;
    surface[i] = make_float4(color.x * alpha + prev.x * ia, color.y * alpha + prev.y * ia, color.z * alpha + prev.z * ia,
                             color.w * alpha + prev.w * ia);
  }
}

int nrf_rb_overlay_depth(nrf_render_buffer* rb, float alpha, const void* depth, float depth_scale, int image_width,
                         int image_height, int fov_axis, float zoom, const float screen_center[2], void* stream) {
  if (!depth || !screen_center || image_width <= 0 || image_height <= 0 || (fov_axis != 0 && fov_axis != 1))
    return rb_fail(NRF_E_INVALID, "bad argument");
  RB_READY();
  hipLaunchKernelGGL(overlay_depth_kernel, dim3(rb_grid(n)), dim3(256), 0, st, rb->W, rb->H, alpha, (const float*)depth, depth_scale,
                     image_width, image_height, fov_axis, zoom, screen_center[0], screen_center[1], (float4*)rb->surface);
  RB_TRY(hipGetLastError());
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_rb_host_to_accumulate_buffer(nrf_render_buffer* rb, const uint8_t* rgb, int count) {
  void* stream = nullptr;
  RB_READY();
  if (!rgb || count != n) return rb_fail(NRF_E_INVALID, "size does not match the resolution");
  std::vector<float> v((size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < 3; ++j) v[(size_t)i * 4 + j] = float(rgb[(size_t)i * 3 + j]) / 255.0;
    v[(size_t)i * 4 + 3] = 1.0;
  }
  RB_TRY(hipMemcpyAsync(rb->accum, v.data(), v.size() * 4, hipMemcpyHostToDevice, st));
  RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_rb_read(nrf_render_buffer* rb, float* accumulate_rgba, float* surface_rgba) {
  void* stream = nullptr;
  RB_READY();
  RB_TRY(hipDeviceSynchronize());
  if (accumulate_rgba) RB_TRY(hipMemcpy(accumulate_rgba, rb->accum, (size_t)n * 16, hipMemcpyDeviceToHost));
  if (surface_rgba) RB_TRY(hipMemcpy(surface_rgba, rb->surface, (size_t)n * 16, hipMemcpyDeviceToHost));
  (void)st;
  return NRF_OK;
}

}  // extern "C"
