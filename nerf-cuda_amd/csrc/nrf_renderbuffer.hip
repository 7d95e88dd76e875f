// nrf_renderbuffer.hip -- the presentation chain of the reference's CudaRenderBuffer on gfx950:
// accumulate (running mean over spp), tonemap (background blend, exposure, curve, sRGB).
// Reference: R/src/render_buffer.cu:224-259 (accumulate_kernel), :261-342 (tonemap), :529-556
// (tonemap_kernel), :590-627 (host methods); colour helpers R/include/nerf-cuda/common_device.cuh:38-60.
// Both kernels are pure HBM streams (32 B read + 16 B written per pixel); one thread per pixel,
// 16-byte accesses.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nerfhip.h"

namespace {

thread_local std::string g_rb_err;
extern "C" const char* nrf_last_error(void);

__host__ __device__ inline float srgb_to_linear1(float srgb) {
  return srgb <= 0.04045f ? srgb / 12.92f : powf((srgb + 0.055f) / 1.055f, 2.4f);
}
__host__ __device__ inline float linear_to_srgb1(float linear) {
  return linear < 0.0031308f ? 12.92f * linear : 1.055f * powf(linear, 0.41666f) - 0.055f;
}

__device__ inline void tonemap_curve(float c[3], int curve) {  // render_buffer.cu:261-318
  if (curve == NRF_TM_IDENTITY) return;
  for (int i = 0; i < 3; ++i) c[i] = fmaxf(c[i], 0.f);
  float k0, k1, k2, k3, k4, k5;
  if (curve == NRF_TM_ACES) {
    k0 = 0.6f * 0.6f * 2.51f; k1 = 0.6f * 0.03f; k2 = 0.0f;
    k3 = 0.6f * 0.6f * 2.43f; k4 = 0.6f * 0.59f; k5 = 0.14f;
  } else if (curve == NRF_TM_HABLE) {
    const float A = 0.15f, B = 0.50f, Cc = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    k0 = A * F - A * E; k1 = Cc * B * F - B * E; k2 = 0.0f;
    k3 = A * F; k4 = B * F; k5 = D * F * F;
    const float Wt = 11.2f;
    const float nom = k0 * (Wt * Wt) + k1 * Wt + k2;
    const float denom = k3 * (Wt * Wt) + k4 * Wt + k5;
    const float white_scale = denom / nom;
    k0 = 4.0f * k0 * white_scale; k1 = 2.0f * k1 * white_scale; k2 = k2 * white_scale;
    k3 = 4.0f * k3; k4 = 2.0f * k4;
  } else {  // Reinhard
    const float Y = 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2];
    const float s = 1.f / (Y + 1.0f);
    for (int i = 0; i < 3; ++i) c[i] = c[i] * s;
    return;
  }
  for (int i = 0; i < 3; ++i) {
    const float sq = c[i] * c[i];
    c[i] = (sq * k0 + k1 * c[i] + k2) / (k3 * sq + k4 * c[i] + k5);
  }
}

__global__ __launch_bounds__(256) void accumulate_kernel(int n, const float4* __restrict__ frame, float4* __restrict__ accum,
                                                         float sample_count, int color_space) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 color = frame[i];
    float4 tmp = accum[i];
    if (color_space == NRF_CS_VISPOSNEG) {
      const float val = color.x - color.y;
      float tmp_val = tmp.x - tmp.y;
      tmp_val = (tmp_val * sample_count + val) / (sample_count + 1);
      tmp.x = fmaxf(tmp_val, 0.0f);
      tmp.y = fmaxf(-tmp_val, 0.0f);
    } else {
      if (color_space == NRF_CS_SRGB) {
        color.x = linear_to_srgb1(color.x); color.y = linear_to_srgb1(color.y); color.z = linear_to_srgb1(color.z);
      }
      tmp.x = (tmp.x * sample_count + color.x) / (sample_count + 1);
      tmp.y = (tmp.y * sample_count + color.y) / (sample_count + 1);
      tmp.z = (tmp.z * sample_count + color.z) / (sample_count + 1);
    }
    tmp.w = (tmp.w * sample_count + color.w) / (sample_count + 1);
    accum[i] = tmp;
  }
}

__global__ __launch_bounds__(256) void tonemap_kernel(int n, float exposure, float4 bg, const float4* __restrict__ accum,
                                                      int color_space, int output_color_space, int curve,
                                                      bool clamp_output_color, float4* __restrict__ surface) {
  // The background color is represented in SRGB, so convert to linear if that's not the rendering space.
  if (color_space != NRF_CS_SRGB) { bg.x = srgb_to_linear1(bg.x); bg.y = srgb_to_linear1(bg.y); bg.z = srgb_to_linear1(bg.z); }
  const float gain = powf(2.0f, exposure);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 color = accum[i];
    const float weight = (1 - color.w) * bg.w;
    float c[3] = {color.x + bg.x * weight, color.y + bg.y * weight, color.z + bg.z * weight};
    color.w += weight;
    if (color_space == NRF_CS_SRGB) for (int k = 0; k < 3; ++k) c[k] = srgb_to_linear1(c[k]);
    for (int k = 0; k < 3; ++k) c[k] *= gain;
    tonemap_curve(c, curve);
    if (output_color_space == NRF_CS_SRGB) for (int k = 0; k < 3; ++k) c[k] = linear_to_srgb1(c[k]);
    float4 o = make_float4(c[0], c[1], c[2], color.w);
    if (clamp_output_color) {
      o.x = fminf(fmaxf(o.x, 0.f), 1.f); o.y = fminf(fmaxf(o.y, 0.f), 1.f);
      o.z = fminf(fmaxf(o.z, 0.f), 1.f); o.w = fminf(fmaxf(o.w, 0.f), 1.f);
    }
    surface[i] = o;
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

int nrf_rb_accumulate(nrf_render_buffer* rb, float exposure, void* stream) {
  (void)exposure;  // unused in the reference as well (render_buffer.cu:595)
  RB_READY();
  if (rb->spp == 0) RB_TRY(hipMemsetAsync(rb->accum, 0, (size_t)n * 16, st));
  hipLaunchKernelGGL(accumulate_kernel, dim3(rb_grid(n)), dim3(256), 0, st, n, (const float4*)rb->frame, (float4*)rb->accum,
                     (float)rb->spp, rb->color_space);
  RB_TRY(hipGetLastError());
  ++rb->spp;
  if (!stream) RB_TRY(hipStreamSynchronize(st));
  return NRF_OK;
}

int nrf_rb_tonemap(nrf_render_buffer* rb, float exposure, const float bg[4], int output_color_space, void* stream) {
  if (!bg || output_color_space < 0 || output_color_space > 2) return rb_fail(NRF_E_INVALID, "bad argument");
  RB_READY();
  hipLaunchKernelGGL(tonemap_kernel, dim3(rb_grid(n)), dim3(256), 0, st, n, exposure, make_float4(bg[0], bg[1], bg[2], bg[3]),
                     (const float4*)rb->accum, rb->color_space, output_color_space, rb->curve, false, (float4*)rb->surface);
  RB_TRY(hipGetLastError());
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
