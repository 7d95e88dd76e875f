// render_buffer.h -- C++ mirror of the reference's ngp::CudaRenderBuffer presentation API
// (include/nerf-cuda/render_buffer.h:160-315) on the C ABI: same method names for resize,
// reset_accumulation, spp, frame/depth/accumulate buffers, clear_frame, accumulate, tonemap,
// host_to_accumulate_buffer, accumulate_buffer_host, overlay_depth.  GL textures, CUDA surfaces, DLSS and the
// image / false-colour overlays of the reference class are out of scope (NVIDIA/GL presentation, training views).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nerfhip.h"
#include "nerf_render.h"

namespace ngp {

enum class EColorSpace : int { Linear = NRF_CS_LINEAR, SRGB = NRF_CS_SRGB, VisPosNeg = NRF_CS_VISPOSNEG };
enum class ETonemapCurve : int { Identity = NRF_TM_IDENTITY, ACES = NRF_TM_ACES, Hable = NRF_TM_HABLE, Reinhard = NRF_TM_REINHARD };

class RenderBuffer {
 public:
  explicit RenderBuffer(int device = 0) { ok(nrf_rb_create(device, &m_rb)); }
  ~RenderBuffer() { nrf_rb_destroy(m_rb); }
  RenderBuffer(const RenderBuffer&) = delete;
  RenderBuffer& operator=(const RenderBuffer&) = delete;

  void resize(const Vector2i& res) { ok(nrf_rb_resize(m_rb, res[0], res[1])); m_res = res; }
  Vector2i in_resolution() const { return m_res; }
  Vector2i out_resolution() const { return m_res; }
  void reset_accumulation() { ok(nrf_rb_reset_accumulation(m_rb)); }
  uint32_t spp() const { uint32_t v = 0; ok(nrf_rb_spp(m_rb, &v)); return v; }
  void set_color_space(EColorSpace cs) { ok(nrf_rb_set_color_space(m_rb, (int)cs)); }
  void set_tonemap_curve(ETonemapCurve c) { ok(nrf_rb_set_tonemap_curve(m_rb, (int)c)); }
  void* frame_buffer() const { void* p = nullptr; ok(nrf_rb_buffers(m_rb, &p, nullptr, nullptr, nullptr)); return p; }
  void* depth_buffer() const { void* p = nullptr; ok(nrf_rb_buffers(m_rb, nullptr, &p, nullptr, nullptr)); return p; }
  void* accumulate_buffer() const { void* p = nullptr; ok(nrf_rb_buffers(m_rb, nullptr, nullptr, &p, nullptr)); return p; }
  void* surface() const { void* p = nullptr; ok(nrf_rb_buffers(m_rb, nullptr, nullptr, nullptr, &p)); return p; }
  std::vector<float> accumulate_buffer_host() {
    std::vector<float> v((size_t)m_res[0] * m_res[1] * 4);
    ok(nrf_rb_read(m_rb, v.data(), nullptr));
    return v;
  }
  std::vector<float> surface_host() {
    std::vector<float> v((size_t)m_res[0] * m_res[1] * 4);
    ok(nrf_rb_read(m_rb, nullptr, v.data()));
    return v;
  }
  void host_to_accumulate_buffer(const unsigned char* rgb, int size) { ok(nrf_rb_host_to_accumulate_buffer(m_rb, rgb, size)); }
  void clear_frame(void* stream = nullptr) { ok(nrf_rb_clear_frame(m_rb, stream)); }
  void accumulate(float exposure, void* stream = nullptr) { ok(nrf_rb_accumulate(m_rb, exposure, stream)); }
  void tonemap(float exposure, const float background_color[4], EColorSpace output_color_space, void* stream = nullptr) {
    ok(nrf_rb_tonemap(m_rb, exposure, background_color, (int)output_color_space, stream));
  }
  // accumulate() + tonemap() as one pass over the planes (nrf_rb_present); rgba8: optional device uint32 [h][w]
  void present(float exposure, const float background_color[4], EColorSpace output_color_space, void* rgba8 = nullptr, void* stream = nullptr) {
    ok(nrf_rb_present(m_rb, exposure, background_color, (int)output_color_space, rgba8, stream));
  }
  // overlay_depth(), render_buffer.h:259-268: `depth` is a device float plane of `resolution` (e.g. depth_buffer())
  void overlay_depth(float alpha, const float* depth, float depth_scale, const Vector2i& resolution, int fov_axis, float zoom,
                     const float screen_center[2], void* stream = nullptr) {
    ok(nrf_rb_overlay_depth(m_rb, alpha, depth, depth_scale, resolution[0], resolution[1], fov_axis, zoom, screen_center, stream));
  }

 private:
  static void ok(int rc) {
    if (rc != NRF_OK) throw std::runtime_error{std::string("render buffer: ") + nrf_last_error()};
  }
  nrf_render_buffer* m_rb = nullptr;
  Vector2i m_res;
};

}  // namespace ngp
