// nerf_render.h -- C++ host mirror of the reference's renderer class on the C ABI.
//
// Same names, argument meaning and error behaviour as ngp::NerfRender
// (reference include/nerf-cuda/nerf_render.h:29-50), ngp::Camera and ngp::Image
// (include/nerf-cuda/common.h:68-89).  Third-party parameter types are replaced
// by small own equivalents: Eigen::Vector2i -> ngp::Vector2i, Eigen::Matrix4f ->
// ngp::Matrix4f (row-major, operator()(row, col)), nlohmann::json -> mpk::Value,
// filesystem::path -> std::string.  Everything device-side happens behind
// include/nerfhip.h; this file contains no HIP.
#pragma once
#include <string>
#include <vector>

#include "../../include/nerfhip.h"
#include "msgpack_lite.h"

namespace ngp {

struct Camera {  // reference common.h:68-74
  float fl_x, fl_y, cx, cy;
};

struct Image {  // reference common.h:76-89; pointers are owned by NerfRender (pinned host memory the GPU's copy engine
                // fills), valid until the second next render call (the reference: until the next render_frame)
  int W, H;
  unsigned char* rgb;    // W * H * 3, row-major
  unsigned char* depth;  // W * H
  Image(int w, int h, unsigned char* in_rgb, unsigned char* in_depth) : W(w), H(h), rgb(in_rgb), depth(in_depth) {}
};

struct Vector2i {
  int v[2];
  Vector2i(int x = 0, int y = 0) : v{x, y} {}
  int& operator[](int i) { return v[i]; }
  int operator[](int i) const { return v[i]; }
};

struct Matrix4f {  // row-major 4x4, camera-to-world in the NeRF/Blender convention
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }
};

class NerfRender {
 public:
  // The reference fixes the device count with the NGPU macro (common.h:91); here it is a
  // constructor argument (or the NERF_NGPU environment variable; NERF_DEVICES="0,1,..." names the
  // devices, repeats allowed).  The devices form an nrf_group: member i renders the tile strips with
  // strip_id % n == i, the shards travel device-to-device to the first member, which untiles
  // (the reference: one thread per device, D2H copies, host de-interleave, nerf_render.cu:252-362).
  // n_gpus < 0: host-only instance that can load and inspect snapshots but not render.
  explicit NerfRender(int n_gpus = 0);
  // the same over an explicit device list (e.g. {3}: a single-device renderer on GPU 3, as a render_server worker)
  explicit NerfRender(const std::vector<int>& devices);
  ~NerfRender();
  NerfRender(const NerfRender&) = delete;
  NerfRender& operator=(const NerfRender&) = delete;

  void reload_network_from_file(const std::string& network_config_path);
  mpk::Value load_network_config(const std::string& network_config_path);
  void reset_network();
  void set_resolution(Vector2i resolution);
  Image render_frame(Camera cam, Matrix4f pos);
  // Batched form (addition): all views in one launch per NRF_MAX_VIEWS cameras (nrf_render_views); every
  // returned Image equals render_frame of that camera.  The images stay valid until the next render call.
  std::vector<Image> render_frames(const std::vector<Camera>& cams, const std::vector<Matrix4f>& poses);
  // Pipelined form of render_frames (addition): submit_frames starts the render + the copy to host memory and returns
  // a ticket at once; wait_frames blocks until that batch is in host memory.  Two batches may be in flight, so the
  // copy (and whatever the caller does with the images) of batch k overlaps the render of batch k + 1; the images of a
  // ticket stay valid until the second next submit_frames.  rgb_only: Image::depth is nullptr, its plane is not copied.
  int submit_frames(const std::vector<Camera>& cams, const std::vector<Matrix4f>& poses, bool rgb_only = false);
  std::vector<Image> wait_frames(int ticket);
  float last_wait_render_ms() const { return m_last_wait_render_ms; }  // device time of the batch wait_frames returned last
  // device ray buffers of the reference are internal to the fused kernel; this fills host copies
  void generate_rays(Camera cam, Matrix4f pos, int threadid);
  // the density grid from the network (nerf_render.cu:388-429, dead and incomplete in the reference; completed in
  // nrf_generate_density_grid); reload_network_from_file calls it for a snapshot that carries no density grid
  void generate_density_grid();
  // Reads the reference's snapshot format (nerf_render.cu:431-473) and, in addition, instant-ngp's own layout
  // (`snapshot.nerf`, `params_binary`, Morton-ordered `density_grid_binary`: see nerf_render.cpp load_ngp_snapshot).
  void load_snapshot(const std::string& filepath_string);

  // additions (the reference has no accessors)
  // Device memory a model may spend on gather copies of its hash grid (nrf_model_desc.gather_copy_budget_mb: 0 = the library's
  // default, a sixteenth of the device's memory; 1 = none); takes effect at the next load_snapshot / reload_network_from_file
  void set_gather_copy_budget_mb(uint32_t mb) { m_gather_copy_budget_mb = mb; }
  int n_gpus() const { return (int)m_ctx.size(); }
  const nrf_model_desc& model_desc() const { return m_desc; }
  nrf_stats last_stats(int gpu = 0) const;
  const std::vector<float>& rays_o() const { return m_rays_o; }
  const std::vector<float>& rays_d() const { return m_rays_d; }

 private:
  void check(int rc, const char* what) const;
  void load_ngp_snapshot(const mpk::Value& config);
  float m_ngp_per_level_scale = 0.0f;  // instant-ngp snapshots: per_level_scale derived from aabb_scale (0: not one)
  bool m_ngp_rgb_sigmoid = false;      //   ... and instant-ngp's logistic colour activation
  nrf_group* m_group = nullptr;
  std::vector<nrf_context*> m_ctx;  // the group's members (owned by the group)
  mpk::Value m_network_config;
  std::string m_network_config_path;
  nrf_model_desc m_desc{};
  uint32_t m_gather_copy_budget_mb = 0;
  bool m_have_snapshot = false, m_have_network = false;
  std::vector<float> m_params, m_density_grid;
  Vector2i resolution;
  float m_last_wait_render_ms = 0.f;
  std::vector<float> m_rays_o, m_rays_d;
};

// Camera path from a NeRF-synthetic / instant-ngp `transforms.json` (intrinsics scaled to width x height; 0 = the file's w / h)
void load_camera_path(const std::string& transforms_json, int width, int height, std::vector<Camera>& cams, std::vector<Matrix4f>& poses);

}  // namespace ngp
