// testbed.cpp -- mirror of the reference's `testbed` main (src/main.cu:131-206) without the
// GLFW/Vulkan/DLSS presentation part: load a snapshot, render ONE frame with the hard-coded camera
// and pose, print "Process time", write image.png / deep.png.
//   usage: testbed [snapshot.msgpack] [width height] [out_prefix] [transforms.json]
// With a transforms.json (NeRF-synthetic / instant-ngp camera path) every frame of the path is rendered as well, in
// batches of 32 views per launch, and written to <out_prefix>path_NNNN.rgb.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "nerf_render.h"
#include "png_lite.h"
#include "render_buffer.h"

constexpr size_t PATH_BATCH_VIEWS = 32;  // frames of a camera path rendered (and held) per launch

using namespace ngp;

int main(int argc, char** argv) {
  std::cout << "Hello, Metavese!" << std::endl;
  try {
    NerfRender* render = new NerfRender();
    const std::string config_path = argc > 1 ? argv[1] : "freality.msgpack";
    render->reload_network_from_file(config_path);  // Init Model
    const int W = argc > 3 ? std::atoi(argv[2]) : 4000 / 8, H = argc > 3 ? std::atoi(argv[3]) : 4000 / 8;
    const std::string prefix = argc > 4 ? argv[4] : "./";
    const float s = (float)W / 500.0f;
    Camera cam = {3550.115f / 8 * s, 3554.515f / 8 * s, 3010.45f / 8 * s, 1996.027f / 8 * s};
    Matrix4f pos;
    const float p[16] = {-0.5575427361517304f, -0.11682263918046752f, 0.8218871992959822f, 3.9673954052389253f,
                         0.8300327085486383f,  -0.094966079921629f,   0.5495699649760266f, 2.667431152445114f,
                         0.013849191732089516f, 0.9886020001326434f,  0.14991425965987268f, 0.45955395816033995f,
                         0.0f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 16; ++i) pos.m[i] = p[i];
    render->set_resolution(Vector2i(W, H));
    const auto t0 = std::chrono::steady_clock::now();
    Image img = render->render_frame(cam, pos);
    const auto t1 = std::chrono::steady_clock::now();
    std::printf("Process time : %f s / frame\n", std::chrono::duration<double>(t1 - t0).count());
    const nrf_stats st = render->last_stats();
    std::printf("samples %llu  device time %.3f ms\n", (unsigned long long)st.n_samples, st.render_ms);
    pnglite::write((prefix + "deep.png").c_str(), img.W, img.H, 1, img.depth);
    pnglite::write((prefix + "image.png").c_str(), img.W, img.H, 3, img.rgb);
    // presentation chain of main.cu:87-129,171-206 without DLSS: u8 image -> accumulate buffer -> tonemap (sRGB)
    {
      RenderBuffer rb;
      rb.resize(Vector2i(img.W, img.H));
      rb.reset_accumulation();
      rb.host_to_accumulate_buffer(img.rgb, img.W * img.H);
      const float bg[4] = {0.f, 0.f, 0.f, 1.f};
      rb.tonemap(0.0f, bg, EColorSpace::SRGB);
      const std::vector<float> result = rb.surface_host();
      std::vector<unsigned char> out((size_t)img.W * img.H * 3);
      for (size_t i = 0; i < (size_t)img.W * img.H; ++i)
        for (int j = 0; j < 3; ++j) {
          const float v = result[i * 4 + j] * 255;
          out[i * 3 + j] = (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
        }
      pnglite::write((prefix + "tonemapped.png").c_str(), img.W, img.H, 3, out.data());
    }
    FILE* f = std::fopen((prefix + "image.rgb").c_str(), "wb");  // raw copy for the parity test
    if (f) { std::fwrite(img.rgb, 1, (size_t)img.W * img.H * 3, f); std::fclose(f); }
    if (argc > 5) {
      std::vector<Camera> cams;
      std::vector<Matrix4f> poses;
      load_camera_path(argv[5], W, H, cams, poses);
      const auto p0 = std::chrono::steady_clock::now();
      size_t done = 0;
      for (size_t first = 0; first < cams.size(); first += PATH_BATCH_VIEWS) {
        const size_t n = std::min(cams.size() - first, PATH_BATCH_VIEWS);
        const std::vector<Image> imgs = render->render_frames(std::vector<Camera>(cams.begin() + first, cams.begin() + first + n),
                                                              std::vector<Matrix4f>(poses.begin() + first, poses.begin() + first + n));
        for (size_t i = 0; i < n; ++i, ++done) {
          char name[64];
          std::snprintf(name, sizeof(name), "path_%04zu.rgb", first + i);
          FILE* pf = std::fopen((prefix + name).c_str(), "wb");
          if (pf) { std::fwrite(imgs[i].rgb, 1, (size_t)W * H * 3, pf); std::fclose(pf); }
        }
      }
      const auto p1 = std::chrono::steady_clock::now();
      std::printf("camera path: %zu frames, %f s / frame\n", done, std::chrono::duration<double>(p1 - p0).count() / (double)(done ? done : 1));
    }
    delete render;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
