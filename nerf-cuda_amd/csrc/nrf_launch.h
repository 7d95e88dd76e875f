// nrf_launch.h -- host-callable launchers implemented in nrf_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace nrf {
struct DevModel;
struct FrameParams;
struct ViewBatch;

// one launch for VB.n_views (<= MAX_VIEWS) cameras; blocks_per_view is filled in by the launcher
// the persistent kernel's work-queue counters follow the statistics counters (counters + COUNTER_BYTES); queues_are_zero:
// the caller has just cleared them together with the statistics
constexpr int RENDER_QUEUE_BYTES = 8 * 4;
constexpr int WIDE_GENERIC_MARCH_WAVES = 8;  // waves of the persistent workgroup of NET_WIDE with the generic march (nrf_render.h persist_waves)
hipError_t launch_render(const DevModel& M, const FrameParams& P, const ViewBatch& VB, void* rgba, void* depth, void* counters,
                         hipStream_t st, bool first_launch, unsigned* plan = nullptr, unsigned plan_cap = 0);
// a call's plan buffer (plan_price_kernel / plan_sort_kernel): 4 header words + a price and an order entry per queue position
constexpr unsigned PLAN_CAP = 60u * 1024u;  // queue positions (strips) of a launch that may be planned (its last workgroup keeps a byte per position in LDS)
constexpr size_t PLAN_BYTES = (4 + 2 * (size_t)PLAN_CAP) * 4;
// Builds the cell-major quad copy of one level of an F = 2 grid (nrf_device.h level_gather_quad) on the device: table = the level's
// entries in the reference order, quads = res * res * (res + 1) 16-byte entries.
hipError_t launch_build_quads(const void* table, uint32_t res, uint32_t size, bool hashed, void* quads, hipStream_t st);
hipError_t launch_encode_grid(const DevModel& M, const void* pos01, uint32_t n, void* out, hipStream_t st, bool fast_interp = false);
hipError_t launch_encode_dir(const DevModel& M, const void* dir01, uint32_t n, void* out, hipStream_t st);
hipError_t launch_mlp_forward(const DevModel& M, const void* feat, const void* dirfeat, uint32_t n, void* out, uint32_t repeat,
                              hipStream_t st);
hipError_t launch_network(const DevModel& M, const void* xyz, const void* dir, uint32_t n, void* sigma, void* rgb, hipStream_t st);
hipError_t launch_density_positions(uint32_t H, float k, void* xyz, void* dir, hipStream_t st);
hipError_t launch_density_update(const void* sigma, uint32_t n, float decay, int n_iterations, void* grid, hipStream_t st);
hipError_t launch_generate_rays(const DevModel& M, const FrameParams& P, void* rays_o, void* rays_d, void* nears, void* fars,
                                hipStream_t st);
hipError_t launch_march(const DevModel& M, float dt_gamma, const void* rays_o, const void* rays_d, const void* rays_t,
                        const void* fars, uint32_t n, uint32_t n_step, void* xyzs, void* dirs, void* deltas, hipStream_t st,
                        uint32_t perturb = 0);
hipError_t launch_composite(const void* sigmas, const void* rgbs, const void* deltas, uint32_t n, uint32_t n_step, void* rays_t,
                            void* state, hipStream_t st);
hipError_t launch_untile(const void* gathered, int shard_count, int tiles_per_shard, int C, int W, int H, int n_views, void* out,
                         hipStream_t st);
hipError_t launch_untile_rgbd8_u8(const void* gathered, int shard_count, int tiles_per_shard, int W, int H, int n_views, void* rgb8,
                                  void* depth8, hipStream_t st);
hipError_t launch_quantize_rgbd8(const void* rgba, const void* depth, uint64_t n, void* out, hipStream_t st);
hipError_t launch_quantize(const void* rgba, const void* depth, int n, void* rgb8, void* depth8, hipStream_t st);
// The march form of a model and the waves of its persistent workgroup: ONE rule, read by set_density_grid (which sizes the
// workgroup's LDS and decides whether the model fits the persistent kernel) and by launch_render (which picks the kernel
// instance); the launch macros refuse a workgroup size other than their instance's.
enum : int { MARCH_FORM_GENERIC = 0, MARCH_FORM_UNIT = 1, MARCH_FORM_POW2 = 2 };
inline int march_form(uint32_t H, uint32_t cascade, float bound) {
  const bool pow2_h = H != 0 && (H & (H - 1)) == 0;
  int eb = 0;
  if (pow2_h && cascade == 1 && bound >= 1.0f) return MARCH_FORM_UNIT;                                  // one cascade with mip_bound == 1
  if (pow2_h && cascade > 1 && bound >= 1.0f && frexpf(bound, &eb) == 0.5f) return MARCH_FORM_POW2;    // several, power-of-two bound
  return MARCH_FORM_GENERIC;
}
// waves of the persistent workgroup of the instance that renders such a model (hot_width: 0 or 16 / 32 / 64 / 128; hot_grid: 0 or the
// F of a GRID instance); the generic
// instance may also run WITH 8 (set_density_grid tries 12 first)
int render_persist_waves_for(uint32_t generic, uint32_t wide, uint32_t wide_sh, uint32_t hot_width, uint32_t hot_grid, int form);
// loads the code objects of the render kernel's instance families now instead of at their first launch
void preload_kernels(bool all);
int render_lds_bytes();
int render_persistent_lds_fixed_bytes(uint32_t generic, uint32_t wide, uint32_t gen_wave_bytes, int waves);
int render_persistent_lds_width_bytes(int width);  // LDS of a width instance's persistent workgroup (16 waves) without its march tables
int render_persistent_lds_widesh_bytes();
int render_width_frags(int width);                 // weight fragments of MlpShape<width>
int render_lds_table_max_bytes();
int render_wide_lds_fixed_bytes();  // wide instance: LDS of render_kernel without the march tables
int render_gen_lds_fixed_bytes(uint32_t gen_wave_bytes);  // generic instance: LDS of render_kernel without the march tables
}  // namespace nrf
