// nrf_kernels_strip.hip -- render_kernel: one workgroup per strip of four tiles (models whose march tables do not fit beside the persistent workgroup; NRF_PERSISTENT=0)
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

#define NRF_LAUNCH_RENDER(G, C, U)                                                                                       \
  do {                                                                                                                   \
    hipError_t e_ = allow_lds(render_kernel<G, C, U>, L.lds);                                                            \
    if (e_ != hipSuccess) return e_;                                                                                     \
    hipLaunchKernelGGL((render_kernel<G, C, U>), dim3(L.blocks), dim3(RENDER_THREADS), L.lds, L.st, *L.M, *L.P, *L.VB,   \
                       (float4*)L.rgba, (float*)L.depth, (unsigned long long*)L.counters);                               \
  } while (0)

#define NRF_LAUNCH_RENDER_PERTURB(G)                                                                                      \
  do {                                                                                                                   \
    hipError_t e_ = allow_lds(render_kernel<G, false, MARCH_GENERIC, true>, L.lds);                                      \
    if (e_ != hipSuccess) return e_;                                                                                     \
    hipLaunchKernelGGL((render_kernel<G, false, MARCH_GENERIC, true>), dim3(L.blocks), dim3(RENDER_THREADS), L.lds, L.st, *L.M, *L.P, \
                       *L.VB, (float4*)L.rgba, (float*)L.depth, (unsigned long long*)L.counters);                        \
  } while (0)

hipError_t launch_strip(const StripLaunch& L) {
  if (L.perturb) {  // nrf_options.perturb > 0 (render_utils.h:585-589): three instances of their own, march tables in global memory
    if (L.lds_tab) return hipErrorInvalidConfiguration;
    if (L.M->generic) NRF_LAUNCH_RENDER_PERTURB(NET_GENERIC);
    else if (L.M->wide) NRF_LAUNCH_RENDER_PERTURB(NET_WIDE);
    else NRF_LAUNCH_RENDER_PERTURB(NET_HOT);
    return hipGetLastError();
  }
  if (L.M->generic) {
    if (L.lds_tab) NRF_LAUNCH_RENDER(NET_GENERIC, true, MARCH_GENERIC); else NRF_LAUNCH_RENDER(NET_GENERIC, false, MARCH_GENERIC);
  } else if (L.M->wide) {
    if (L.unit) NRF_LAUNCH_RENDER(NET_WIDE, true, MARCH_UNIT);
    else if (L.pow2) NRF_LAUNCH_RENDER(NET_WIDE, true, MARCH_POW2);
    else if (L.lds_tab) NRF_LAUNCH_RENDER(NET_WIDE, true, MARCH_GENERIC);
    else NRF_LAUNCH_RENDER(NET_WIDE, false, MARCH_GENERIC);
  } else {
    if (L.unit) NRF_LAUNCH_RENDER(NET_HOT, true, MARCH_UNIT);
    else if (L.pow2) NRF_LAUNCH_RENDER(NET_HOT, true, MARCH_POW2);
    else if (L.lds_tab) NRF_LAUNCH_RENDER(NET_HOT, true, MARCH_GENERIC);
    else NRF_LAUNCH_RENDER(NET_HOT, false, MARCH_GENERIC);
  }
  return hipGetLastError();
}

// the HIP runtime loads a translation unit's code object at the first launch of one of its kernels: touch one here, so that
// nrf_load_model pays for it (once per process and device) and not the first frame (preload_kernels, nrf_kernels.hip)
void preload_strip() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_kernel<NET_HOT, true, MARCH_UNIT>));
}

}  // namespace nrf
