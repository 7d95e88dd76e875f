// nrf_kernels_grid.hip -- persistent render kernel, GRID instances: base.json's MLPs behind another hash grid (F = 1, F = 2 with fewer than 16 levels, F = 4 / 8; Linear, Smoothstep or Nearest)
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

#define NRF_LAUNCH_GRID(G)                                                                                               \
  do {                                                                                                                   \
    if (L.unit) NRF_LAUNCH_PERSISTENT(G, MARCH_UNIT);                                                                    \
    else if (L.pow2) NRF_LAUNCH_PERSISTENT(G, MARCH_POW2);                                                               \
    else NRF_LAUNCH_PERSISTENT(G, MARCH_GENERIC);                                                                        \
  } while (0)

hipError_t launch_persistent_grid(const PersistLaunch& L) {
  const uint32_t f = L.M->hot_grid;
  if (f == 1) NRF_LAUNCH_GRID(NET_GRID1);
  else if (f == 2) NRF_LAUNCH_GRID(NET_GRID2);
  else if (f == 4) NRF_LAUNCH_GRID(NET_GRID4);
  else NRF_LAUNCH_GRID(NET_GRID8);
  return hipGetLastError();
}

// (see preload_kernels, nrf_kernels.hip)
void preload_grid() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_persistent_kernel<NET_GRID4, MARCH_UNIT, persist_waves(NET_GRID4), false, false, false>));
}

}  // namespace nrf
