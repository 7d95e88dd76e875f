// nrf_kernels_width.hip -- persistent render kernel, register-resident instances for 16 / 32 / 128 neurons and for other numbers of hidden layers at 64
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

hipError_t launch_persistent_width(const PersistLaunch& L) {
  const uint32_t w = L.M->hot_width;
  if (w == 16) { if (L.unit) NRF_LAUNCH_PERSISTENT(NET_W16, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_W16, MARCH_GENERIC); }
  else if (w == 32) { if (L.unit) NRF_LAUNCH_PERSISTENT(NET_W32, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_W32, MARCH_GENERIC); }
  else if (w == 64) { if (L.unit) NRF_LAUNCH_PERSISTENT(NET_DEPTH, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_DEPTH, MARCH_GENERIC); }
  else if (w == HOT_WIDTH_ACT) { if (L.unit) NRF_LAUNCH_PERSISTENT(NET_ACT, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_ACT, MARCH_GENERIC); }
  else { if (L.unit) NRF_LAUNCH_PERSISTENT(NET_W128, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_W128, MARCH_GENERIC); }
  return hipGetLastError();
}

// the HIP runtime loads a translation unit's code object at the first launch of one of its kernels: touch one here, so that
// nrf_load_model pays for it (once per process and device) and not the first frame (preload_kernels, nrf_kernels.hip)
void preload_width() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_persistent_kernel<NET_W32, MARCH_UNIT, persist_waves(NET_W32), false, false, false>));
}

}  // namespace nrf
