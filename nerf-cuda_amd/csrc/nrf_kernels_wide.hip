// nrf_kernels_wide.hip -- persistent render kernel, wide instances: Frequency directions up to 80 values (NET_WIDE), SH degree 5..8 with per-ray rows (NET_WIDE_SH)
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

hipError_t launch_persistent_wide(const PersistLaunch& L) {
  if (L.M->wide_sh) {
    if (L.unit) NRF_LAUNCH_PERSISTENT(NET_WIDE_SH, MARCH_UNIT); else NRF_LAUNCH_PERSISTENT(NET_WIDE_SH, MARCH_GENERIC);
  } else {
    if (L.unit) NRF_LAUNCH_PERSISTENT(NET_WIDE, MARCH_UNIT);
    else if (L.pow2) NRF_LAUNCH_PERSISTENT(NET_WIDE, MARCH_POW2);
    else NRF_LAUNCH_PERSISTENT_W(NET_WIDE, MARCH_GENERIC, WIDE_GENERIC_MARCH_WAVES, false);  // (nrf_api.hip sets persist_waves alike)
  }
  return hipGetLastError();
}

// the HIP runtime loads a translation unit's code object at the first launch of one of its kernels: touch one here, so that
// nrf_load_model pays for it (once per process and device) and not the first frame (preload_kernels, nrf_kernels.hip)
void preload_wide() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_persistent_kernel<NET_WIDE, MARCH_UNIT, persist_waves(NET_WIDE), false, false, false>));
}

}  // namespace nrf
