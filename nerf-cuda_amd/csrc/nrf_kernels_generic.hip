// nrf_kernels_generic.hip -- persistent render kernel, generic instance (any grid / MLP / direction-encoding shape of the JSON vocabulary; one march form)
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

hipError_t launch_persistent_generic(const PersistLaunch& L) {
  const bool wl = L.M->gen_weights_lds != 0;
  if (L.waves == 12 && wl) NRF_LAUNCH_PERSISTENT_W(NET_GENERIC, MARCH_GENERIC, 12, true);
  else if (L.waves == 12) NRF_LAUNCH_PERSISTENT_W(NET_GENERIC, MARCH_GENERIC, 12, false);
  else if (wl) NRF_LAUNCH_PERSISTENT_W(NET_GENERIC, MARCH_GENERIC, 8, true);
  else NRF_LAUNCH_PERSISTENT_W(NET_GENERIC, MARCH_GENERIC, 8, false);
  return hipGetLastError();
}

// the HIP runtime loads a translation unit's code object at the first launch of one of its kernels: touch one here, so that
// nrf_load_model pays for it (once per process and device) and not the first frame (preload_kernels, nrf_kernels.hip)
void preload_generic() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_persistent_kernel<NET_GENERIC, MARCH_GENERIC, 12, false, false, false>));
}

}  // namespace nrf
