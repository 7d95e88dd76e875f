// nrf_kernels_hot.hip -- persistent render kernel, register-resident instance of the base.json shape (BASELINE config 2: the headline)
// (one family of render-kernel instances per translation unit: nrf_render.h)
#include "nrf_render.h"

namespace nrf {

// fast_interp (opt-in single-rounding interpolation) has instances of its own, so that the default symbols keep the bit-exact arithmetic
#define NRF_LAUNCH_HOT(U)                                                                                                \
  do {                                                                                                                   \
    if (L.P->out_mode == OUT_U8) {                                                                                       \
      if (L.P->fast_interp) NRF_LAUNCH_PERSISTENT_F(NET_HOT, U, persist_waves(NET_HOT), false, true, true);              \
      else NRF_LAUNCH_PERSISTENT_F(NET_HOT, U, persist_waves(NET_HOT), false, true, false);                              \
    } else {                                                                                                             \
      if (L.P->fast_interp) NRF_LAUNCH_PERSISTENT_F(NET_HOT, U, persist_waves(NET_HOT), false, false, true);             \
      else NRF_LAUNCH_PERSISTENT_F(NET_HOT, U, persist_waves(NET_HOT), false, false, false);                             \
    }                                                                                                                    \
  } while (0)

hipError_t launch_persistent_hot(const PersistLaunch& L) {
  if (L.unit) NRF_LAUNCH_HOT(MARCH_UNIT);
  else if (L.pow2) NRF_LAUNCH_HOT(MARCH_POW2);
  else NRF_LAUNCH_HOT(MARCH_GENERIC);
  return hipGetLastError();
}

// the HIP runtime loads a translation unit's code object at the first launch of one of its kernels: touch one here, so that
// nrf_load_model pays for it (once per process and device) and not the first frame (preload_kernels, nrf_kernels.hip)
void preload_hot() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&render_persistent_kernel<NET_HOT, MARCH_UNIT, persist_waves(NET_HOT), false, false, false>));
}

}  // namespace nrf
