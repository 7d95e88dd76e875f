import ctypes as C, pathlib, sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path("nerf-cuda_amd/libnerfhip_prof.so").resolve()
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0); ctx.load_model(desc)
W, H = 1920, 1080
ctx.set_resolution(W, H)
ctx.lib.nrf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for az in (0, 90):
    ctx.render(syn.default_camera(W, H), syn.orbit_pose(az, 30))
    out = (C.c_ulonglong * 8)(); ctx.lib.nrf_debug_counters(ctx.h, out)
    s, r, t_nos, rays_nos, rays_s, _, _, trips = [int(x) for x in out]
    print(f"az {az}: samples {s} trips {trips}; rays w/ samples {rays_s} ({(trips-t_nos)/max(rays_s,1):.1f} trips each); "
          f"rays marching but sample-free {rays_nos} ({t_nos/max(rays_nos,1):.1f} trips each, {100*t_nos/trips:.1f}% of all trips)")
