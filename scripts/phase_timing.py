"""Prints the per-phase share of wave cycles of render_kernel (diagnostic build libnerfhip_prof.so)."""
import ctypes as C, pathlib, sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "nerf-cuda_amd/libnerfhip_prof.so").resolve()
if len(sys.argv) > 2 and sys.argv[2] == "config4":
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
else:
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
ctx.load_model(desc)
W, H = 1920, 1080
ctx.set_resolution(W, H)
cam = syn.default_camera(W, H)
ctx.lib.nrf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for az in (0, 45, 90):
    pose = syn.orbit_pose(az, 30)
    for _ in range(3):
        ctx.render(cam, pose)
    st = ctx.stats()
    out = (C.c_ulonglong * 16)()
    ctx.lib.nrf_debug_counters(ctx.h, out)
    s, r, m, n, c, tot, waves = [int(x) for x in out[:7]]
    slots = int(st.n_network_evals)
    other = tot - m - n - c
    print(f"az {az}: {st.render_ms:.3f} ms samples {s} rounds {r} waves {waves}  samples/round {s/max(r,1):.1f}")
    print(f"   cycles/wave {tot/waves:.0f}  march {100*m/tot:.1f}%  network {100*n/tot:.1f}%  composite {100*c/tot:.1f}%  setup+final {100*other/tot:.1f}%")
    lt, wi = int(out[8]), int(out[9])
    print(f"   march: lane trips {lt} ({lt/max(s,1):.1f}/sample), wave trip-iterations {wi} ({wi/max(r,1):.1f}/round), lane efficiency {100*lt/max(64*wi,1):.1f}%")
    print(f"   setup (raygen, SH, box clip, coarse DDA) {100*int(out[10])/tot:.1f}% of wave cycles = {int(out[10])/waves:.0f} cycles/wave")
    print(f"   MFMA tile slots evaluated {slots} = {100*s/max(slots,1):.1f}% filled")
    print(f"   per round: march {m/r:.0f}  network {n/r:.0f}  composite {c/r:.0f} cycles")
    if int(out[13]) and int(out[14]):  # persistent kernel: the waves' time in the tile loop
        n, total, longest, sched = int(out[13]), int(out[12]), int(out[14]), int(out[15])
        print(f"   tile loop: {n} waves, mean {total/n:.0f} cycles, longest {longest} -> the mean wave is in the loop for {100*total/n/longest:.1f}% "
              f"of the longest one's time; waiting for a tile: {100*sched/total:.2f}% of the loop time; in tiles with live rays: {100*tot/total:.1f}%")
    elif int(out[13]):
        held = int(out[12])
        print(f"   workgroup hold: {int(out[13])} workgroups held their wave slots for {held/tot:.3f} x the waves' own spans "
              f"({100*(1-tot/held):.1f}% of the held slot time is a finished wave waiting for the slowest tile of its strip)")
