"""Diagnostic build (make prof) + NRF_MARCH_BUDGET=4095 + NRF_TAIL_SPLIT=0: dumps per-tile cost (Mcycles on one wave), start time,
samples, workgroup-local wave and SIMD of one 1080p view to gpurun_out/tile_cost_az<az>.npy for scripts/sched_sim.py."""
import os, pathlib, sys
os.environ["NRF_MARCH_BUDGET"] = "4095"
os.environ["NRF_TAIL_SPLIT"] = "0"
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path("nerf-cuda_amd/libnerfhip_prof.so").resolve()
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)
W, H = 1920, 1080
c.set_resolution(W, H)
cam = syn.default_camera(W, H)
os.makedirs("gpurun_out", exist_ok=True)
for az in (0, 45, 90, 135):
    for _ in range(2):
        c.render(cam, syn.orbit_pose(az, 30))
    rgba, depth = c.read_f32()
    ms = c.stats().render_ms
    planes = np.stack([depth[::8, k::8][:, :240] for k in range(8)])  # start, cost, samples/1000, bt, simd, wave, slot, (start)
    np.save(f"gpurun_out/tile_cost_az{az}.npy", planes.astype(np.float32))
    print(az, ms, planes[1].sum(), planes[1].max())
    # the same view with the density switched off: no ray terminates early, a tile's samples = its occupied march steps
    o = nh.default_options(); o.density_scale = 1e-9
    c.set_options(o)
    c.render(cam, syn.orbit_pose(az, 30))
    _, d2 = c.read_f32()
    np.save(f"gpurun_out/tile_steps_az{az}.npy", (d2[::8, 2::8][:, :240] * 1e3).astype(np.float32))
    c.set_options(nh.default_options())
