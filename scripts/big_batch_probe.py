import sys, time
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)
W, H, V = 3840, 2160, 40
c.set_resolution(W, H); c.set_max_views(V)
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(9.0 * i, 30.0) for i in range(V)])
t0 = time.perf_counter(); f = c.render_views(cams, poses); dt = time.perf_counter() - t0
print(f"{V} views of {W}x{H}: {dt*1e3:.1f} ms, samples {c.stats().n_samples}, evals {c.stats().n_network_evals}", flush=True)
a0, _ = c.read_view_f32(0); a39, _ = c.read_view_f32(V - 1); a33, _ = c.read_view_f32(33)
c.render(cams[0], poses[0]); b0, _ = c.read_f32()
c.render(cams[V - 1], poses[V - 1]); b39, _ = c.read_f32()
c.render(cams[33], poses[33]); b33, _ = c.read_f32()
print("views equal single renders:", np.array_equal(a0, b0), np.array_equal(a39, b39), np.array_equal(a33, b33))
