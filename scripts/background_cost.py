"""Cost of tiles that never sample: a camera so far away that the object is sub-tile-size, vs the bench view."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
W, H = 1920, 1080
cam = syn.default_camera(W, H)
c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(8)
s = torch.cuda.Stream()
for name, radius in (("bench view", 4.0311), ("object far away (all background)", 400.0)):
    poses = np.stack([syn.orbit_pose(45.0 * i, 30.0, radius=radius) for i in range(8)])
    cams = np.stack([cam] * 8)
    for _ in range(3):
        c.render_views(cams, poses, stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c.render_views(cams, poses, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 80
    print(f"{name}: {dt*1e3:.4f} ms per frame, samples of last batch {c.stats().n_samples}")
