import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

small, ks, _ = models.build_model(log2_hashmap_size=12, H=32)
big, kb, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
W, H = 1920, 1080
cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
for trial in range(16):
    ctx.load_model(small)
    ctx.set_resolution(64, 64)
    ctx.render(syn.default_camera(64, 64), pose)
    ctx.load_model(big)
    ctx.set_resolution(W, H)
    fr = []
    for i in range(4):
        ctx.render(cam, pose)
        a, d = ctx.read_f32()
        fr.append(a.copy())
    for i in range(3):
        diff = np.abs(fr[i] - fr[3]).max(axis=2)
        ys, xs = np.nonzero(diff)
        print(f"trial {trial} frame {i} vs 3: {len(ys)} px differ max {diff.max():.2e}", [(int(x), int(y)) for y, x in list(zip(ys, xs))[:6]])
