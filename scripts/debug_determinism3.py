import sys, os
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

import pathlib
if os.environ.get("NERFHIP_LIB"): nh.LIB_PATH = pathlib.Path(os.environ["NERFHIP_LIB"]).resolve()
big, kb, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
W, H = 1920, 1080
cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
ctx.load_model(big)
ctx.set_resolution(W, H)
N = int(os.environ.get("NFRAMES", "150"))
ref = None
bad = 0
for i in range(N):
    ctx.render(cam, pose)
    a, d = ctx.read_f32()
    if ref is None:
        ref = a.copy(); refd = d.copy(); continue
    diff = np.abs(a - ref).max(axis=2)
    ys, xs = np.nonzero(diff)
    if len(ys):
        bad += 1
        tiles = sorted(set(((int(y) // 8) * 240 + int(x) // 8) for y, x in zip(ys, xs)))
        lanes = sorted(set((int(y) % 8) * 8 + int(x) % 8 for y, x in zip(ys, xs)))
        print(f"frame {i}: {len(ys)} px, max {diff.max():.2e}, tiles {tiles}, lanes {lanes}, depth diff {np.abs(d-refd).max():.2e}")
print(f"{bad} / {N-1} frames differ from frame 0")
