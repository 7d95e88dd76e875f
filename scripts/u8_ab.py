#!/usr/bin/env python3
"""Device time of one 16-view 1080p launch (and of single views) by output form: float planes, packed rgbd8, 8-bit planar
(nrf_bind_output_u8: the OUT_U8 instance of the persistent kernel), host frames (OUT_U8 + progress reporting + skip_outside).
Usage: python scripts/u8_ab.py [views=16]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import models
import nerfhip as nh
import synthetic as syn

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
cams = np.stack([cam] * V)
ps = [poses[v % 8] for v in range(V)]
ctx = nh.NerfHip(0)
ctx.load_model(desc)
ctx.set_resolution(W, H)
ctx.set_max_views(V)
st = torch.cuda.Stream()
rgba = torch.zeros((V, H, W, 4), device="cuda")
depth = torch.zeros((V, H, W), device="cuda")
packed = torch.zeros((V, H, W), dtype=torch.int32, device="cuda")
rgb8 = torch.zeros((V, H, W, 3), dtype=torch.uint8, device="cuda")
d8 = torch.zeros((V, H, W), dtype=torch.uint8, device="cuda")


def run(name, bind, n=12):
    bind()
    ms = []
    for i in range(n):
        ctx.render_views(cams, ps, stream=st.cuda_stream)
        torch.cuda.synchronize()
        ms.append(ctx.stats().render_ms)
    print(f"{name:28s} {np.mean(ms[2:]) / V:.4f} ms per frame  (min {np.min(ms) / V:.4f})", flush=True)


for rep in range(2):
    run("float planes", lambda: ctx.bind_output(rgba.data_ptr(), depth.data_ptr()))
    run("packed rgbd8", lambda: ctx.bind_output_rgbd8(packed.data_ptr()))
    run("8-bit planar (bound)", lambda: ctx.bind_output_u8(rgb8.data_ptr(), d8.data_ptr()))
    ctx.bind_output(0, 0)
    ms = []
    c32, p32 = np.ascontiguousarray(cams, np.float32), np.ascontiguousarray(np.stack(ps), np.float32).reshape(V, 16)
    ms2 = []
    for i in range(12):
        f = ctx.render_host_u8_raw(c32, p32)
        ms.append(f.render_ms)
        ms2.append(ctx.stats().render_ms)
    print(f"{'host frames':28s} {np.mean(ms[2:]) / V:.4f} ms per frame  (min {np.min(ms) / V:.4f})  ev0..ev1: {np.mean(ms2[2:]) / V:.4f}", flush=True)
