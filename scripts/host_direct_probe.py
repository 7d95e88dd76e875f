#!/usr/bin/env python3
"""Does the host end of render_frame need a copy at all?  The OUT_U8 instance of the persistent kernel writes the
reference's Image planes with a few dword stores per tile row; bound (nrf_bind_output_u8) to PINNED HOST memory those stores
cross PCIe themselves (posted writes) and the frame is in host memory when the kernel ends.  Measures wall-clock per call
(render + synchronize) for: device planes, pinned host planes, and nrf_render_host_u8 (device planes + progressive copies),
one view and V views per launch; checks that the bytes are the same.
Usage: python scripts/host_direct_probe.py [views=16]"""
import sys
import time
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
ctx = nh.NerfHip(0)
ctx.load_model(desc)
ctx.set_resolution(W, H)
ctx.set_max_views(V)
st = torch.cuda.Stream()

dev_rgb = torch.zeros((V, H, W, 3), dtype=torch.uint8, device="cuda")
dev_d = torch.zeros((V, H, W), dtype=torch.uint8, device="cuda")
pin_rgb = torch.zeros((V, H, W, 3), dtype=torch.uint8).pin_memory()
pin_d = torch.zeros((V, H, W), dtype=torch.uint8).pin_memory()


def run(name, n_views, bind, n=14):
    bind()
    cams = np.stack([cam] * n_views)
    ps = [poses[v % 8] for v in range(n_views)]
    wall, dev = [], []
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render_views(cams, ps, stream=st.cuda_stream)
        st.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(ctx.stats().render_ms)
    print(f"{name:34s} {n_views:2d} views: wall {np.median(wall[2:]) / n_views:.4f} ms per frame (min {np.min(wall) / n_views:.4f}), "
          f"kernel-side {np.median(dev[2:]) / n_views:.4f}", flush=True)


for n_views in (1, V):
    for rep in range(2):
        run("device planes (no copy)", n_views, lambda: ctx.bind_output_u8(dev_rgb.data_ptr(), dev_d.data_ptr()))
        run("pinned host planes (zero-copy)", n_views, lambda: ctx.bind_output_u8(pin_rgb.data_ptr(), pin_d.data_ptr()))
        ctx.bind_output(0, 0)
        c32 = np.ascontiguousarray(np.stack([cam] * n_views), np.float32)
        p32 = np.ascontiguousarray(np.stack([poses[v % 8] for v in range(n_views)]), np.float32).reshape(n_views, 16)
        wall = []
        for i in range(14):
            t0 = time.perf_counter()
            f = ctx.render_host_u8_raw(c32, p32)
            wall.append((time.perf_counter() - t0) * 1e3)
        print(f"{'nrf_render_host_u8 (copies)':34s} {n_views:2d} views: wall {np.median(wall[2:]) / n_views:.4f} ms per frame "
              f"(min {np.min(wall) / n_views:.4f}), kernel-side {f.render_ms / n_views:.4f}", flush=True)
    same = bool((dev_rgb[:n_views].cpu() == pin_rgb[:n_views]).all()) and bool((dev_d[:n_views].cpu() == pin_d[:n_views]).all())
    print(f"   bytes identical (device planes vs pinned planes): {same}", flush=True)
