#!/usr/bin/env python3
"""One render_frame to host bytes (nrf_render_host_u8, one 1080p view per call): wall-clock per call, the device part and the
host tail, over the eight orbit poses (median of 15 calls each).  Run under different NRF_PLAN_BAND_BINS / NRF_PLAN_BAND_THR /
NRF_HOST_MERGE settings to compare (the environment is read at context creation)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import models, nerfhip as nh, synthetic as syn

W, H = 1920, 1080
V = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # views per call
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
cam = np.ascontiguousarray(np.stack([syn.default_camera(W, H)] * V), np.float32)
ctx = nh.NerfHip(0); ctx.load_model(desc); ctx.set_resolution(W, H); ctx.set_max_views(V)
tot_w = tot_d = 0.0
for az in range(0, 360, 45):
    pose = np.ascontiguousarray(np.stack([syn.orbit_pose(float(az + 45 * v), 30.0) for v in range(V)]), np.float32).reshape(V, 16)
    wall, dev = [], []
    for i in range(17):
        t0 = time.perf_counter()
        f = ctx.render_host_u8_raw(cam, pose)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(f.render_ms)
    w, d = float(np.median(wall[2:])), float(np.median(dev[2:]))
    tot_w += w; tot_d += d
print(f"{V} view(s) per call to host bytes: {tot_w / 8:.4f} ms per call, device {tot_d / 8:.4f}, host tail {(tot_w - tot_d) / 8:.4f}")
