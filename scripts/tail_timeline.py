#!/usr/bin/env python3
"""Single 1080p view with the diagnostic build (make prof): when does every wave of the persistent kernel leave, per
workgroup (= CU)?  Shows what is left of a frame's tail after tail splitting: the spread between compute units."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import models
import nerfhip as nh
import synthetic as syn

nh.LIB_PATH = ROOT / "nerf-cuda_amd" / "libnerfhip_prof.so"
W, H = 1920, 1080
desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0)
c.load_model(desc)
c.set_resolution(W, H)
cam = syn.default_camera(W, H)
c.lib.nrf_debug_wave_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
for az in (0.0, 45.0, 90.0, 135.0):
    pose = syn.orbit_pose(az, 30.0)
    for _ in range(3):
        c.render(cam, pose)
    wt = (C.c_ulonglong * (2 * 4096))()
    assert c.lib.nrf_debug_wave_times(c.h, wt, 4096) == 0
    a = np.frombuffer(wt, np.uint64).reshape(4096, 2).astype(np.int64)
    helped = a[:, 1] & 0xff
    # stamps of s_memrealtime: one 100 MHz counter for the whole device; unit below: microseconds
    t0 = a[:, 0].min()
    beg = (a[:, 0] - t0) * 1e-2
    end = ((a[:, 1] >> 8) - t0) * 1e-2
    wg_beg = beg.reshape(256, 16).min(axis=1)
    print("   workgroup starts (us), percentiles 0/10/50/90/99/100:", " ".join(f"{np.percentile(wg_beg, q):.0f}" for q in (0, 10, 50, 90, 99, 100)),
          "; wave start spread inside a workgroup (max):", f"{(beg.reshape(256, 16).max(axis=1) - wg_beg).max():.0f}")
    wg_end = end.reshape(256, 16).max(axis=1)
    print(f"az {az:5.1f}: render_ms {c.stats().render_ms:.3f}; waves start {beg.min():.0f}..{beg.max():.0f} us; last wave leaves at {end.max():.0f} us; "
          f"mean wave end {end.mean():.0f} us; workgroup ends: min {wg_end.min():.0f} p10 {np.percentile(wg_end, 10):.0f} median {np.median(wg_end):.0f} "
          f"p90 {np.percentile(wg_end, 90):.0f} max {wg_end.max():.0f} us; waves that helped: {(helped > 0).sum()}, rays taken over (capped 255/wave): {helped.sum()}", flush=True)
