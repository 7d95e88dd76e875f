#!/usr/bin/env python3
"""Device time of the bench.py launch (16 views of 1920x1080, float planes) with a given build of the library.
   python scripts/launch_ms.py <lib.so> [reps]  ->  one line: mean_ms min_ms n_samples n_network_evals [wave_trip_iterations]"""
import ctypes as C
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

nh.LIB_PATH = Path(sys.argv[1]).resolve()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
V, W, H = 16, 1920, 1080
desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
cams = np.stack([cam] * V)
ps = [syn.orbit_pose(45.0 * (v % 8), 30.0) for v in range(V)]
ctx = nh.NerfHip(0)
ctx.load_model(desc)
ctx.set_resolution(W, H)
rgba = torch.zeros((V, H, W, 4), device="cuda")
depth = torch.zeros((V, H, W), device="cuda")
ctx.bind_output(rgba.data_ptr(), depth.data_ptr())
st = torch.cuda.Stream()
ms = []
for i in range(reps + 2):
    ctx.render_views(cams, ps, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms.append(ctx.stats().render_ms)
s = ctx.stats()
extra = ""
if "prof" in nh.LIB_PATH.name:
    ctx.lib.nrf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * 16)()
    ctx.lib.nrf_debug_counters(ctx.h, out)
    extra = f" {int(out[9])}"
print(f"{np.mean(ms[2:]):.4f} {np.min(ms[2:]):.4f} {s.n_samples} {s.n_network_evals}{extra}", flush=True)
