#!/usr/bin/env python3
"""Does a device-to-pinned-host copy run WHILE the persistent render kernel is resident?  Launches a 16-view 1080p render
(~13 ms) on one stream, then issues a 133 MB hipMemcpyAsync (torch copy_) on another stream with no dependency, and reports
when the copy ended relative to the render."""
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

V, W, H = 16, 1920, 1080
desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
cams = np.stack([cam] * V)
ps = [syn.orbit_pose(45.0 * (v % 8), 30.0) for v in range(V)]
ctx = nh.NerfHip(0)
ctx.load_model(desc)
ctx.set_resolution(W, H)
ctx.set_max_views(V)
import os
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1 if os.environ.get("PROBE_PRIO") else 0)
src = torch.zeros((V * W * H * 4,), dtype=torch.uint8, device="cuda")
dst = torch.zeros((V * W * H * 4,), dtype=torch.uint8, pin_memory=True)
for rep in range(4):
    torch.cuda.synchronize()
    e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    e0.record(sa)
    ctx.render_views(cams, ps, stream=sa.cuda_stream)
    e1.record(sa)
    time.sleep(0.002)  # the kernel is resident now
    t0 = time.perf_counter()
    with torch.cuda.stream(sb):
        c0.record(sb)
        dst.copy_(src, non_blocking=True)
        c1.record(sb)
    t_issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print(f"render {e0.elapsed_time(e1):.2f} ms; copy issued ~2 ms after launch (call took {t_issue:.3f} ms), ran {c0.elapsed_time(c1):.2f} ms, "
          f"ended {e0.elapsed_time(c1):.2f} ms after the render's start", flush=True)
