#!/usr/bin/env python3
"""Several device-to-pinned-host copies issued one after the other (1 ms apart) while the persistent render is resident:
when does each one really run?  PROBE_STREAMS=n: round-robin over n high-priority copy streams; PROBE_SRC=render: the
source is the buffer the render writes (bound 8-bit planes)."""
import os
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
n_streams = int(os.environ.get("PROBE_STREAMS", "1"))
sa = torch.cuda.Stream()
sbs = [torch.cuda.Stream(priority=-1) for _ in range(n_streams)]
rgb8 = torch.zeros((V, H, W, 3), dtype=torch.uint8, device="cuda")
d8 = torch.zeros((V, H, W), dtype=torch.uint8, device="cuda")
if os.environ.get("PROBE_SRC") == "render":
    ctx.bind_output_u8(rgb8.data_ptr(), d8.data_ptr())
    src = rgb8.view(-1)
else:
    src = torch.zeros((V * W * H * 3,), dtype=torch.uint8, device="cuda")
dst = torch.zeros((V * W * H * 3,), dtype=torch.uint8, pin_memory=True)
N, chunk = 8, int(os.environ.get("PROBE_CHUNK", W * H * 3))
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = []
    e0.record(sa)
    ctx.render_views(cams, ps, stream=sa.cuda_stream)
    e1.record(sa)
    for i in range(N):
        time.sleep(0.001)
        sb = sbs[i % n_streams]
        with torch.cuda.stream(sb):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if os.environ.get("PROBE_EVENTS", "1") == "1":
                a.record(sb)
            dst[i * chunk:(i + 1) * chunk].copy_(src[i * chunk:(i + 1) * chunk], non_blocking=True)
            b.record(sb)
            evs.append((a, b))
    torch.cuda.synchronize()
    print(f"render {e0.elapsed_time(e1):.2f} ms; copies ended at " + " ".join(f"{e0.elapsed_time(b):.2f}" for a, b in evs), flush=True)
