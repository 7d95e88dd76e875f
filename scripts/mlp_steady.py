#!/usr/bin/env python3
"""mlp_forward_kernel (both MLPs on resident fp16 rows): fraction of the dense fp16 MFMA peak, fed from HBM at 2^22 and 2^24
samples per launch (the larger one: steady state -- launch ramp and tail are a smaller share), and register-resident (every
chunk evaluated 64 times).  NRF_MLP_INTERLEAVE=k forces the issue order "one MFMA, k vector instructions" on the body."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import models
import nerfhip as nh

FLOP, PEAK = 20480, 2500.0
desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
ctx = nh.NerfHip(0)
ctx.load_model(desc)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
res = {}
with torch.cuda.stream(st):
    for log2n in (22, 24):
        n = 1 << log2n
        feat = (torch.rand((n, 32), device=dev) - 0.5).half()
        dirf = (torch.rand((n, 16), device=dev) - 0.5).half()
        out = torch.empty((n, 4), dtype=torch.float16, device=dev)
        torch.cuda.synchronize(dev)
        for rep, key in ((1, f"hbm_2^{log2n}"), (64, f"resident_2^{log2n}")):
            if rep == 64 and log2n == 24:
                continue
            f = (lambda: ctx.mlp_forward(feat.data_ptr(), dirf.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream)) if rep == 1 else \
                (lambda: ctx.mlp_forward_repeat(feat.data_ptr(), dirf.data_ptr(), n, out.data_ptr(), rep, stream=st.cuda_stream))
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 if rep == 1 else 4
            e0.record(st)
            for _ in range(reps):
                f()
            e1.record(st)
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            res[key] = (ms, n * rep * FLOP / (ms * 1e-3) / 1e12)
print(f"NRF_MLP_INTERLEAVE={os.environ.get('NRF_MLP_INTERLEAVE', '0')}: " +
      "; ".join(f"{k}: {ms:.4f} ms, {tf:.0f} TFLOP/s = {tf / PEAK:.4f} of peak" for k, (ms, tf) in res.items()), flush=True)
