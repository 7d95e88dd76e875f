"""24 single-view launches (8 poses x 3) after a few batches, for `rocprofv3 --kernel-trace`: scripts/single_view_trace.sh prints
the per-kernel durations (planning kernels, render kernel) and the gaps between them.  usage: single_view_trace.py [lib.so]"""
import os, sys, pathlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
if len(sys.argv) > 1:
    nh.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
W, H, V = 1920, 1080, 16
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
st = torch.cuda.Stream()
for _ in range(4):
    c.render_views(np.stack([cam] * V), np.stack([poses[v % 8] for v in range(V)]), stream=st.cuda_stream)
torch.cuda.synchronize()
ms = []
for rep in range(3):
    for p in poses:
        c.render(cam, p, stream=st.cuda_stream); torch.cuda.synchronize()
        ms.append(c.stats().render_ms)
print(f"{nh.LIB_PATH.name}: event-timed {np.mean(ms):.4f} ms per view")
