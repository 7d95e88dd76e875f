import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
W, H, V = 1920, 1080, 16
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
st = torch.cuda.Stream()
cams = np.stack([cam] * V); pv = np.stack([poses[v % 8] for v in range(V)])
for _ in range(3):
    c.render_views(cams, pv, stream=st.cuda_stream)
torch.cuda.synchronize()
# one at a time, synchronised
ms = []
for rep in range(3):
    for p in poses:
        c.render(cam, p, stream=st.cuda_stream); torch.cuda.synchronize(); ms.append(c.stats().render_ms)
print(f"one view per launch, host sync after each: {np.mean(ms[8:]):.4f} ms (event-timed per launch)")
# back to back on one stream
for n in (8, 24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(st)
    for i in range(n):
        c.render(cam, poses[i % 8], stream=st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    print(f"{n} single-view launches queued back to back on one stream: {e0.elapsed_time(e1) / n:.4f} ms per view")
# 8 views in one launch
for nv in (2, 4, 8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cv = np.stack([cam] * nv); pp = np.stack(poses[:nv])
    c.render_views(cv, pp, stream=st.cuda_stream); torch.cuda.synchronize()
    e0.record(st)
    for _ in range(4):
        c.render_views(cv, pp, stream=st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    print(f"{nv} views per launch: {e0.elapsed_time(e1) / 4 / nv:.4f} ms per view")
