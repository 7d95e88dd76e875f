"""Throughput against the hash-table size: T = 2^19 (24 MB, lives in L2 / Infinity Cache) ... 2^23 (268 MB, beyond the 256 MB
Infinity Cache): where the gathers become HBM traffic.  16 views of 1920x1080 per launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
W, H, V = 1920, 1080, 16
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(45.0 * (i % 8), 30.0) for i in range(V)])
for log2t in [int(a) for a in sys.argv[1:]] or (19, 20, 21, 22, 23):
    desc, keep, _ = models.build_model(log2_hashmap_size=log2t, H=128)
    c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
    s = torch.cuda.Stream()
    c.render_views(cams, poses, stream=s.cuda_stream); torch.cuda.synchronize()
    samples = c.stats().n_samples
    t0 = time.perf_counter()
    for _ in range(3):
        c.render_views(cams, poses, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    table_mb = nh.expected_n_params(desc) * 2 / 1e6
    print(f"T = 2^{log2t}: parameters {table_mb:7.1f} MB (fp16), {dt/V*1e3:.3f} ms per view, {samples/dt/1e6:.0f} Msamples/s, "
          f"{samples/dt*512/1e12:.2f} TB/s of gathered entries", flush=True)
    c.close()
    del desc, keep
