"""One rank's share of a tile-sharded step on one GPU: V views of 1920x1080, strips shard_index of shard_count
(what a rank of `bench.py --gpus N` renders per step in its weak-scaling form: V = 16 N).
usage: scripts/shard_bench.py [shard_count] [shard_index] [views]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = 1920, 1080
V = int(sys.argv[3]) if len(sys.argv) > 3 else 16 * N
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)
o = nh.default_options(); o.shard_count, o.shard_index = N, idx
c.set_options(o); c.set_resolution(W, H)
n_px = nh.tiles_per_shard(W, H, N) * 64
rgba = torch.zeros((V, n_px, 4), device="cuda"); depth = torch.zeros((V, n_px), device="cuda")
c.bind_output(rgba.data_ptr(), depth.data_ptr())
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(45.0 * (i % 8), 30.0) for i in range(V)])
s = torch.cuda.Stream()
for _ in range(2):
    c.render_views(cams, poses, stream=s.cuda_stream)
torch.cuda.synchronize()
samples = c.stats().n_samples
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    c.render_views(cams, poses, stream=s.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"shard {idx}/{N}: {V} views in {dt*1e3:.3f} ms = {dt/V*N*1e3:.4f} ms per whole-frame equivalent, {samples/dt/1e6:.0f} Msamples/s", flush=True)
