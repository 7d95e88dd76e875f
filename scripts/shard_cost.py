"""Kernel time of an N-way tile-sharded frame (shards rendered one after the other on one GPU)
against the unsharded frame: the locality/balance cost of the partition itself."""
import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
ctx.load_model(desc)
W, H = 1920, 1080
cam, pose = syn.default_camera(W, H), syn.orbit_pose(45, 30)
for count in (1, 2, 4, 8):
    times = []
    for idx in range(count):
        o = nh.default_options(); o.shard_index, o.shard_count = idx, count
        ctx.set_options(o)
        ctx.set_resolution(W, H)
        best = 1e9
        for _ in range(5):
            ctx.render(cam, pose)
            best = min(best, ctx.stats().render_ms)
        times.append(best)
    print(f"shards {count}: per-shard ms {['%.3f' % t for t in times]}  sum {sum(times):.3f}  max {max(times):.3f}")
