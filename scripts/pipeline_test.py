"""Throughput with several frames in flight (one context + stream per in-flight frame)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
W, H = 1920, 1080
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
for count, idx in ((1, 0), (2, 0), (4, 0), (8, 0)):
    for depth in (3, 4, 6, 8):
        ctxs, streams = [], []
        for d in range(depth):
            c = nh.NerfHip(0); c.load_model(desc)
            o = nh.default_options(); o.shard_index, o.shard_count = idx, count
            c.set_options(o); c.set_resolution(W, H)
            ctxs.append(c); streams.append(torch.cuda.Stream())
        def run(n):
            for i in range(n):
                c, s = ctxs[i % depth], streams[i % depth]
                c.render(cam, poses[i % 8], stream=s.cuda_stream)
        run(8); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 64 * count; run(n); ti = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"shards {count} depth {depth}: {dt/n*1e3:.3f} ms per frame-shard (host issue {ti/n*1e3:.3f} ms)")
        for c in ctxs: c.close()
