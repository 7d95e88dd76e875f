"""Throughput of frame shards: per-frame-shard time for (shard count, views per launch, launches in flight)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
W, H = 1920, 1080
cam = syn.default_camera(W, H)
for count in (1, 2, 4, 8):
    for views, depth in ((1, 1), (1, 3), (16, 1), (16, 2)):
        ctxs, streams = [], []
        for d in range(depth):
            c = nh.NerfHip(0); c.load_model(desc)
            o = nh.default_options(); o.shard_index, o.shard_count = 0, count
            c.set_options(o); c.set_resolution(W, H); c.set_max_views(views)
            ctxs.append(c); streams.append(torch.cuda.Stream())
        cams = np.stack([cam] * views)
        def run(n):
            for i in range(n):
                c, s = ctxs[i % depth], streams[i % depth]
                poses = np.stack([syn.orbit_pose(45.0 * ((i * views + v) % 8), 30.0) for v in range(views)])
                c.render_views(cams, poses, stream=s.cuda_stream)
        n = max(4, 128 * count // views)
        run(2 * depth); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"shards {count} views/launch {views} in flight {depth}: {dt/(n*views)*1e3:.4f} ms per frame-shard "
              f"(x{count} = {dt/(n*views)*1e3*count:.3f} ms per whole frame)", flush=True)
        for c in ctxs: c.close()
