"""Throughput of the other BASELINE.json configurations on one GPU (they are parity-test cases, not bench lines):
config 4 -- bound 16, 5 cascades, 1024 steps per ray, 1920x1080; config 5 -- 64 camera requests of 800x800."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn


def run(name, desc, W, H, n_views, radius, opts=None, reps=4):
    c = nh.NerfHip(0); c.load_model(desc)
    if opts is not None:
        c.set_options(opts)
    c.set_resolution(W, H); c.set_max_views(n_views)
    cams = np.stack([syn.default_camera(W, H)] * n_views)
    poses = np.stack([syn.orbit_pose(360.0 * i / n_views, 25.0, radius=radius) for i in range(n_views)])
    s = torch.cuda.Stream()
    c.render_views(cams, poses, stream=s.cuda_stream); torch.cuda.synchronize()
    samples = c.stats().n_samples
    t0 = time.perf_counter()
    for _ in range(reps):
        c.render_views(cams, poses, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {n_views} views of {W}x{H} in {dt*1e3:.2f} ms = {dt/n_views*1e3:.3f} ms per view, "
          f"{samples/dt/1e6:.0f} Msamples/s, {samples/n_views/1e6:.2f} M samples per view", flush=True)
    c.close()


desc2, k2, _ = models.build_model(log2_hashmap_size=19, H=128)
run("config 2 (bound 1, 1 cascade)", desc2, 1920, 1080, 16, 4.0311)
run("config 5 (64 requests of 800x800, one launch)", desc2, 800, 800, 64, 4.0311)
desc4, k4, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
o4 = nh.default_options(); o4.max_steps = 1024
# config 4 from a snapshot in instant-ngp's own layout (aabb_scale 32 = bound 16: Morton-ordered fp16 density grid of six
# cascades, params_binary, per_level_scale derived from aabb_scale), read back through nerfhip.desc_from_config
import tempfile
with tempfile.TemporaryDirectory() as td:
    pls = nh.default_per_level_scale(32.0, 16, 16)
    dn, kn, cfgn = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0, per_level_scale=pls)
    f = os.path.join(td, "scene_ngp.msgpack")
    syn.write_ngp_snapshot(f, cfgn, kn[0], kn[1], 32)
    t0 = time.perf_counter()
    desc_ngp, keep_ngp = nh.desc_from_config(syn.read_snapshot(f))
    print(f"instant-ngp layout snapshot: {os.path.getsize(f) / 1e6:.1f} MB, loaded in {time.perf_counter() - t0:.2f} s "
          f"(bound {desc_ngp.bound}, {desc_ngp.cascade} cascades, per_level_scale {desc_ngp.per_level_scale:.6f})", flush=True)
run("config 4, instant-ngp layout snapshot (aabb_scale 32, max_steps 1024)", desc_ngp, 1920, 1080, 16, 4.0311, o4)
run("config 4 (bound 16, 5 cascades, max_steps 1024)", desc4, 1920, 1080, 16, 4.0311, o4)
run("config 4, camera inside the volume (radius 1.5/0.33)", desc4, 1920, 1080, 16, 1.5 / 0.33, o4)
