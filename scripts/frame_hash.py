"""Bit-level fingerprint of frames rendered with a given library build: two builds that claim the same
semantics must print identical lines (used after every "exact" optimisation of the kernel).
usage: scripts/frame_hash.py [path/to/libnerfhip.so]"""
import hashlib, pathlib, sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
if len(sys.argv) > 1:
    nh.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)


def render(cam, pose, W, H):
    """Into poisoned bound buffers: a pixel the kernel does not write shows up in the hash."""
    rgba = torch.full((H, W, 4), 7.0, device="cuda")
    depth = torch.full((H, W), 7.0, device="cuda")
    torch.cuda.synchronize()
    c.bind_output(rgba.data_ptr(), depth.data_ptr())
    c.render(cam, pose)
    c.bind_output(0, 0)
    return rgba.cpu().numpy(), depth.cpu().numpy()


for (W, H) in ((1920, 1080), (800, 800), (333, 211)):
    c.set_resolution(W, H)
    cam = syn.default_camera(W, H)
    for az, el in ((0, 30), (45, 30), (90, 30), (135, -20), (200, 60), (290, 5)):
        rgba, depth = render(cam, syn.orbit_pose(az, el), W, H)
        print(W, H, az, el, hashlib.sha1(rgba.tobytes()).hexdigest()[:16], hashlib.sha1(depth.tobytes()).hexdigest()[:16],
              c.stats().n_samples, c.stats().n_composited)

# BASELINE config 4 shape: bound 16, 5 cascades (generic march instance, per-cascade visibility walk)
desc4, keep4, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
c.load_model(desc4)
o4 = nh.default_options(); o4.max_steps = 1024
c.set_options(o4)
for (W, H) in ((640, 360), (201, 133)):
    c.set_resolution(W, H)
    cam = syn.default_camera(W, H)
    for az, el, radius in ((0, 30, 4.0311), (120, -15, 1.5 / 0.33), (250, 70, 9.0 / 0.33), (33, 5, 0.4 / 0.33)):
        rgba, depth = render(cam, syn.orbit_pose(az, el, radius=radius), W, H)
        print("c4", W, H, az, el, radius, hashlib.sha1(rgba.tobytes()).hexdigest()[:16], hashlib.sha1(depth.tobytes()).hexdigest()[:16],
              c.stats().n_samples, c.stats().n_composited)
