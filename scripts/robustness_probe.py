"""Unusual inputs must neither hang nor fault: NaN / inf poses, zero focal length, 1x1 and 3840x2160 frames,
camera at the centre of the volume, degenerate axis-aligned directions."""
import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)
def go(name, W, H, cam, pose):
    c.set_resolution(W, H)
    c.render(cam, pose)
    rgba, depth = c.read_f32()
    print(f"{name}: ok, samples {c.stats().n_samples}, finite {bool(np.isfinite(rgba).all())}, alpha max {np.nanmax(rgba[..., 3]):.3f}", flush=True)
cam = syn.default_camera(64, 48)
pose = syn.orbit_pose(30, 30)
bad = pose.copy(); bad[0, 3] = np.nan
go("NaN translation", 64, 48, cam, bad)
bad = pose.copy(); bad[1, 1] = np.inf
go("inf rotation entry", 64, 48, cam, bad)
go("zero pose", 64, 48, cam, np.zeros((4, 4), np.float32))
z = cam.copy(); z[0] = 0
go("zero focal length", 64, 48, z, pose)
go("1x1 frame", 1, 1, syn.default_camera(1, 1), pose)
go("7x5 frame", 7, 5, syn.default_camera(7, 5), pose)
centre = np.eye(4, dtype=np.float32)
go("camera at the origin, axis-aligned", 64, 48, cam, centre)
go("3840x2160 frame", 3840, 2160, syn.default_camera(3840, 2160), pose)
print("all done")
