import sys
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, oracle_py as op, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0); ctx.load_model(desc)
o = op.Oracle(desc)
W, H = 480, 270
cam = syn.default_camera(1920, 1080) / np.float32(4)
for az in (0, 90):
    pose = syn.orbit_pose(az, 30)
    ctx.set_resolution(W, H); ctx.render(cam, pose)
    g = ctx.stats().n_samples
    rgba, depth = ctx.read_f32()
    want, wd, st = o.render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    ref, rd, st2 = o.render(cam, pose, W, H, schedule=op.SCHED_REFERENCE)
    print(f"az {az}: gpu samples {g}  oracle per-ray {st.n_samples} (needed)  reference schedule {st2.n_samples}  waste {100*(g/st.n_samples-1):.1f}%  maxdiff {np.abs(rgba-want).max():.2e}")
