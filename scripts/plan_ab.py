"""A/B of the planned queue order (plan_kernel) against positions in order: one view per launch, eight orbit poses;
frames must be bit-identical, times are the kernel-side render_ms (median of 7)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models, nerfhip as nh, synthetic as syn
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
cam = syn.default_camera(W, H)
ctx = {}
for name, v in (("plain", "0"), ("planned", "16384")):
    os.environ["NRF_PLAN_MAX_POS"] = v
    c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H)
    ctx[name] = c
tot = {"plain": 0.0, "planned": 0.0}
for az in range(0, 360, 45):
    pose = syn.orbit_pose(az, 30)
    out = {}
    for name, c in ctx.items():
        ms = []
        for _ in range(7):
            c.render(cam, pose)
            ms.append(c.stats().render_ms)
        out[name] = (np.median(ms), c.read_f32(), c.stats())
        tot[name] += np.median(ms)
    same = np.array_equal(out["plain"][1][0], out["planned"][1][0]) and np.array_equal(out["plain"][1][1], out["planned"][1][1])
    print(f"az {az:3d}: plain {out['plain'][0]:.4f} ms  planned {out['planned'][0]:.4f} ms  ({100 * (out['planned'][0] / out['plain'][0] - 1):+.1f} %)  "
          f"identical {same}  composited {out['plain'][2].n_composited} / {out['planned'][2].n_composited}")
print(f"mean: plain {tot['plain'] / 8:.4f}  planned {tot['planned'] / 8:.4f}  ({100 * (tot['planned'] / tot['plain'] - 1):+.1f} %)")
