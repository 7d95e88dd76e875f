#!/usr/bin/env python3
"""Barrier fast-forward (nrf_device.h fast_forward_to_barrier) by viewing octant: a ray needs a NEGATIVE direction component
(in the reference's ngp axes) to have barrier planes; rays without one keep their trips.  One 1080p view per launch, kernel-side
render_ms (median of 9), NRF_MARCH_FF=0 against the default, frames compared bit for bit.  The column `neg` is the share of
the centre ray's direction components that are negative."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models, nerfhip as nh, synthetic as syn

desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
W, H = 1920, 1080
cam = syn.default_camera(W, H)
ctx = {}
for name, v in (("trips", "0"), ("fast-forward", "1")):
    os.environ["NRF_MARCH_FF"] = v
    c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H)
    ctx[name] = c
tot = {k: 0.0 for k in ctx}
poses = [(az, el) for el in (30.0, -35.0) for az in (45.0, 135.0, 225.0, 315.0)] + [(0.0, 30.0), (90.0, 0.0)]
for az, el in poses:
    pose = syn.orbit_pose(az, el)
    # centre ray in ngp axes: nerf_matrix_to_ngp cycles the axes (x, y, z) -> (y, z, x); the camera looks along -z of its frame
    dn = -pose[:3, 2]
    d_ngp = np.array([dn[1], dn[2], dn[0]])
    out = {}
    for name, c in ctx.items():
        ms = []
        for _ in range(9):
            c.render(cam, pose)
            ms.append(c.stats().render_ms)
        out[name] = (float(np.median(ms)), c.read_f32())
        tot[name] += out[name][0]
    same = all(np.array_equal(a, b) for a, b in zip(out["trips"][1], out["fast-forward"][1]))
    print(f"az {az:5.1f} el {el:5.1f}  centre ray (ngp) {np.array2string(d_ngp, precision=2, suppress_small=True):>22s}  neg {int((d_ngp < 0).sum())}/3   "
          f"trips {out['trips'][0]:.4f} ms   fast-forward {out['fast-forward'][0]:.4f} ms   ({100 * (out['fast-forward'][0] / out['trips'][0] - 1):+.1f} %)  identical {same}", flush=True)
print(f"mean: trips {tot['trips'] / len(poses):.4f}  fast-forward {tot['fast-forward'] / len(poses):.4f}  ({100 * (tot['fast-forward'] / tot['trips'] - 1):+.1f} %)")
