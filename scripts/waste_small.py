#!/usr/bin/env python3
"""Evaluated over composited samples of SMALL frames (the guards of tests/test_parity_gpu.py, test_generic_gpu.py and
test_golden.py): a ray queues up to eight samples per round, those behind its terminating one are evaluated for nothing; a
launch with fewer tiles than the chip has waves is all tail (idle waves split the rendering ones' rays from the first round
on), which is why such launches keep the transmittance-dependent queue since round 5 (nrf_api.hip render_views_impl).
Prints the ratio per case, five renders each (the evaluated count depends on timing): the guards are set above the maximum.
usage: python scripts/waste_small.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402
import nerfhip as nh  # noqa: E402
import synthetic as syn  # noqa: E402
from test_generic_gpu import SHAPES  # noqa: E402


def ratios(ctx, W, H, az, el, radius=4.0311, reps=5):
    ctx.set_resolution(W, H)
    out = []
    for _ in range(reps):
        ctx.render(syn.default_camera(W, H), syn.orbit_pose(az, el, radius=radius))
        st = ctx.stats()
        out.append(st.n_samples / max(st.n_composited, 1))
    return max(out), int(st.n_composited)


def main():
    ctx = nh.NerfHip(0)
    print("tiny model (T = 2^12, H = 32): tests/test_parity_gpu.py test_render_frame_matches_oracle")
    desc, keep, _ = models.build_model(log2_hashmap_size=12, H=32)
    ctx.load_model(desc)
    for W, H, az, el in ((64, 64, 30, 30), (100, 52, 135, 10), (8, 8, 300, 45), (33, 70, 250, -20), (72, 48, 215, 25)):
        r, n = ratios(ctx, W, H, az, el)
        print(f"  {W}x{H}: max evaluated / composited {r:.4f}  ({n} composited)")
    print("other shapes, 72x48 (tests/test_generic_gpu.py)")
    for name in sorted(SHAPES):
        d, k, _ = models.build_model(log2_hashmap_size=12, H=32, **SHAPES[name])
        ctx.load_model(d)
        r, n = ratios(ctx, 72, 48, 215, 25)
        print(f"  {name:20s} {r:.4f}  ({n} composited)")
    print("config-2 model (tests/test_golden.py sizes)")
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
    ctx.load_model(desc)
    for W, H in ((333, 211), (640, 360), (800, 800), (1920, 1080)):
        worst = 0.0
        for az, el in ((0, 30), (45, 30), (90, 30), (135, -20), (200, 60), (290, 5)):
            r, n = ratios(ctx, W, H, az, el, reps=3)
            worst = max(worst, r)
        print(f"  {W}x{H}: max over six poses {worst:.4f}")
    desc4, keep4, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
    ctx.load_model(desc4)
    o = nh.default_options()
    o.max_steps = 1024
    ctx.set_options(o)
    for W, H in ((201, 133), (640, 360)):
        worst = 0.0
        for az, el, radius in ((0, 30, 4.0311), (120, -15, 4.545454545454545), (250, 70, 27.27272727272727), (33, 5, 1.2121212121212122)):
            r, n = ratios(ctx, W, H, az, el, radius=radius, reps=3)
            worst = max(worst, r)
        print(f"  config-4 shape {W}x{H}: max over four poses {worst:.4f}")
    ctx.close()


if __name__ == "__main__":
    main()
