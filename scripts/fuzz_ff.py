"""Randomised equality run of the barrier fast-forward (nrf_device.h fast_forward_to_barrier / _pow2): NRF_MARCH_FF=0 (every
trip ahead of t_skip simulated) against the default, bit for bit, poisoned output planes.  Random camera positions (outside
the aabb, in every shell, inside the innermost cube), random view directions (every sign pattern), random dt_gamma (steps
at dt_min / growing / at dt_max), grids of 32 / 64 / 128 cells, 1-5 cascades with bounds 1-16, both schedulings.
usage: scripts/fuzz_ff.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models, synthetic as syn
import test_persistent_gpu as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
shapes = [(1, 1.0, 128), (1, 1.0, 64), (1, 1.0, 32), (1, 2.0, 64), (2, 2.0, 64), (3, 4.0, 32), (3, 4.0, 128), (4, 8.0, 64), (5, 16.0, 64), (5, 16.0, 128), (2, 4.0, 64)]
built = [models.build_model(log2_hashmap_size=13, H=H, cascade=C, bound=b) for C, b, H in shapes]


def random_pose(bound):
    """camera-to-world looking from a random point at a random target near the object (or anywhere)"""
    r = float(rng.choice([0.3, 0.8, 1.2, 1.9, 2.1, 3.9, 4.2, 8.5, 17.0, 45.0])) * (0.7 + 0.6 * rng.random())
    v = rng.normal(size=3); v /= np.linalg.norm(v)
    pos = v * r / 0.33
    target = rng.normal(size=3) * (0.3 if rng.random() < 0.8 else 0.5 * bound) / 0.33
    zc = pos - target; zc /= np.linalg.norm(zc)
    up = np.array([0.0, 0.0, 1.0]) if abs(zc[2]) < 0.95 else np.array([1.0, 0.0, 0.0])
    xc = np.cross(up, zc); xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = xc, yc, zc, pos
    return m.astype(np.float32)


bad = 0
rays = 0
for case in range(n_cases):
    k = int(rng.integers(0, len(shapes)))
    C, bound, gH = shapes[k]
    desc = built[k][0]
    W, H = int(rng.integers(40, 400)), int(rng.integers(30, 260))
    n = int(rng.integers(4, 24))
    poses = [random_pose(bound) for _ in range(n)]
    sched = T.PERSISTENT if rng.random() < 0.7 else T.STRIP
    kw = {"dt_gamma": float(rng.choice([0.0, 1.0 / 256.0, 1.0 / 128.0, 1.0 / 64.0, 1.0 / 16.0, rng.random() / 50.0])), "max_steps": 1024}
    try:
        ref = T._render(desc, W, H, poses, dict(sched, NRF_MARCH_FF="0"), opts_kw=kw)
        got = T._render(desc, W, H, poses, sched, opts_kw=kw)
        rays += ref[3]
        same_px = np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
        if not same_px or got[2:] != ref[2:]:
            d = np.abs(got[0] - ref[0])
            print("DIFF", "shape", shapes[k], W, H, n, kw, "pixels equal", same_px, "max|d|", float(np.nanmax(d)), "differing px", int((d.max(axis=-1) > 0).sum()),
                  "samples/rays", got[2:], ref[2:], flush=True)
            bad += 1
    except Exception as e:
        bad += 1
        print("ERROR", shapes[k], W, H, n, kw, str(e)[:300], flush=True)
    if case % 20 == 19:
        print(f"  .. {case + 1} cases, {rays / 1e6:.1f} M rays, {bad} mismatches", flush=True)
print(f"{n_cases} random cases ({rays / 1e6:.1f} M rays), {bad} mismatches", flush=True)
