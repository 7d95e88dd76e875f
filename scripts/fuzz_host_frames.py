#!/usr/bin/env python3
"""Randomised equality run of the host end of render_frame: nrf_submit_host_u8 / nrf_wait_host_u8 (the kernel writes the 8-bit
Image, finished strip rows are copied while the launch renders, the calling thread fills the background rows) against
nrf_render + nrf_read_u8, byte for byte.  ONE context per model lives through all of its cases, so the slots' background
bookkeeping sees every sequence of frame sizes, view counts, cameras (outside, inside, far, looking away), background
colours, rgb-only flags and one / two calls in flight.   usage: scripts/fuzz_host_frames.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models
import nerfhip as nh
import synthetic as syn
import test_persistent_gpu as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
built = [models.build_model(log2_hashmap_size=14, H=64), models.build_model(log2_hashmap_size=14, H=32, cascade=3, bound=4.0),
         models.build_model(log2_hashmap_size=13, H=32, dir_otype="Frequency", n_frequencies=12),   # wide instance
         models.build_model(log2_hashmap_size=13, H=32, n_neurons=32),                               # width instance
         models.build_model(log2_hashmap_size=13, H=32, sh_degree=6),                                # wide-SH instance
         models.build_model(log2_hashmap_size=13, H=32, n_features_per_level=4, n_levels=8)]         # generic instance
ctxs, refs = [], []
for b in built:
    c, r = nh.NerfHip(0), nh.NerfHip(0)
    c.load_model(b[0])
    r.load_model(b[0])
    ctxs.append(c)
    refs.append(r)
bad = 0
pending = [None] * len(built)  # (ticket, expected frames, rgb_only) of a call still in flight
sizes = [None] * len(built)


def expect(mi, W, H, cams, poses, opts):
    r = refs[mi]
    r.set_options(opts)
    r.set_resolution(W, H)
    out = []
    for cm, p in zip(cams, poses):
        r.render(cm, p)
        out.append(r.read_u8())
    return out


def check(mi, got, want, rgb_only, what):
    global bad
    rgb, depth = got
    ok = all(np.array_equal(rgb[v], want[v][0]) for v in range(len(want)))
    if not rgb_only:
        ok = ok and all(np.array_equal(depth[v], want[v][1]) for v in range(len(want)))
    else:
        ok = ok and depth is None
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)


for case in range(n_cases):
    mi = int(rng.integers(0, len(built)))
    c = ctxs[mi]
    if sizes[mi] is None or rng.random() < 0.25:
        if pending[mi] is not None:  # a new resolution drops the slots: finish the call in flight first
            t, want, ro, what = pending[mi]
            check(mi, c.wait_host_u8(t), want, ro, what)
            pending[mi] = None
        W, H = (int(rng.integers(1, 500)), int(rng.integers(1, 360))) if rng.random() < 0.8 else (int(rng.integers(1, 120)) * 4, int(rng.integers(1, 50)) * 8)
        sizes[mi] = (W, H)
        c.set_resolution(W, H)
    W, H = sizes[mi]
    n = int(rng.integers(1, 7))
    opts = nh.default_options()
    opts.bg_color = float(rng.choice([1.0, 1.0, 1.0, 0.0, 0.37]))
    if pending[mi] is None:
        c.set_options(opts)
    else:
        opts = c_opts[mi]
    poses = [T._poses(str(rng.choice(["orbit", "orbit", "inside", "away", "far"])), 3)[int(rng.integers(0, 3))] for _ in range(n)]
    cam = syn.default_camera(W, H)
    cams = [cam * np.float32(rng.uniform(0.8, 1.3)) for _ in range(n)]
    rgb_only = bool(rng.random() < 0.3)
    want = expect(mi, W, H, cams, poses, opts)
    what = (case, mi, W, H, n, rgb_only, opts.bg_color)
    t = c.submit_host_u8(cams, poses, flags=nh.NRF_HOST_RGB_ONLY if rgb_only else 0)
    if pending[mi] is not None:  # two calls in flight: the older one is waited for after the newer one was submitted
        t0, want0, ro0, what0 = pending[mi]
        check(mi, c.wait_host_u8(t0), want0, ro0, what0)
        pending[mi] = None
    if rng.random() < 0.5:
        check(mi, c.wait_host_u8(t, copy=bool(rng.random() < 0.5)), want, rgb_only, what)
    else:
        pending[mi] = (t, want, rgb_only, what)
        c_opts = globals().setdefault("c_opts", {})
        c_opts[mi] = opts
for mi, p in enumerate(pending):
    if p is not None:
        check(mi, ctxs[mi].wait_host_u8(p[0]), p[1], p[2], p[3])
print(f"{n_cases} random host-frame cases, {bad} mismatches")
sys.exit(1 if bad else 0)
