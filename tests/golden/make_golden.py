#!/usr/bin/env python3
"""Generates tests/golden/tiny_scene.npz with the CPU oracle.

The reference holds no golden vectors for this path and cannot be run (CUDA-only), so these
fixtures pin the ORACLE (against accidental change) and the HIP path (against the oracle), not
the reference binary; SURVEY.md 8(c) "parity unpinned" still applies.
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import models  # noqa: E402
import oracle_py as op  # noqa: E402
import synthetic as syn  # noqa: E402

LOG2T, H, W, HH = 8, 16, 24, 16


def main():
    desc, keep, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=2024)
    o = op.Oracle(desc)
    rng = np.random.default_rng(99)
    pos01 = np.concatenate([rng.random((120, 3), dtype=np.float32),
                            np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.25]], np.float32)])
    d = rng.normal(size=(len(pos01), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d01 = (np.float32(0.5) * d + np.float32(0.5)).astype(np.float32)
    xyz = ((pos01 - np.float32(0.5)) * np.float32(2.0)).astype(np.float32)
    feat, dirf = o.encode_grid(pos01), o.encode_dir(d01)
    out4 = o.mlp_forward(feat, dirf)
    sigma, rgb = o.network(xyz, d)
    cam, pose = syn.default_camera(W, HH), syn.orbit_pose(40, 25)
    ro, rd, nr, fr = o.generate_rays(cam, pose, W, HH)
    xyzs, dirs, deltas = o.march(ro, rd, nr, fr, 4)
    rgba, depth, st = o.render(cam, pose, W, HH, schedule=op.SCHED_PER_RAY)
    np.savez_compressed(Path(__file__).with_name("tiny_scene.npz"),
                        params=keep[0], density_grid=keep[1].astype(np.uint8), pos01=pos01, dir=d, feat=feat, dirf=dirf,
                        out4=out4, sigma=sigma, rgb=rgb, cam=cam, pose=pose, rays_d=rd, nears=nr, fars=fr, xyzs=xyzs,
                        deltas=deltas, rgba=rgba, depth=depth, n_samples=np.int64(st.n_samples),
                        meta=np.array([LOG2T, H, W, HH, 2024], np.int64))
    print("wrote tiny_scene.npz:", st.n_samples, "samples")


# A second fixture for the shapes outside base.json (round 2): a Frequency-4 / Smoothstep / F = 4 x 8 levels / 32-neuron
# model with 2 + 1 hidden layers -- pins the oracle's generalised encoders and MLP, and through it the generic instance.
GENERIC_KW = dict(dir_otype="Frequency", n_frequencies=4, interpolation="Smoothstep", n_features_per_level=4, n_levels=8,
                  n_neurons=32, density_hidden_layers=2, rgb_hidden_layers=1)


def main_generic():
    desc, keep, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=2025, **GENERIC_KW)
    o = op.Oracle(desc)
    rng = np.random.default_rng(98)
    pos01 = np.concatenate([rng.random((90, 3), dtype=np.float32), np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.25]], np.float32)])
    d = rng.normal(size=(len(pos01), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d01 = (np.float32(0.5) * d + np.float32(0.5)).astype(np.float32)
    feat, dirf = o.encode_grid(pos01), o.encode_dir(d01)
    out4 = o.mlp_forward(feat, dirf)
    cam, pose = syn.default_camera(W, HH), syn.orbit_pose(130, 20)
    rgba, depth, st = o.render(cam, pose, W, HH, schedule=op.SCHED_PER_RAY)
    np.savez_compressed(Path(__file__).with_name("generic_scene.npz"),
                        params=keep[0], density_grid=keep[1].astype(np.uint8), pos01=pos01, dir=d, feat=feat, dirf=dirf,
                        out4=out4, cam=cam, pose=pose, rgba=rgba, depth=depth, n_samples=np.int64(st.n_samples),
                        meta=np.array([LOG2T, H, W, HH, 2025], np.int64))
    print("wrote generic_scene.npz:", st.n_samples, "samples, widths", o.feat_width, o.dir_width)


if __name__ == "__main__":
    main()
    main_generic()
