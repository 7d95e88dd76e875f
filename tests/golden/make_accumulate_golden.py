#!/usr/bin/env python3
"""Generates tests/golden/accumulate_modes.npz with the CPU oracle's MLP accumulate modes.

The reference accumulates its MLP products in fp16 WMMA fragments (T/src/fully_fused_mlp.cu:69,334,437); the arithmetic
contract of this repository (oracle default and HIP path) accumulates in fp32.  `nrfo_set_mlp_accumulate` emulates the
reference's accumulator (fp16 running sum, rounded after every block of n products); this fixture pins those emulations --
network outputs of the tiny scene's inputs and whole frames -- so that the measured distance "fp32 accumulate vs the
reference's fp16 accumulate" (bench.py `parity.vs_fp16_accumulate`) refers to something that cannot drift.
Like every fixture here it is made by the oracle, not by the reference binary (CUDA-only): parity stays unpinned.
Run from the repo root:  python tests/golden/make_accumulate_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import models  # noqa: E402
import nerfhip as nh  # noqa: E402
import oracle_py as op  # noqa: E402
import synthetic as syn  # noqa: E402

MODES = (op.ACC_FP16_K16, op.ACC_FP16_K8, op.ACC_FP16_K4, op.ACC_FP16_STEP)
FW = FH = 64  # the config-2 model's small frame


def main():
    G = np.load(Path(__file__).with_name("tiny_scene.npz"))
    LOG2T, H, W, HH, SEED = [int(v) for v in G["meta"]]
    _, _, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    desc, keep = nh.desc_from_config(cfg, G["params"], G["density_grid"].astype(np.float32))
    out = {"modes": np.array(MODES, np.int64)}
    for mode in MODES:
        o = op.Oracle(desc, accumulate=mode)
        out[f"tiny_out4_{mode}"] = o.mlp_forward(G["feat"], G["dirf"])
        rgba, depth, st = o.render(G["cam"], G["pose"], W, HH, schedule=op.SCHED_PER_RAY)
        out[f"tiny_rgba_{mode}"], out[f"tiny_depth_{mode}"] = rgba, depth
        out[f"tiny_n_{mode}"] = np.int64(st.n_samples)
    # BASELINE config 2's model (T = 2^19, 64-wide MLPs, SH-4) on a 64x64 frame: the reference granularity and the bound
    desc2, keep2, _ = models.build_model(log2_hashmap_size=19, H=128)
    cam, pose = syn.default_camera(FW, FH), syn.orbit_pose(30, 30)
    out["c2_cam"], out["c2_pose"] = cam, pose
    for mode in (op.ACC_FP16_K16, op.ACC_FP16_STEP):
        rgba, depth, st = op.Oracle(desc2, accumulate=mode).render(cam, pose, FW, FH, schedule=op.SCHED_PER_RAY)
        out[f"c2_rgba_{mode}"], out[f"c2_depth_{mode}"], out[f"c2_n_{mode}"] = rgba, depth, np.int64(st.n_samples)
    np.savez_compressed(Path(__file__).with_name("accumulate_modes.npz"), **out)
    print("wrote accumulate_modes.npz")


if __name__ == "__main__":
    main()
