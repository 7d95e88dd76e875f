#!/usr/bin/env python3
"""Generates tests/golden/contract_modes.npz with the CPU oracle in its FMA-contraction mode (nrfo_set_contract).

nvcc contracts `a * b + c` into one fused multiply-add by default and the reference's build sets no -fmad=false
(R/CMakeLists.txt:71-79); the arithmetic contract of this repository (oracle default and HIP path) rounds every operation.
`nrfo_set_contract(m, 1)` evaluates the expressions the reference's device source writes as `a * b + c` with fmaf (the
list is in oracle/nerf_oracle.h); this fixture pins that mode -- stage outputs of the tiny scene and whole frames with
per-ray sample counts -- so that the measured distance "no contraction vs nvcc's contraction" (bench.py
`parity.vs_fma_contract`) refers to something that cannot drift.  Made by the oracle, not by the reference binary
(CUDA-only): parity stays unpinned.
Run from the repo root:  python tests/golden/make_contract_golden.py
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

FW, FH = 96, 64  # the config-2 model's small frame


def main():
    G = np.load(Path(__file__).with_name("tiny_scene.npz"))
    LOG2T, H, W, HH, SEED = [int(v) for v in G["meta"]]
    _, _, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    desc, keep = nh.desc_from_config(cfg, G["params"], G["density_grid"].astype(np.float32))
    o = op.Oracle(desc, contract=True)
    out = {}
    out["tiny_feat"] = o.encode_grid(G["pos01"])
    out["tiny_dirf"] = o.encode_dir((G["dir"] * np.float32(0.5) + np.float32(0.5)).astype(np.float32))
    ro, rd, nr, fr = o.generate_rays(G["cam"], G["pose"], W, HH)
    out["tiny_rays_d"], out["tiny_nears"], out["tiny_fars"] = rd, nr, fr
    xyzs, dirs, deltas = o.march(ro, rd, nr, fr, 4)
    out["tiny_xyzs"], out["tiny_deltas"] = xyzs, deltas
    rgba, depth, st, counts, hashes = o.render_rays(G["cam"], G["pose"], W, HH)
    out["tiny_rgba"], out["tiny_depth"], out["tiny_counts"], out["tiny_n"] = rgba, depth, counts, np.int64(st.n_samples)
    # BASELINE config 2's model (T = 2^19, 64-wide MLPs, SH-4)
    desc2, keep2, _ = models.build_model(log2_hashmap_size=19, H=128)
    cam, pose = syn.default_camera(FW, FH), syn.orbit_pose(30, 30)
    out["c2_cam"], out["c2_pose"] = cam, pose
    rgba, depth, st, counts, hashes = op.Oracle(desc2, contract=True).render_rays(cam, pose, FW, FH)
    out["c2_rgba"], out["c2_depth"], out["c2_counts"], out["c2_hashes"], out["c2_n"] = rgba, depth, counts, hashes, np.int64(st.n_samples)
    np.savez_compressed(Path(__file__).with_name("contract_modes.npz"), **out)
    print("wrote contract_modes.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
