"""Seeded synthetic models of the reference's network shapes: what bench.py, smoke(), the scripts and the tests build their
models with (host-only ABI calls: works without a GPU).  Lives in the package so that bench.py does not import from tests/."""
from __future__ import annotations

import numpy as np

import nerfhip as nh
import synthetic as syn


def build_model(log2_hashmap_size=19, H=128, cascade=1, bound=1.0, seed=1337, **cfg_kw):
    """Returns (desc, keepalive, config).  Works without a GPU (host-only ABI calls)."""
    cfg = syn.base_config(log2_hashmap_size=log2_hashmap_size, **cfg_kw)
    # first pass: level table needs only the encoding block + bound
    probe = dict(cfg)
    probe["snapshot"] = {"aabb": [-bound] * 3 + [bound] * 3, "bound": bound, "cascade": cascade,
                         "density_grid_size": H, "params": [0.0], "density_grid": [0.0]}
    d0, _ = nh.desc_from_config(probe)
    lt = nh.level_table(d0)
    n_grid = int(lt.offset[d0.n_levels]) * int(d0.n_features_per_level)
    cfg, params, grid = syn.make_scene(n_grid, seed=seed, H=H, cascade=cascade, bound=bound, config=cfg)
    desc, keep = nh.desc_from_config(cfg, params, grid)
    return desc, keep, cfg


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)
