"""f3: instant-ngp's own snapshot layout (`snapshot.nerf`, `params_binary`, Morton-ordered `density_grid_binary`) and
`transforms.json` camera paths, in both loaders (Python: nerfhip.desc_from_config; C++: NerfRender::load_snapshot).
The reference reads neither (it consumes only its array form, nerf_render.cu:441-453); the layout follows instant-ngp's
public Testbed::save_snapshot / NerfNetwork and is documented in nerfhip.py "instant-ngp snapshots".  Proof obligation:
a model written in that layout loads to exactly the model of its array-form snapshot -- every parameter, every grid
cell, every derived hyper-parameter -- and renders the same bytes."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

import models
import nerfhip as nh
import synthetic as syn

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "nerf-cuda_amd" / "host"


def _info(path):
    r = subprocess.run([str(HOST / "snapshot_info"), str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def _pair(tmp_path, aabb_scale, log2T=12, H=32):
    """(array-form snapshot, ngp-form snapshot, desc of the array form) of one model with bound = aabb_scale / 2."""
    bound = aabb_scale / 2.0
    cascade = 1 if aabb_scale == 1 else aabb_scale.bit_length() - 1
    pls = nh.default_per_level_scale(float(aabb_scale), 16, 16)  # instant-ngp derives it from aabb_scale, not from bound
    desc, keep, cfg = models.build_model(log2_hashmap_size=log2T, H=H, bound=bound, cascade=cascade, per_level_scale=pls)
    # instant-ngp's cascades are nested consistently (a coarse cell is at least the maximum of its children), and the
    # loader enforces that by max-pooling: give the array form the same property, so that both files describe one grid
    g = keep[1].reshape(cascade, H, H, H)
    q = H // 4
    for k in range(1, cascade):
        inner = g[k][q:q + H // 2, q:q + H // 2, q:q + H // 2]
        np.maximum(inner, g[k - 1].reshape(H // 2, 2, H // 2, 2, H // 2, 2).max(axis=(1, 3, 5)), out=inner)
    arr, ngp = tmp_path / f"array_{aabb_scale}.msgpack", tmp_path / f"ngp_{aabb_scale}.msgpack"
    syn.write_snapshot(arr, cfg, keep[0], keep[1], binary="__half")
    syn.write_ngp_snapshot(ngp, cfg, keep[0], keep[1], aabb_scale)
    return arr, ngp, desc, keep, cfg


@pytest.mark.parametrize("aabb_scale", [1, 4, 32])
def test_ngp_layout_loads_to_the_array_form_model_in_both_loaders(tmp_path, aabb_scale):
    arr, ngp, desc, keep, cfg = _pair(tmp_path, aabb_scale)
    a, b = _info(arr), _info(ngp)  # the C++ loader on both files
    for k in a:
        if k == "mean_density":
            continue  # array form: the generator's grid mean; ngp form: mean over instant-ngp's (empty) cascade 0
        assert a[k] == b[k], (k, a[k], b[k])
    assert b["bound"] == aabb_scale / 2.0 and b["cascade"] == desc.cascade and b["n_params"] == b["expected"]
    cfg_ngp = syn.read_snapshot(ngp)
    assert nh.is_ngp_snapshot(cfg_ngp) and not nh.is_ngp_snapshot(syn.read_snapshot(arr))
    d2, k2 = nh.desc_from_config(cfg_ngp)  # the Python loader
    d1, k1 = nh.desc_from_config(syn.read_snapshot(arr))
    np.testing.assert_array_equal(k2[0], k1[0])
    np.testing.assert_array_equal(k2[1], k1[1])
    np.testing.assert_array_equal(k2[1], keep[1])  # Morton -> x-major came back exactly
    for f in ("bound", "scale", "cascade", "density_grid_size", "per_level_scale", "n_levels", "base_resolution", "n_params",
              "rgb_output_activation"):
        assert getattr(d1, f) == getattr(d2, f), f
    assert list(d2.aabb) == list(d1.aabb)
    w = (np.arange(k2[0].size, dtype=np.float64) % 97) + 1
    assert abs(b["psum"] - float((k2[0].astype(np.float64) * w).sum())) <= 1e-9 * abs(b["psum"])  # C++ == Python, element by element


def test_ngp_cascades_are_max_pooled_like_instant_ngp_bitfield(tmp_path):
    """Reference cascade k = instant-ngp cascade k + 1, each cell raised to the maximum of its eight children in the
    finer instant-ngp cascade (a cell instant-ngp trained as occupied at the fine level must not vanish), mean_density
    from instant-ngp's cascade 0, logistic colours as the rgb output activation, fp32 grid blobs told from fp16 by size."""
    import msgpack

    H, aabb_scale = 8, 2
    rng = np.random.default_rng(3)
    ngp = rng.random((2, H, H, H)).astype(np.float16).astype(np.float32)  # [cascade][x][y][z]
    ax = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    m = nh.morton3d(X, Y, Z).reshape(-1)
    assert sorted(m.tolist()) == list(range(H ** 3))
    blob = np.zeros((2, H ** 3), np.float32)
    for c in range(2):
        blob[c, m] = ngp[c].reshape(-1)
    desc, keep, cfg = models.build_model(log2_hashmap_size=10, H=H, bound=1.0)
    want = ngp[1].copy()
    q = H // 4
    want[q:q + H // 2, q:q + H // 2, q:q + H // 2] = np.maximum(want[q:q + H // 2, q:q + H // 2, q:q + H // 2],
                                                                  ngp[0].reshape(H // 2, 2, H // 2, 2, H // 2, 2).max(axis=(1, 3, 5)))
    for dtype in (np.float16, np.float32):
        c = {k: v for k, v in cfg.items() if k != "snapshot"}
        c["encoding"] = dict(cfg["encoding"], per_level_scale=float(desc.per_level_scale))
        c["snapshot"] = {"params_binary": keep[0].astype(np.float16).tobytes(), "params_type": "__half", "density_grid_size": H,
                         "density_grid_binary": blob.astype(dtype).tobytes(), "nerf": {"aabb_scale": aabb_scale},
                         "aabb": {"min": [-0.5] * 3, "max": [1.5] * 3}}
        f = tmp_path / f"pool_{np.dtype(dtype).name}.msgpack"
        f.write_bytes(msgpack.packb(c, use_single_float=True, use_bin_type=True))
        d, k = nh.desc_from_config(syn.read_snapshot(f))
        np.testing.assert_array_equal(k[1].reshape(H, H, H), want)
        assert d.cascade == 1 and d.bound == 1.0 and list(d.aabb) == [-1.0] * 3 + [1.0] * 3
        assert d.mean_density == pytest.approx(float(ngp[0].mean()), rel=1e-6)
        assert d.rgb_output_activation == nh.ACT["sigmoid"]  # instant-ngp's logistic (snapshot.nerf.rgb_activation default)
        info = _info(f)
        v = (np.arange(want.size, dtype=np.float64) % 89) + 1
        assert abs(info["gsum"] - float((want.reshape(-1).astype(np.float64) * v).sum())) <= 1e-9 * abs(info["gsum"])
        assert info["roa"] == nh.ACT["sigmoid"] and info["mean_density"] == pytest.approx(d.mean_density, rel=1e-6)


def test_transforms_json_round_trip_and_intrinsics(tmp_path):
    poses = [syn.orbit_pose(30.0 * i, 20.0) for i in range(5)]
    f = tmp_path / "transforms.json"
    syn.write_transforms_json(f, poses, 800, 800)
    cams, got, W, H = syn.load_transforms_json(f)
    assert (W, H) == (800, 800) and got.shape == (5, 4, 4)
    np.testing.assert_allclose(got, np.stack(poses), rtol=0, atol=1e-7)
    np.testing.assert_allclose(cams[0], [1111.1110311937682, 1111.1110311937682, 400, 400], rtol=1e-6)  # Blender-synthetic focal
    cams2, _, W2, H2 = syn.load_transforms_json(f, 200, 100)  # rescaled to another resolution
    np.testing.assert_allclose(cams2[0], [0.5 * 200 / np.tan(0.5 * 0.6911112070083618)] * 2 + [100, 50], rtol=1e-6)


@pytest.mark.gpu
def test_ngp_snapshot_renders_the_array_form_bytes_and_the_camera_path(tmp_path):
    """GPU leg: C++ testbed on the instant-ngp-layout file == on the array form, bit for bit; every frame of a
    transforms.json path == the Python binding's render of that pose from the Python-loaded ngp snapshot."""
    arr, ngp, desc, keep, cfg = _pair(tmp_path, 4)
    W, H = 96, 64
    poses = [syn.orbit_pose(50.0 * i, 15.0 + 10 * i, radius=3.0) for i in range(4)]
    tj = tmp_path / "transforms.json"
    syn.write_transforms_json(tj, poses, W, H)
    outs = {}
    for name, snap in (("arr", arr), ("ngp", ngp)):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run([str(HOST / "testbed"), str(snap), str(W), str(H), str(out) + "/", str(tj)], capture_output=True,
                           text=True, timeout=180)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "camera path: 4 frames" in r.stdout
        outs[name] = [np.fromfile(out / "image.rgb", np.uint8)] + [np.fromfile(out / f"path_{i:04d}.rgb", np.uint8) for i in range(4)]
    for a, b in zip(outs["arr"], outs["ngp"]):
        np.testing.assert_array_equal(a, b)
    assert outs["ngp"][1].min() < 250
    d2, k2 = nh.desc_from_config(syn.read_snapshot(ngp))
    cams, got_poses, _, _ = syn.load_transforms_json(tj)
    ctx = nh.NerfHip(0)
    ctx.load_model(d2)
    ctx.set_resolution(W, H)
    for i in range(4):
        ctx.render(cams[i], got_poses[i])
        np.testing.assert_array_equal(ctx.read_u8()[0].reshape(-1), outs["ngp"][1 + i])
    ctx.close()


def test_params_blob_as_tcnn_trainer_serializes_it(tmp_path):
    """The one part of instant-ngp's layout that the REFERENCE TREE itself pins: instant-ngp's snapshot block starts as
    tcnn's `Trainer::serialize()` -- `{"n_params": n, "params_binary": <sizeof(PARAMS_T) * n bytes of the inference
    parameters>}` (T/include/tiny-cuda-nn/trainer.h:267-279; PARAMS_T = __half for a FullyFusedMLP model, common.h:56-72), written
    as a nlohmann binary_t (msgpack `bin`) by `gpu_memory_to_json_binary` and read back either as that or as the JSON object
    `{"bytes": [...], "subtype": ...}` (gpu_memory_json.h:37-72) -- in the network's own parameter order, which is the
    reference's `set_params` order (nerf_network.h:273-291).  A file holding exactly those two keys (no `params_type`: tcnn
    v1.6 writes none) loads to the array-form model in both loaders, in both binary forms; a blob that disagrees with
    `n_params` is refused."""
    import msgpack
    arr, ngp, desc, keep, cfg = _pair(tmp_path, 1)
    base = syn.read_snapshot(ngp)
    params16 = np.asarray(keep[0], np.float32).astype(np.float16)
    want = nh.desc_from_config(syn.read_snapshot(arr))[1][0]
    for form in ("bin", "object"):
        c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in base.items()}
        snap = dict(c["snapshot"])
        snap.pop("params_type", None)
        snap["n_params"] = int(params16.size)                                   # trainer.h:271
        blob = params16.tobytes()                                               # trainer.h:272: sizeof(PARAMS_T) * n_params bytes
        snap["params_binary"] = blob if form == "bin" else {"bytes": list(blob), "subtype": None}  # gpu_memory_json.h:59-66
        c["snapshot"] = snap
        d, k = nh.desc_from_config(c)
        np.testing.assert_array_equal(k[0], want)
        assert d.n_params == params16.size == nh.expected_n_params(d)
        f = tmp_path / f"tcnn_{form}.msgpack"
        f.write_bytes(msgpack.packb(c, use_single_float=True, use_bin_type=True))
        info = _info(f)                                                         # the C++ loader
        assert info["n_params"] == info["expected"] == int(params16.size)
        assert info == {**_info(ngp), **{k2: info[k2] for k2 in info if k2 not in _info(ngp)}}
    bad = {k: (dict(v) if isinstance(v, dict) else v) for k, v in base.items()}
    bad["snapshot"] = dict(bad["snapshot"], n_params=int(params16.size) - 8)
    with pytest.raises(RuntimeError, match="n_params"):
        nh.desc_from_config(bad)
    f = tmp_path / "bad.msgpack"
    f.write_bytes(msgpack.packb(bad, use_single_float=True, use_bin_type=True))
    r = subprocess.run([str(HOST / "snapshot_info"), str(f)], capture_output=True, text=True)
    assert r.returncode != 0 and "n_params" in r.stderr
