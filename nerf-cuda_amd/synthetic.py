"""Deterministic synthetic snapshots and cameras for tests and bench.py.

The reference ships neither weights (`freality.msgpack` is absent) nor a
dataset, and there is no network, so every measurement in this repository runs
on a seeded synthetic scene in the reference's own snapshot format
(reference src/nerf_render.cu:431-473,
include/nerf-cuda/nerf_network.h:424-433):

  * density grid: analytic "Lego-like" occupancy (union of boxes, cylinders
    and a sphere inside +-0.5 of the unit scene), value 1 inside, 0 outside;
  * hash table: U(-0.5, 0.5) (tcnn's +-1e-4 init would give a blank scene);
  * MLP weights: seeded normal weights, with the density row and the three
    colour rows made positive and rescaled so that sigma = exp(g0) has a
    median of ~36 (rays saturate after ~10-25 samples, like a trained solid)
    and rgb lands in [0, 1].

Cameras follow the Blender-synthetic convention the reference's `main` uses
(camera-to-world, OpenGL axes, radius 4.0311, scene scale 0.33).
"""
from __future__ import annotations

import math

import numpy as np

# n_params layout (reference nerf_network.h:273-291; fully_fused_mlp.cu:636-687)
N_D0, N_D1, N_R0, N_R1, N_R2 = 64 * 32, 16 * 64, 64 * 32, 64 * 64, 16 * 64
N_MLP = N_D0 + N_D1 + N_R0 + N_R1 + N_R2  # 10240


def base_config(log2_hashmap_size=19, n_levels=16, base_resolution=16, sh_degree=4, rgb_output_activation="None",
                dir_otype="SphericalHarmonics", n_frequencies=2, grid_type=None, per_level_scale=None,
                n_features_per_level=2, interpolation=None, n_neurons=64, density_hidden_layers=1, rgb_hidden_layers=2,
                activation="ReLU", density_output_activation="None", sigma_activation=None, density_n_output=None):
    """The four network blocks of reference configs/nerf/base.json (defaults), or any other shape the
    reference's JSON vocabulary can describe."""
    if dir_otype == "SphericalHarmonics":
        dir_nested = {"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": sh_degree}
    elif dir_otype == "Frequency":
        dir_nested = {"n_dims_to_encode": 3, "otype": "Frequency", "n_frequencies": n_frequencies}
    else:
        dir_nested = {"n_dims_to_encode": 3, "otype": "Identity"}
    encoding = {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
                "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution}
    if grid_type:  # "Hash" (default) | "Dense" | "Tiled": tcnn GridEncoding `type` (grid.h:1365-1386)
        encoding["type"] = grid_type
    if per_level_scale:
        encoding["per_level_scale"] = float(per_level_scale)
    if interpolation:  # "Linear" (default) | "Nearest" | "Smoothstep"
        encoding["interpolation"] = interpolation
    network = {"otype": "FullyFusedMLP", "activation": activation, "output_activation": density_output_activation,
               "n_neurons": n_neurons, "n_hidden_layers": density_hidden_layers}
    if sigma_activation:
        network["sigma_activation"] = sigma_activation
    if density_n_output:
        network["n_output_dims"] = int(density_n_output)
    return {
        "encoding": encoding,
        "network": network,
        "dir_encoding": {"otype": "Composite", "nested": [dir_nested, {"otype": "Identity", "n_bins": 4, "degree": 4}]},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": activation, "output_activation": rgb_output_activation,
                        "n_neurons": n_neurons, "n_hidden_layers": rgb_hidden_layers},
    }


def network_shape(config):
    """(feat_raw, feat_w, width, density_hidden, rgb_hidden, dir_raw, dir_w) of a config's four blocks
    (widths as NerfNetwork derives them, reference nerf_network.h:95-135)."""
    enc, net, rgb = config["encoding"], config["network"], config["rgb_network"]
    feat_raw = int(enc.get("n_levels", 16)) * int(enc.get("n_features_per_level", 2))
    nested = config["dir_encoding"]["nested"][0]
    ot = nested["otype"].lower()
    dir_raw = {"sphericalharmonics": int(nested.get("degree", 4)) ** 2, "frequency": 6 * int(nested.get("n_frequencies", 12)),
               "identity": 3}[ot]
    pad16 = lambda v: (v + 15) // 16 * 16  # noqa: E731
    return (feat_raw, pad16(feat_raw), int(net.get("n_neurons", 128)), int(net.get("n_hidden_layers", 5)),
            int(rgb.get("n_hidden_layers", 5)), dir_raw, pad16(dir_raw))


def _occupancy(x, y, z):
    """Analytic occupancy in NGP coordinates (y is up)."""
    def box(x0, x1, y0, y1, z0, z1):
        return (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1) & (z >= z0) & (z <= z1)

    occ = box(-0.42, 0.42, -0.22, -0.10, -0.25, 0.25)       # chassis slab
    occ |= box(-0.20, 0.15, -0.10, 0.15, -0.18, 0.18)       # cabin
    occ |= box(0.42, 0.50, -0.25, 0.00, -0.30, 0.30)        # blade
    occ |= box(-0.36, -0.30, -0.10, 0.26, -0.03, 0.03)      # exhaust / mast
    for cx in (-0.25, 0.25):                                 # wheels: cylinders along z
        r2 = (x - cx) ** 2 + (y + 0.25) ** 2
        occ |= (r2 <= 0.1 ** 2) & (np.abs(z) >= 0.2) & (np.abs(z) <= 0.3)
    occ |= ((x + 0.05) ** 2 + (y - 0.22) ** 2 + z ** 2) <= 0.09 ** 2  # beacon sphere
    return occ


def density_grid(H=128, cascade=1, bound=1.0):
    """Float grid [C*H^3], index level*H^3 + x*H^2 + y*H + z (reference nerf_render.h:64-65)."""
    out = np.zeros((cascade, H, H, H), np.float32)
    for c in range(cascade):
        mip_bound = min(float(2 ** c), float(bound))
        centers = ((np.arange(H, dtype=np.float64) + 0.5) / H * 2.0 - 1.0) * mip_bound
        half = mip_bound / H
        occ = np.zeros((H, H, H), bool)
        # conservative: a cell is occupied if its centre or any corner is inside
        for ox in (-half, 0.0, half):
            for oy in (-half, 0.0, half):
                for oz in (-half, 0.0, half):
                    X, Y, Z = np.meshgrid(centers + ox, centers + oy, centers + oz, indexing="ij", sparse=True)
                    occ |= _occupancy(X, Y, Z)
        out[c][occ] = 1.0
    return out.reshape(-1)


def _simulate_features(rng, n, table_amp=0.5):
    """Marginal distribution of hash-grid features: sum_8 w_c v_c, v ~ U(+-amp), trilinear w."""
    f = rng.random((n, 32, 3))
    v = rng.uniform(-table_amp, table_amp, (n, 32, 8))
    w = np.ones((n, 32, 8))
    for c in range(8):
        for d in range(3):
            w[:, :, c] *= np.where((c >> d) & 1, f[:, :, d], 1.0 - f[:, :, d])
    return (w * v).sum(-1).astype(np.float32)


def _sh4(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return np.stack([
        np.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], axis=1).astype(np.float32)


def mlp_weights(seed=1337, sigma_log_median=3.6, rgb_mean=0.5):
    """The 10240 MLP parameters, calibrated on simulated inputs (fp32 numpy)."""
    rng = np.random.default_rng(seed)
    D0 = rng.normal(0.0, 0.5, (64, 32)).astype(np.float32)
    D1 = rng.normal(0.0, 0.3, (16, 64)).astype(np.float32)
    D1[0] = rng.uniform(0.0, 1.0, 64)                      # density row: positive -> g0 > 0
    R0 = rng.normal(0.0, 0.25, (64, 32)).astype(np.float32)
    R1 = rng.normal(0.0, 0.2, (64, 64)).astype(np.float32)
    R2 = rng.normal(0.0, 0.2, (16, 64)).astype(np.float32)
    R2[:3] = np.abs(rng.normal(0.0, 0.2, (3, 64)))         # colour rows: positive
    feat = _simulate_features(rng, 4096)
    h = np.maximum(feat @ D0.T, 0.0)
    D1[0] *= sigma_log_median / float((h @ D1[0]).mean())
    g = h @ D1.T
    dirs = rng.normal(size=(4096, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rin = np.concatenate([g, _sh4(dirs)], axis=1).astype(np.float32)
    h2 = np.maximum(np.maximum(rin @ R0.T, 0.0) @ R1.T, 0.0)
    for c in range(3):
        R2[c] *= rgb_mean / float((h2 @ R2[c]).mean())
    return np.concatenate([m.reshape(-1) for m in (D0, D1, R0, R1, R2)]).astype(np.float32)


def mlp_weights_general(shape, seed=1337, sigma_log_median=3.6, rgb_mean=0.5):
    """MLP parameters for any network shape (see network_shape), calibrated like mlp_weights: sigma = exp(g0) with a
    median of ~36 and rgb around 0.5 on simulated inputs.  Order: density first | hidden... | last, rgb likewise
    (reference nerf_network.h:273-291), each matrix row-major [out][in]."""
    feat_raw, feat_w, width, dens_hidden, rgb_hidden, dir_raw, dir_w = shape
    rng = np.random.default_rng(seed)

    def matrices(n_in, hidden, first_std, hidden_std, last_std):
        dims = [n_in] + [width] * hidden + [16]
        out = []
        for i in range(len(dims) - 1):
            std = (first_std * math.sqrt(32.0 / dims[i]) if i == 0 else
                   (last_std if i == len(dims) - 2 else hidden_std) * math.sqrt(64.0 / dims[i]))
            out.append(rng.normal(0.0, std, (dims[i + 1], dims[i])).astype(np.float32))
        return out

    def forward(mats, x):
        for m in mats[:-1]:
            x = np.maximum(x @ m.T, 0.0)
        return x

    D = matrices(feat_w, dens_hidden, 0.5, 0.25, 0.3)
    D[-1][0] = rng.uniform(0.0, 1.0, width)                     # density row: positive -> g0 > 0
    R = matrices(16 + dir_w, rgb_hidden, 0.25, 0.2, 0.2)
    R[-1][:3] = np.abs(rng.normal(0.0, 0.2, (3, width)))        # colour rows: positive
    feat = np.zeros((2048, feat_w), np.float32)
    feat[:, :feat_raw] = _simulate_features(rng, 2048)[:, np.arange(feat_raw) % 32]
    h = forward(D, feat)
    D[-1][0] *= sigma_log_median / max(float((h @ D[-1][0]).mean()), 1e-3)
    g = h @ D[-1].T
    dirf = np.ones((2048, dir_w), np.float32)
    dirf[:, :dir_raw] = rng.uniform(-1.0, 1.0, (2048, dir_raw))
    h2 = forward(R, np.concatenate([g, dirf], axis=1).astype(np.float32))
    for c in range(3):
        R[-1][c] *= rgb_mean / max(float((h2 @ R[-1][c]).mean()), 1e-3)
    return np.concatenate([m.reshape(-1) for m in D + R]).astype(np.float32)


BASE_SHAPE = (32, 32, 64, 1, 2, 16, 16)


def make_scene(n_grid_params, seed=1337, H=128, cascade=1, bound=1.0, scale=0.33, config=None):
    """Returns (config dict with a `snapshot` block but without the two big arrays,
    params float32 [n_params], density_grid float32 [C*H^3]).  `n_grid_params` is the
    number of hash-table values (F * total entries), which depends on the level table."""
    cfg = dict(config or base_config())
    rng = np.random.default_rng(seed + 1)
    table = rng.uniform(-0.5, 0.5, n_grid_params).astype(np.float32)
    shape = network_shape(cfg)
    mlp = mlp_weights(seed) if shape[:5] == BASE_SHAPE[:5] and shape[6] == 16 else mlp_weights_general(shape, seed)
    params = np.concatenate([mlp, table])
    grid = density_grid(H, cascade, bound)
    cfg["snapshot"] = {
        "aabb": [-bound, -bound, -bound, bound, bound, bound],
        "bound": float(bound), "scale": float(scale), "cascade": int(cascade), "density_grid_size": int(H),
        "mean_density": float(grid.mean()),
    }
    return cfg, params, grid


# --------------------------------------------------------------------------- cameras
def orbit_pose(azimuth_deg, elevation_deg=30.0, radius=4.0311):
    """Blender/NeRF camera-to-world (OpenGL axes: x right, y up, z backward), z-up world,
    looking at the origin.  Row-major 4x4."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    pos = np.array([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    zc = pos / np.linalg.norm(pos)                       # camera backward axis
    xc = np.cross(np.array([0.0, 0.0, 1.0]), zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    m = np.eye(4, dtype=np.float64)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = xc, yc, zc, pos
    return m.astype(np.float32)


def default_camera(width, height, fov_x=0.6911112070083618):
    """{fl_x, fl_y, cx, cy} (reference common.h:68-74).  Blender-synthetic field of view applied
    to the SHORTER image side so the object stays in frame at 16:9."""
    fl = 0.5 * min(width, height) / math.tan(0.5 * fov_x)
    return np.array([fl, fl, width * 0.5, height * 0.5], np.float32)


REFERENCE_MAIN_POSE = np.array([  # the hard-coded pose of reference src/main.cu:152-155 / render_server.cu:53-56
    [-0.5575427361517304, -0.11682263918046752, 0.8218871992959822, 3.9673954052389253],
    [0.8300327085486383, -0.094966079921629, 0.5495699649760266, 2.667431152445114],
    [0.013849191732089516, 0.9886020001326434, 0.14991425965987268, 0.45955395816033995],
    [0.0, 0.0, 0.0, 1.0]], np.float32)


def write_snapshot(path, config, params, density_grid, binary=None):
    """Writes a snapshot in the format the reference consumes: msgpack of a
    JSON object whose `snapshot.params` / `snapshot.density_grid` are plain arrays of numbers
    (float32 on the wire).  binary="__half" | "float": instant-ngp's convention instead -- the same values
    as raw blobs `params_binary` / `density_grid_binary` with `params_type` / `density_grid_type`."""
    import msgpack

    cfg = {k: v for k, v in config.items() if k != "snapshot"}
    snap = dict(config["snapshot"])
    if binary:
        dt = np.float16 if binary == "__half" else np.float32
        snap["params_binary"] = np.asarray(params, np.float32).astype(dt).tobytes()
        snap["params_type"] = binary
        if density_grid is not None:
            snap["density_grid_binary"] = np.asarray(density_grid, np.float32).astype(dt).tobytes()
            snap["density_grid_type"] = binary
        cfg["snapshot"] = snap
        with open(path, "wb") as f:
            f.write(msgpack.packb(cfg, use_single_float=True, use_bin_type=True))
        return
    snap["params"] = np.asarray(params, np.float32).tolist()
    if density_grid is not None:  # None: a snapshot without a grid (NerfRender::generate_density_grid makes one)
        snap["density_grid"] = np.asarray(density_grid, np.float32).tolist()
    cfg["snapshot"] = snap
    with open(path, "wb") as f:
        f.write(msgpack.packb(cfg, use_single_float=True))


def read_snapshot(path):
    """Inverse of write_snapshot: returns the config dict (arrays as lists)."""
    import msgpack

    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), raw=False)


def write_ngp_snapshot(path, config, params, density_grid, aabb_scale, grid_dtype="__half"):
    """The same model in instant-ngp's snapshot layout (nerfhip.py "instant-ngp snapshots"): `density_grid` is the
    reference's x-major float grid [C][H][H][H] with bound = aabb_scale / 2; it becomes instant-ngp's cascades 1..K
    (cascade 0 for aabb_scale 1) in Morton order, cascade 0 is left empty, `per_level_scale` is left to the loader
    and the aabb is written in unit-cube coordinates."""
    import msgpack

    from nerfhip import morton3d

    snap_in = config["snapshot"]
    H = int(snap_in["density_grid_size"])
    C = int(snap_in["cascade"])
    bound = aabb_scale / 2.0
    assert float(snap_in["bound"]) == bound and C == (1 if aabb_scale == 1 else aabb_scale.bit_length() - 1)
    g = np.asarray(density_grid, np.float32).reshape(C, H, H, H)
    ax = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    m = morton3d(X, Y, Z).reshape(-1)
    n_ngp = aabb_scale.bit_length()
    out = np.zeros((n_ngp, H ** 3), np.float32)
    for k in range(C):
        out[0 if aabb_scale == 1 else k + 1, m] = g[k].reshape(-1)
    dt = np.float16 if grid_dtype == "__half" else np.float32
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in config.items() if k != "snapshot"}
    cfg["encoding"].pop("per_level_scale", None)
    aabb = [float(v) + 0.5 for v in snap_in["aabb"]]
    cfg["snapshot"] = {
        "version": 1, "n_params": int(np.asarray(params).size), "params_type": "__half",
        "params_binary": np.asarray(params, np.float32).astype(np.float16).tobytes(),
        "density_grid_size": H, "density_grid_binary": out.astype(dt).tobytes(),
        "aabb": {"min": aabb[:3], "max": aabb[3:]},
        "nerf": {"aabb_scale": int(aabb_scale), "rgb_activation": "None",
                 "dataset": {"aabb_scale": int(aabb_scale), "scale": float(snap_in.get("scale", 0.33)), "offset": [0.5, 0.5, 0.5]}},
    }
    with open(path, "wb") as f:
        f.write(msgpack.packb(cfg, use_single_float=True, use_bin_type=True))


def write_transforms_json(path, poses, width, height, camera_angle_x=0.6911112070083618):
    """A camera path in the NeRF-synthetic / instant-ngp `transforms.json` layout."""
    import json

    frames = [{"file_path": f"./frame_{i:04d}", "transform_matrix": np.asarray(p, np.float64).reshape(4, 4).tolist()}
              for i, p in enumerate(poses)]
    with open(path, "w") as f:
        json.dump({"camera_angle_x": camera_angle_x, "w": int(width), "h": int(height), "frames": frames}, f)


def load_transforms_json(path, width=None, height=None):
    """`transforms.json` -> (cams [n][4] = fl_x, fl_y, cx, cy; poses [n][4][4]; width, height).  Intrinsics: explicit
    fl_x / fl_y / cx / cy / w / h when present, else from camera_angle_x (camera_angle_y) over the given resolution."""
    import json

    with open(path) as f:
        t = json.load(f)
    W = int(width or t.get("w", 0))
    H = int(height or t.get("h", 0))
    if W <= 0 or H <= 0:
        raise RuntimeError("transforms.json carries no resolution (w, h): pass width and height")
    sx = W / float(t.get("w", W))
    sy = H / float(t.get("h", H))
    if "fl_x" in t:
        fl_x = float(t["fl_x"]) * sx
    elif "camera_angle_x" in t:
        fl_x = 0.5 * W / math.tan(0.5 * float(t["camera_angle_x"]))
    else:
        raise RuntimeError("transforms.json: neither fl_x nor camera_angle_x")
    if "fl_y" in t:
        fl_y = float(t["fl_y"]) * sy
    elif "camera_angle_y" in t:
        fl_y = 0.5 * H / math.tan(0.5 * float(t["camera_angle_y"]))
    else:
        fl_y = fl_x
    cx = float(t["cx"]) * sx if "cx" in t else 0.5 * W
    cy = float(t["cy"]) * sy if "cy" in t else 0.5 * H
    poses = np.stack([np.asarray(fr["transform_matrix"], np.float32).reshape(4, 4) for fr in t["frames"]])
    cams = np.tile(np.array([fl_x, fl_y, cx, cy], np.float32), (len(poses), 1))
    return cams, poses, W, H
