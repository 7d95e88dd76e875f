"""Throughput of the three network instances (register-resident / wide / generic) in both schedulings of the render kernel:
16 views of 1920x1080 per launch, persistent form against NRF_PERSISTENT=0 (one workgroup per strip), same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

W, H, V = 1920, 1080, 16
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(45.0 * (i % 8), 30.0) for i in range(V)])
SHAPES = [
    ("base.json shape (register-resident)", {}),
    ("Frequency-12 directions (wide)", dict(dir_otype="Frequency", n_frequencies=12)),
    ("16 neurons (width instance when persistent)", dict(n_neurons=16)),
    ("32 neurons (width instance when persistent)", dict(n_neurons=32)),
    ("128 neurons (width instance when persistent)", dict(n_neurons=128)),
    ("32 neurons, NRF_WIDTH_INSTANCES=0 (generic)", dict(n_neurons=32, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("128 neurons, NRF_WIDTH_INSTANCES=0 (generic)", dict(n_neurons=128, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("F = 4 x 8 levels, Smoothstep (GRID instance when persistent)", dict(n_features_per_level=4, n_levels=8, interpolation="Smoothstep")),
    ("F = 4 x 8 levels (GRID instance)", dict(n_features_per_level=4, n_levels=8)),
    ("F = 8 x 4 levels (GRID instance)", dict(n_features_per_level=8, n_levels=4)),
    ("F = 2 x 8 levels (GRID instance)", dict(n_levels=8)),
    ("F = 2 x 16 levels, Smoothstep (GRID instance)", dict(interpolation="Smoothstep")),
    ("F = 1 x 16 levels (GRID instance, round 5)", dict(n_features_per_level=1)),
    ("F = 1 x 16 levels, NRF_WIDTH_INSTANCES=0 (generic)", dict(n_features_per_level=1, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("F = 2 x 16 levels, Nearest (GRID instance, round 5)", dict(interpolation="Nearest")),
    ("F = 4 x 8 levels, Nearest (GRID instance, round 5)", dict(n_features_per_level=4, n_levels=8, interpolation="Nearest")),
    ("F = 2 x 16 levels, Nearest, NRF_WIDTH_INSTANCES=0 (generic)", dict(interpolation="Nearest", _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("F = 4 x 8 levels, Smoothstep, NRF_WIDTH_INSTANCES=0 (generic)", dict(n_features_per_level=4, n_levels=8, interpolation="Smoothstep", _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("F = 2 x 8 levels, NRF_WIDTH_INSTANCES=0 (generic)", dict(n_levels=8, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("SH degree 6 (wide-SH form when persistent)", dict(sh_degree=6)),
    ("SH degree 8 (wide-SH form when persistent)", dict(sh_degree=8)),
    ("SH degree 6, NRF_WIDTH_INSTANCES=0 (generic)", dict(sh_degree=6, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("hidden layers 2 + 2 (depth instance when persistent)", dict(density_hidden_layers=2, rgb_hidden_layers=2)),
    ("hidden layers 1 + 1 (depth instance)", dict(density_hidden_layers=1, rgb_hidden_layers=1)),
    ("hidden layers 3 + 4 (depth instance)", dict(density_hidden_layers=3, rgb_hidden_layers=4)),
    ("Squareplus hidden activations (NET_ACT instance, round 6)", dict(activation="Squareplus")),
    ("Softplus hidden activations (NET_ACT instance, round 6)", dict(activation="Softplus")),
    ("Sigmoid hidden activations (NET_ACT instance, round 6)", dict(activation="Sigmoid")),
    ("Squareplus hidden activations, NRF_WIDTH_INSTANCES=0 (generic)", dict(activation="Squareplus", _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("Sigmoid hidden activations, NRF_WIDTH_INSTANCES=0 (generic)", dict(activation="Sigmoid", _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("hidden layers 2 + 2, NRF_WIDTH_INSTANCES=0 (generic)", dict(density_hidden_layers=2, rgb_hidden_layers=2, _env={"NRF_WIDTH_INSTANCES": "0"})),
    ("hidden layers 3 + 4, NRF_WIDTH_INSTANCES=0 (generic)", dict(density_hidden_layers=3, rgb_hidden_layers=4, _env={"NRF_WIDTH_INSTANCES": "0"})),
]
only = sys.argv[1] if len(sys.argv) > 1 else None  # substring filter on the shape names
for name, kw in SHAPES:
    if only and only not in name:
        continue
    kw = dict(kw)
    env = kw.pop("_env", {})
    os.environ.update(env)
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128, **kw)
    res = []
    for persistent in ("0", "1"):
        os.environ["NRF_PERSISTENT"] = persistent
        c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
        s = torch.cuda.Stream()
        c.render_views(cams, poses, stream=s.cuda_stream); torch.cuda.synchronize()
        samples = c.stats().n_samples
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            c.render_views(cams, poses, stream=s.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res.append((dt / V * 1e3, samples / dt / 1e6))
        c.close()
    for k in env:
        os.environ.pop(k, None)
    print(f"{name:48s} per strip {res[0][0]:7.3f} ms/view {res[0][1]:7.0f} Msamples/s | persistent {res[1][0]:7.3f} ms/view {res[1][1]:7.0f} Msamples/s "
          f"({100 * (res[0][0] / res[1][0] - 1):+.0f} %)", flush=True)
