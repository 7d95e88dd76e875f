"""Looks for performance cliffs over the knobs a user of the reference's base.json / snapshots turns: table size, level
geometry, volume bound / cascades, density-grid size.  Prints the kernel instance every variant gets (0 register-resident,
1 generic, 2 wide; +16 persistent form) and ms per 1080p view in a 4-view launch."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
W, H, V = 1920, 1080, 4
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(45.0 * i, 30.0) for i in range(V)])
VARIANTS = [
    ("base", {}),
    ("T = 2^14", dict(log2_hashmap_size=14)), ("T = 2^24", dict(log2_hashmap_size=24)),
    ("base_resolution 8", dict(base_resolution=8)), ("base_resolution 64", dict(base_resolution=64)),
    ("finest level 512 (per_level_scale 1.26)", dict(per_level_scale=1.2599)), ("finest level 8192 (per_level_scale 1.5157)", dict(per_level_scale=1.5157)),
    ("density grid 64", dict(H=64)), ("density grid 256", dict(H=256)),
    ("bound 1.5", dict(bound=1.5)), ("bound 2, 2 cascades", dict(bound=2.0, cascade=2)), ("bound 3, 3 cascades", dict(bound=3.0, cascade=3)),
    ("bound 8, 4 cascades", dict(bound=8.0, cascade=4)), ("bound 64, 7 cascades", dict(bound=64.0, cascade=7)),
    ("bound 128, 8 cascades", dict(bound=128.0, cascade=8)),
    ("Dense grid type", dict(grid_type="Dense", n_levels=16, per_level_scale=1.1)), ("Tiled grid type", dict(grid_type="Tiled")),
    ("SH degree 2", dict(sh_degree=2)), ("SH degree 1", dict(sh_degree=1)), ("Identity directions", dict(dir_otype="Identity")),
    ("rgb output Sigmoid", dict(rgb_output_activation="Sigmoid")),
]
for name, kw in VARIANTS:
    try:
        geo = dict(log2_hashmap_size=19, H=128)
        geo.update({k: kw[k] for k in list(kw) if k in ("log2_hashmap_size", "H", "bound", "cascade")})
        rest = {k: v for k, v in kw.items() if k not in geo}
        desc, keep, _ = models.build_model(**geo, **rest)
        c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
        c.lib.nrf_debug_instance.argtypes = [C.c_void_p]
        inst = c.lib.nrf_debug_instance(c.h)
        o = nh.default_options()
        if geo.get("bound", 1.0) > 1.0:
            o.max_steps = 1024
        c.set_options(o)
        s = torch.cuda.Stream()
        c.render_views(cams, poses, stream=s.cuda_stream); torch.cuda.synchronize()
        samples = c.stats().n_samples
        t0 = time.perf_counter()
        for _ in range(3):
            c.render_views(cams, poses, stream=s.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name:44s} instance {inst:2d}  {dt/V*1e3:7.3f} ms per view  {samples/dt/1e6:7.0f} Msamples/s  ({samples/V/1e6:.1f} M samples per view)", flush=True)
        c.close()
    except Exception as e:
        print(f"{name:44s} {type(e).__name__}: {str(e)[:120]}", flush=True)
