"""16-view 1080p launches of a generic-instance model (for rocprofv3 --pmc runs): python scripts/gen_profile_target.py [shape]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import models, nerfhip as nh, synthetic as syn
shape = sys.argv[1] if len(sys.argv) > 1 else "n32"
kw = {"n32": dict(n_neurons=32), "f4": dict(n_features_per_level=4, n_levels=8, interpolation="Smoothstep"), "sh6": dict(dir_otype="SphericalHarmonics", sh_degree=6)}[shape]
os.environ["NRF_WIDTH_INSTANCES"] = "0"
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128, **kw)
c = nh.NerfHip(0); c.load_model(desc)
W, H, V = 1920, 1080, 16
c.set_resolution(W, H); c.set_max_views(V)
cams = np.stack([syn.default_camera(W, H)] * V)
poses = [syn.orbit_pose(45.0 * (v % 8), 30.0) for v in range(V)]
for i in range(4):
    c.render_views(cams, poses)
    print(shape, "ms per view", c.stats().render_ms / V, "samples", c.stats().n_samples, flush=True)
