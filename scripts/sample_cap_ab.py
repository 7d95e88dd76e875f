"""A/B of the per-round sample cap (NRF_SAMPLE_CAP = 0: always up to 8; 1: 1/2/4/8 by transmittance thresholds; 2: the samples a ray
still needs to reach T < 1e-4 if each halves T): 16-view launches and single views of the bench scene, device time per launch
(nrf_stats.render_ms), evaluated / composited samples.  Contexts of the three modes alternate on one box."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

W, H, V = 1920, 1080, 16
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
cams = np.stack([cam] * V)
pv = np.stack([poses[v % 8] for v in range(V)])
modes = [int(m) for m in (sys.argv[1:] or ["0", "1", "2"])]
ctxs = {}
for m in modes:
    os.environ["NRF_SAMPLE_CAP"] = str(m)
    c = nh.NerfHip(0); c.load_model(desc); c.set_resolution(W, H); c.set_max_views(V)
    ctxs[m] = c
st = torch.cuda.Stream()
res = {m: {"batch": [], "single": [], "host1": []} for m in modes}
for rep in range(4):
    for m in modes:
        c = ctxs[m]
        for i in range(4):
            c.render_views(cams, pv, stream=st.cuda_stream); torch.cuda.synchronize()
            if rep: res[m]["batch"].append(c.stats().render_ms)
        sb = c.stats()
        for p in poses:
            c.render(cam, p, stream=st.cuda_stream); torch.cuda.synchronize()
            if rep: res[m]["single"].append(c.stats().render_ms)
        s1 = c.stats()
        res[m]["ratio_batch"] = sb.n_samples / sb.n_composited
        res[m]["ratio_single"] = s1.n_samples / s1.n_composited
        res[m]["rounds_batch"] = sb.n_rounds
for m in modes:
    r = res[m]
    print(f"NRF_SAMPLE_CAP={m}: 16 views {np.mean(r['batch']):.3f} ms per launch ({np.mean(r['batch'])/V:.4f} per view), evaluated/composited "
          f"{r['ratio_batch']:.4f}, rounds {r['rounds_batch']}; one view {np.mean(r['single']):.4f} ms (min {np.min(r['single']):.4f}), "
          f"evaluated/composited {r['ratio_single']:.4f}", flush=True)
