"""One view per launch (the reference's render_frame shape) under variations of the context state: bound / unbound output planes,
after 16-view launches or not, planned queue order on / off.  usage: single_view_probe.py [lib.so]"""
import os, sys, pathlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn
if len(sys.argv) > 1:
    nh.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
W, H, V = 1920, 1080, 16
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
cam = syn.default_camera(W, H)
poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
cams = np.stack([cam] * V); pv = np.stack([poses[v % 8] for v in range(V)])
st = torch.cuda.Stream()


def run(tag, bound, batch_first, plan=None, set_opts=False):
    if plan is not None:
        os.environ["NRF_PLAN_MAX_POS"] = str(plan)
    c = nh.NerfHip(0); c.load_model(desc)
    os.environ.pop("NRF_PLAN_MAX_POS", None)
    if set_opts:
        c.set_options(nh.default_options())
    c.set_resolution(W, H)
    if bound:
        rgba = torch.zeros((V, W * H, 4), device="cuda"); depth = torch.zeros((V, W * H), device="cuda")
        c.bind_output(rgba.data_ptr(), depth.data_ptr())
    else:
        c.set_max_views(V)
    torch.cuda.synchronize()
    if batch_first:
        for _ in range(6):
            c.render_views(cams, pv, stream=st.cuda_stream)
        torch.cuda.synchronize()
    ms = []
    for rep in range(3):
        for p in poses:
            c.render(cam, p, stream=st.cuda_stream); torch.cuda.synchronize()
            ms.append(c.stats().render_ms)
    print(f"{tag}: first pass {np.mean(ms[:8]):.4f} ms, later passes {np.mean(ms[8:]):.4f} ms", flush=True)
    c.close()


print("lib:", nh.LIB_PATH.name)
run("unbound, after batches", False, True)
run("bound, after batches", True, True)
run("bound, set_options, after batches (bench.py)", True, True, set_opts=True)
run("bound, no batches", True, False)
run("unbound, plan off", False, True, plan=0)
run("bound, plan off", True, True, plan=0)
