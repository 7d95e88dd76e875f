import sys, os, ctypes as C, pathlib
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path("nerf-cuda_amd/libnerfhip_v12.so").resolve()
big, kb, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
ctx.load_model(big)
W, H = 1920, 1080
ctx.set_resolution(W, H)
cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
ref=None; bad=0
for i in range(int(os.environ.get("NFRAMES", "200"))):
    ctx.render(cam, pose)
    a_, d_ = ctx.read_f32()
    if ref is None: ref = a_.copy(); continue
    diff = np.abs(a_ - ref).max(axis=2)
    if diff.max() > 0:
        bad += 1
        ys, xs = np.nonzero(diff)
        print("frame", i, len(ys), "px differ; tile", (int(ys[0])//8)*240+int(xs[0])//8, "max", diff.max())
print("image-level glitches:", bad)
buf = (C.c_uint32 * 8192)()
rc = ctx.lib.nrf_debug_read(buf, 8192)
a = np.frombuffer(buf, np.uint32)
print("rc", rc, "events", a[0])
for k in range(min(int(a[0]), 40)):
    r = a[16 + 16 * k: 32 + 16 * k]
    f = lambda u: np.array([u], np.uint32).view(np.float32)[0]
    h = lambda u: np.array([u], np.uint32).view(np.float16)
    print(f"flags {r[0]} lane {r[1]} n {r[2]} S {r[3]} dirf1 {h(r[4])},{h(r[5])} dirf2 {h(r[6])},{h(r[7])} o0 {f(r[8]):.5f}/{f(r[9]):.5f} sig {f(r[10]):.4f}/{f(r[11]):.4f} ray {r[12]} blk {r[13]} tid {r[14]}")
