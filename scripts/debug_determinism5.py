import sys, os, ctypes as C, pathlib
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path("nerf-cuda_amd/libnerfhip_v14.so").resolve()
big, kb, _ = models.build_model(log2_hashmap_size=19, H=128)
ctx = nh.NerfHip(0)
ctx.load_model(big)
W, H = 1920, 1080
ctx.set_resolution(W, H)
cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
ref = None; bad = 0
for i in range(int(os.environ.get("NFRAMES", "300"))):
    ctx.render(cam, pose)
    a_, d_ = ctx.read_f32()
    if ref is None: ref = a_.copy(); continue
    diff = np.abs(a_ - ref).max(axis=2)
    if diff.max() > 0:
        bad += 1
        ys, xs = np.nonzero(diff)
        print("frame", i, len(ys), "px differ; tile", (int(ys[0])//8)*240+int(xs[0])//8)
print("image-level glitches:", bad)
buf = (C.c_uint32 * 8192)()
rc = ctx.lib.nrf_debug_read(buf, 8192)
a = np.frombuffer(buf, np.uint32)
print("rc", rc, "events", a[0])
for k in range(min(int(a[0]), 80)):
    r = a[16 + 16 * k: 32 + 16 * k]
    h = lambda u: np.array([u], np.uint32).view(np.float16)
    print(f"round {r[0]} lane {r[1]} j {r[2]} S {r[3]} got {r[4]:08x} {h(r[4])} want {r[5]:08x} {h(r[5])} d0 {r[6]:08x} d7 {r[7]:08x} blk {r[8]} wave {r[9]} A {r[10]} nstep {r[11]} tile {r[12]}")
