import sys
sys.path[:0]=['nerf-cuda_amd','tests']
import numpy as np, models, nerfhip as nh, synthetic as syn, time
W,H=1920,1080
desc,keep,cfg=models.build_model(log2_hashmap_size=19,H=128)
cam=syn.default_camera(W,H)
ctx=nh.NerfHip(0); ctx.load_model(desc); ctx.set_resolution(W,H); ctx.set_max_views(16)
for V in (1,16):
    cams=np.ascontiguousarray(np.stack([cam]*V),np.float32)
    ps=np.ascontiguousarray(np.stack([syn.orbit_pose(45.0*(v%8),30.0) for v in range(V)]),np.float32).reshape(V,16)
    for i in range(3):
        t0=time.perf_counter(); f=ctx.render_host_u8_raw(cams,ps); dt=(time.perf_counter()-t0)*1e3
        print(f"V={V} call {dt:.3f} ms device {f.render_ms:.3f}", file=sys.stderr, flush=True)
