"""The gather root of an 8-GPU tile-sharded step on ONE GPU, with a stand-in for the RCCL gather on the communication stream
(a one-workgroup kernel that spins for the ~3 ms the 930 MB of seven shards need over seven xGMI links, plus one local
copy), the quantise pass and the untile of the 128 frames: what does the root's duty cost per step, and what if the
root rotates (every rank assembles every N-th step's frames)?  A persistent render kernel owns every compute unit,
so the exchange of step i cannot start while a render is resident: it runs in the gaps between renders.
(Tried and dropped: leaving 8-32 compute units free for the exchange -- the next render's workgroups take them, and with
one render stream the quantise / untile passes are too slow on so few units: 14.6-18 ms per step.)
usage: scripts/overlap_test.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nerf-cuda_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import models, nerfhip as nh, synthetic as syn

N, idx, W, H = 8, 0, 1920, 1080
V = 16 * N
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
tps = nh.tiles_per_shard(W, H, N)
n_px = tps * 64
cams = np.stack([syn.default_camera(W, H)] * V)
poses = np.stack([syn.orbit_pose(45.0 * (i % 8), 30.0) for i in range(V)])
dev = torch.device("cuda", 0)


def run(reserved, exchange=True, steps=16, one_render_stream=False, fused_quantise=False, rotate_root=False):
    slots = []
    for _ in range(2):
        c = nh.NerfHip(0); c.load_model(desc)
        o = nh.default_options(); o.shard_count, o.shard_index = N, idx
        c.set_options(o); c.set_resolution(W, H)
        sl = type("S", (), {})()
        sl.ctx, sl.stream = c, torch.cuda.Stream(dev)
        sl.rgba = torch.zeros((V, n_px, 4), device=dev); sl.depth = torch.zeros((V, n_px), device=dev)
        c.bind_output(sl.rgba.data_ptr(), sl.depth.data_ptr())
        sl.send = torch.zeros((V, n_px), dtype=torch.int32, device=dev)
        sl.all = torch.empty((N, V, n_px), dtype=torch.int32, device=dev)
        sl.frame = torch.empty((V, H, W), dtype=torch.int32, device=dev)
        sl.rendered, sl.gathered = torch.cuda.Event(), torch.cuda.Event()
        slots.append(sl)
    comm = torch.cuda.Stream(dev)
    if one_render_stream:  # renders in order on ONE stream: never two persistent kernels resident, the reserved CUs stay free
        for sl in slots:
            sl.stream = slots[0].stream

    def step(i):
        sl = slots[i % 2]
        sl.stream.wait_event(sl.gathered)
        sl.ctx.render_views(cams, poses, stream=sl.stream.cuda_stream)
        if exchange:
            if not one_render_stream and not fused_quantise:
                sl.ctx.quantize_rgbd8(sl.rgba.data_ptr(), sl.depth.data_ptr(), V * n_px, sl.send.data_ptr(), stream=sl.stream.cuda_stream)
            sl.rendered.record(sl.stream)
            with torch.cuda.stream(comm):
                comm.wait_event(sl.rendered)
                if one_render_stream and not fused_quantise:
                    sl.ctx.quantize_rgbd8(sl.rgba.data_ptr(), sl.depth.data_ptr(), V * n_px, sl.send.data_ptr(), stream=comm.cuda_stream)
                # stand-in for the gather on rank 0: 7 shards arrive over xGMI -- link-bound (~3 ms for 930 MB over 7 links), driven
                # by a kernel that needs a wave slot but little else (a spinning one-workgroup kernel), plus one local copy
                root = (not rotate_root) or i % N == 0  # rotating root: this rank assembles every N-th step's frames
                torch.cuda._sleep(int((3e-3 if root else 0.4e-3) * 2.1e9))
                if root:
                    sl.all[0].copy_(sl.send, non_blocking=True)
                    sl.ctx.untile_views(sl.all.data_ptr(), N, tps, 1, V, sl.frame.data_ptr(), stream=comm.cuda_stream)
                sl.gathered.record(comm)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for sl in slots:
        sl.ctx.close()
    return dt * 1e3


alone = run(0, exchange=False)
both = run(0, exchange=True)
fq = run(0, exchange=True, fused_quantise=True)
rot = run(0, exchange=True, rotate_root=True)
print(f"render only {alone:.3f} ms per step; root every step: {both:.3f}; the same without the quantise pass: {fq:.3f}; "
      f"root duty every {N}th step only: {rot:.3f}", flush=True)
