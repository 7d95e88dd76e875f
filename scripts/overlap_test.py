"""A rank of an 8-GPU tile-sharded step on ONE GPU, with a stand-in for the RCCL gather on the communication stream (a
one-workgroup kernel that spins for the 2.5-3 ms that 133 MB per sender / 930 MB into the sink need over point-to-point
xGMI links): what does the exchange cost per step, in the forms bench.py can take?
  A  a stream per slot, float planes + quantise pass, untile on the sink          (round-2 bench before this script)
  B  ONE render stream, the kernel writes the packed 8-bit shard itself (nrf_bind_output_rgbd8)
A persistent render kernel owns every compute unit it runs on: a kernel of another stream gets a wave slot only when it
becomes eligible at the same moment as a render (form B: the exchange of step i waits for render i's event, render
i + 1 for render i on its stream -- both start when render i ends) or when no render is resident (form A: in the gap
before the render that reuses step i's buffers).  Leaving 4-16 compute units free for the exchange did not help form B
(14.2 -> 14.3-14.6 ms) and was dropped.  `sink_every`: 1 = this rank assembles
every step's frames (a fixed sink), 8 = every 8th step's (the sink rotates over the ranks).
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
LINK_MS = 2.7


def run(form, exchange=True, sink_every=1, steps=16):
    slots = []
    for _ in range(2):
        c = nh.NerfHip(0); c.load_model(desc)
        o = nh.default_options(); o.shard_count, o.shard_index = N, idx
        c.set_options(o); c.set_resolution(W, H)
        sl = type("S", (), {})()
        sl.ctx, sl.stream = c, torch.cuda.Stream(dev)
        sl.send = torch.zeros((V, n_px), dtype=torch.int32, device=dev)
        if form == "A":
            sl.rgba = torch.zeros((V, n_px, 4), device=dev); sl.depth = torch.zeros((V, n_px), device=dev)
            c.bind_output(sl.rgba.data_ptr(), sl.depth.data_ptr())
        else:
            c.bind_output_rgbd8(sl.send.data_ptr())
        sl.all = torch.empty((N, V, n_px), dtype=torch.int32, device=dev)
        sl.frame = torch.empty((V, H, W), dtype=torch.int32, device=dev)
        sl.rendered, sl.gathered = torch.cuda.Event(), torch.cuda.Event()
        slots.append(sl)
    comm = torch.cuda.Stream(dev)
    if form == "B":
        for sl in slots:
            sl.stream = slots[0].stream

    def step(i):
        sl = slots[i % 2]
        sl.stream.wait_event(sl.gathered)
        sl.ctx.render_views(cams, poses, stream=sl.stream.cuda_stream)
        if exchange:
            if form == "A":
                sl.ctx.quantize_rgbd8(sl.rgba.data_ptr(), sl.depth.data_ptr(), V * n_px, sl.send.data_ptr(), stream=sl.stream.cuda_stream)
            sl.rendered.record(sl.stream)
            with torch.cuda.stream(comm):
                comm.wait_event(sl.rendered)
                torch.cuda._sleep(int(LINK_MS * 1e-3 * 2.1e9))  # the send (every rank) / the receives (the sink): link-bound
                if i % sink_every == 0:
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


print(f"render only: form A {run('A', exchange=False):.3f} ms per step, form B {run('B', exchange=False):.3f}", flush=True)
for sink_every in (1, 8):
    a = run("A", sink_every=sink_every)
    print(f"sink duty every {sink_every} step(s): form A {a:.3f} ms per step, form B {run('B', sink_every=sink_every):.3f}", flush=True)
