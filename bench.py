#!/usr/bin/env python3
"""Headline benchmark: megasamples/s (+ frames/s) of the nerf_render hot path at
1920x1080 on the seeded synthetic Lego-like scene (BASELINE.json configs[1]:
hash grid L=16 F=2 T=2^19, 64-wide MLPs, SH-4 directions).

A "step" = one batch of 16 camera views (16 whole 1920x1080 frames of the orbit):
ray generation -> occupancy march -> hash-grid + SH encoding -> fused MLPs ->
compositing -> RGBA/depth in HBM, all inside ONE launch of the fused gfx950
kernel per rank (nrf_render_views; --views-per-step 1 gives one frame per step).
The views of a batch are independent frames: batching only lets the workgroups
of view v+1 take the wave slots that the few long-lived tiles of view v leave
idle (one frame alone: 1.0 ms; in a batch: 0.83 ms per frame).

Multi-GPU (one process per GPU, RCCL):
  --config 3 (default for N > 1; BASELINE.json configs[2]): every frame's tile
      strips are dealt round-robin to the ranks (tile-parallel: every rank renders
      1/N of EVERY frame), two steps are in flight, every rank renders its shard
      straight into the reference's 8-bit image format (r, g, b, depth: 4 B/px;
      nrf_bind_output_rgbd8), all renders of a rank go to one stream, and the only
      exchange is one RCCL gather of the batch's shards to the step's sink rank
      (step % N by default, --gather-root 0: always rank 0), followed by an untile
      kernel there (--gather-format f32 ships float RGBA instead).
      --scaling weak (default): a step is 16 x N frames, so every rank renders 16
      frames' worth of tiles per step whatever N is; --scaling strong: a step is
      the same 16 frames at every N (1/N-th of the work per rank and step).
  --config 5 (BASELINE.json configs[4]): 64 camera requests of 800x800 per step,
      replica-parallel: rank r renders whole frames of requests r*64/N .. and the
      8-bit images are gathered on rank 0 (no untile: frames are whole).  Strong
      scaling by definition (64 requests per step); --scaling weak: 8 x N requests.

`python bench.py --gpus N` WITHOUT a launcher starts the N ranks itself (fresh
`torch.distributed.run` children, started before this process touches a GPU)
and relays rank 0's line; under a launcher WORLD_SIZE must equal --gpus, or the
run fails.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (render_persistent_kernel); the contract's figure: algorithmic
                gather bytes (512 B/sample = 16 levels x 8 corners x half2) over the
                kernel's mean duration measured with HIP events on its stream,
                against the HBM peak -- plus what the counters say really binds it
                (`limiter`) and the measured HBM rate (`hbm_gbs_measured`).
  cpu_baseline  the CPU oracle (a port, not the reference binary: the reference
                is CUDA-only) timed on this host on a bounded sample.
  parity        the metric's image-quality leg: PSNR / max |d| of the HIP frame
                against the oracle's frame of that sample.
"""
from __future__ import annotations

import os

# Steps in flight run on separate HIP streams; the runtime maps streams onto 4 hardware queues by
# default and kernels sharing a queue serialise.  8 queues keep the render streams, the RCCL stream
# and the untile stream apart (measured with single-view steps: 0.196 -> 0.124 ms per 1/8-frame
# shard, scripts/pipeline_test.py).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import hashlib
import json
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
# the package first (nerfhip, synthetic, models); tests/ only holds the oracle's binding, which the cpu_baseline leg imports
for p in (ROOT / "tests", ROOT / "nerf-cuda_amd"):
    sys.path.insert(0, str(p))

WIDTH, HEIGHT = 1920, 1080
BYTES_PER_SAMPLE = 16 * 8 * 4      # hash-grid gather, the path's algorithmic traffic
FLOP_PER_SAMPLE = 20480            # both MLPs, padded
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16
CLOCK_HZ = 2.4e9                   # max engine clock (the chip holds ~2.0-2.1 GHz under this load: nrf_stats.shader_clock_mhz)
TA_UNITS = 256                     # one texture addresser per compute unit
TA_ADDR_PER_CLK = 4.0              # lane addresses a texture addresser takes per clock (profiles/r02/gather_probe.txt: a 64-lane gather of
                                   # 4-, 8- or 16-byte entries holds it ~16-17 cycles whatever the entry size)
GATHER_ADDR_PER_SAMPLE = 16 * 8    # hot instance: 16 levels x 8 corners, one lane address each
DEFAULT_VIEWS = 16                 # camera views per step (one launch)
DEFAULT_DEPTH = 1                  # steps in flight at N = 1 (N > 1: 2, so that the gather of one step overlaps the next)
CONFIG5_REQUESTS, CONFIG5_RES = 64, 800  # BASELINE.json configs[4]
PMC_FILE = ROOT / "profiles" / "r06" / "pmc_traffic.json"


def kernel_source_sha16() -> str:
    """Fingerprint of the kernel sources the committed PMC summary was collected for (bench.py drops the
    PMC-derived fields when the sources have changed since)."""
    h = hashlib.sha256()
    # every file under csrc/ feeds the render launch: device code, launch shapes (nrf_api.hip, nrf_launch.h), the generic instance
    for f in sorted((ROOT / "nerf-cuda_amd" / "csrc").iterdir()):
        if f.suffix in (".h", ".hip"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 3, 5),
                    help="BASELINE.json configuration (1-based): 2/3 = 1920x1080 frames, tile-sharded over the ranks "
                         "(default); 5 = 64 requests of 800x800 per step, replica-parallel")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region and its line (no api / fast_interp / mlp_kernel / cpu_baseline objects): what the "
                         "rocprofv3 passes of scripts/profile_gpu.sh run, so that every launch they see is the headline kernel's")
    ap.add_argument("--cpu-sample-div", type=int, default=1,
                    help="CPU baseline renders a (W/div)x(H/div) frame (default: the whole 1920x1080 frame: a few seconds on 128 cores)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="steps in flight; 0 = default (1 at N = 1, else 2)")
    ap.add_argument("--views-per-step", type=int, default=0,
                    help="camera views rendered by ONE launch per step (nrf_render_views); 0 = default")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: weak = the step grows with N (16 x N frames tile-sharded over the ranks; config 5: 8 x N "
                         "requests), strong = the same step at every N.  Default: weak for configs 2/3, strong for config 5")
    ap.add_argument("--gather-format", choices=("rgbd8", "f32"), default="rgbd8",
                    help="N > 1: what the ranks send to rank 0 -- the reference's 8-bit image (r,g,b,depth: 4 B/px, "
                         "quantised on the rendering GPU) or the float RGBA plane (16 B/px)")
    ap.add_argument("--gather-root", choices=("rotate", "0"), default="rotate",
                    help="N > 1: the rank that assembles a step's frames -- step %% N (default: every rank is the sink of every "
                         "N-th step, so no GPU carries the receive + untile of all frames) or always rank 0")
    ap.add_argument("--lib", default=None, help="another build of libnerfhip.so (A/B comparisons on one box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--check", action="store_true", default=None,
                    help="rank 0 also renders the frame unsharded and compares (default: on whenever there is more than one rank)")
    ap.add_argument("--no-check", dest="check", action="store_false")
    ap.add_argument("--dry-exchange", action="store_true",
                    help="rehearsal of the N > 1 plumbing WITHOUT a GPU: the same launch, step plan (views per rank, shard / replica "
                         "blocks, pose indices), rotating sink, gloo gather, untile and check, with shards filled by a per-pixel function "
                         "of (pose, x, y) instead of renders.  What 8 ranks can execute on a box whose GPU takes at most 6 processes")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-GPU exchange even with ONE rank: init_process_group(--backend), the shard rendered tile-major "
                         "and packed, the gather through the backend (RCCL: a one-rank communicator), the untile, the one-stream "
                         "render / exchange hand-off and --check.  What a one-GPU box can execute of the N > 1 path; the line then "
                         "carries no api / configs / cpu_baseline objects")
    return ap.parse_args()


def self_launch(args) -> int:
    """`bench.py --gpus N` without a launcher: start the N ranks as fresh children of torch.distributed.run.
    This (parent) process never imports torch nor touches a GPU; it relays the children's stdout (rank 0's
    JSON line) and returns their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env["NRF_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.run(cmd, env=env).returncode


def step_plan(args, rank, world, replica):
    """What a rank renders per step: (V views per rank and step, V_step views per step, shard_index, shard_count, scaling)."""
    scaling = args.scaling or ("strong" if replica else "weak")
    if replica:
        # rank r takes the contiguous block r*V .. of the step's requests: whole frames, shard_count 1
        if scaling == "strong":  # BASELINE's form: 64 requests per step whatever N is
            V_step = args.views_per_step or CONFIG5_REQUESTS
            assert V_step % world == 0, "config 5: the requests of a step must divide over the ranks"
            V = V_step // world
        else:                    # 8 requests per rank and step (= the 8-GPU share of BASELINE's 64)
            V = args.views_per_step or CONFIG5_REQUESTS // 8
            V_step = V * world
        return V, V_step, 0, 1, scaling
    # tile-parallel: every rank renders its strips of ALL V_step frames (launches of up to NRF_MAX_VIEWS views)
    base = args.views_per_step or DEFAULT_VIEWS
    V = V_step = base * world if scaling == "weak" else base
    return V, V_step, rank, world, scaling


def step_pose_indices(i, rank, V, V_step, replica, n_poses):
    """(global pose indices of step i rendered by rank `rank`)"""
    if replica:
        return [(i * V_step + rank * V + v) % n_poses for v in range(V)]
    return [(i * V + v) % n_poses for v in range(V)]


def step_root(i, world, gather_root):
    """the rank that assembles step i's frames: step % N (every rank is the sink once in N steps) or always rank 0"""
    return i % world if gather_root == "rotate" else 0


def dry_exchange(args, rank, world):
    """--dry-exchange: every part of an N-rank step except the render, on the CPU (gloo).  A "rendered" pixel is the packed
    value f(pose, x, y); rank r fills its tile-major shard of all the step's views (configs 2/3) or its block of whole frames
    (config 5), the step's sink gathers, untiles (nerfhip.untile_numpy: the mapping the device kernel is tested against) and
    compares with f evaluated directly.  One JSON line from rank 0, like a real run's, with `dry_exchange: true`."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import nerfhip as nh
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world
    config = args.config or 3
    replica = config == 5
    W = args.width or (CONFIG5_RES if replica else WIDTH)
    H = args.height or (CONFIG5_RES if replica else HEIGHT)
    V, V_step, shard_index, shard_count, scaling = step_plan(args, rank, world, replica)
    n_poses = CONFIG5_REQUESTS if replica else 8
    tps = nh.tiles_per_shard(W, H, shard_count)
    ys, xs = np.mgrid[0:H, 0:W]

    def frame_of(pose):  # the "render" of a pose: one packed int32 per pixel
        return (xs | (ys << 11) | ((pose + 1) << 22)).astype(np.int32)

    if replica:
        n_px = W * H
    else:
        n_px = tps * 64
        tiles = nh.shard_tile_ids(W, H, rank, world)
        lane = np.arange(64)
        px = np.array([t[0] for t in tiles])[:, None] * 8 + (lane & 7)[None, :]
        py = np.array([t[1] for t in tiles])[:, None] * 8 + (lane >> 3)[None, :]
        inside = (px < W) & (py < H)
    ok, sink_steps = True, 0
    t0 = time.perf_counter()
    for i in range(args.warmup + args.steps):
        mine = step_pose_indices(i, rank, V, V_step, replica, n_poses)
        send = np.zeros((V, n_px), np.int32)
        for v, pose in enumerate(mine):
            f = frame_of(pose)
            if replica:
                send[v] = f.reshape(-1)
            else:
                send[v, :len(tiles) * 64].reshape(len(tiles), 64)[inside] = f[py[inside], px[inside]]  # (a rank may own fewer tiles than tps)
        root = step_root(i, world, args.gather_root)
        parts = [torch.empty((V, n_px), dtype=torch.int32) for _ in range(world)] if rank == root else None
        dist.gather(torch.from_numpy(send), parts, dst=root)
        if rank == root:
            sink_steps += 1
            g = np.stack([p.numpy() for p in parts])  # [world][V][n_px]
            for r in range(world if replica else 1):
                for v, pose in enumerate(step_pose_indices(i, r, V, V_step, replica, n_poses)):
                    got = g[r, v].reshape(H, W) if replica else nh.untile_numpy(g[:, v, :, None], W, H)[..., 0]
                    ok = ok and np.array_equal(got, frame_of(pose))
    elapsed = time.perf_counter() - t0
    flags = torch.tensor([int(ok), sink_steps], dtype=torch.int64)
    all_flags = [torch.zeros(2, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
    dist.gather(flags, all_flags, dst=0)
    dist.barrier()
    if rank == 0:
        sinks = [int(f[1]) for f in all_flags]
        print(json.dumps({
            "dry_exchange": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": scaling,
            "sharded_frame_equals_unsharded": all(int(f[0]) == 1 for f in all_flags),
            "sink_steps_per_rank": sinks,
            "config": {"workload": f"BASELINE config {config}: {W}x{H}, NO render (pixels = f(pose, x, y))",
                       "parallelism": (f"replica{world}" if replica else f"tile{world}"), "views_per_step": V_step,
                       "views_per_rank_and_step": V, "tiles_per_shard": (None if replica else tps),
                       "gather_root": ("step % N" if args.gather_root == "rotate" else "rank 0")},
            "distributed": {"world_size": world, "backend": "gloo",
                            "launcher": "bench.py self-launch" if os.environ.get("NRF_BENCH_SELF_LAUNCHED") else "external"},
            "elapsed_s": round(elapsed, 3)}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist  # the exchange leg of the path (a process group, gather, untile) is executed
    if world != args.gpus:
        # a line with the wrong n_gpus is worse than no line
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the ranks itself)")

    if args.dry_exchange:
        if world < 2:
            sys.exit("bench.py: --dry-exchange rehearses the N > 1 exchange: use --gpus N with N >= 2")
        return dry_exchange(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    import models
    import nerfhip as nh
    import synthetic as syn
    if args.lib:
        nh.LIB_PATH = Path(args.lib).resolve()

    if world > 1 and not args.single_device and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPUs "
                 "(rehearse on one GPU with --backend gloo --single-device)")
    dev_index = 0 if (world == 1 or args.single_device) else local_rank
    torch.cuda.set_device(dev_index)
    stdout_fd = None
    if dist_on:
        # RCCL prints a banner (version, host, library path) on STDOUT when it builds its first communicator; the contract is ONE
        # JSON line there: everything between here and the line itself goes to stderr at the file-descriptor level
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:  # --force-dist without a launcher: this process is the one rank
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    dev = torch.device("cuda", dev_index)
    config = args.config or (3 if dist_on else 2)
    replica = config == 5
    W = args.width or (CONFIG5_RES if replica else WIDTH)
    H = args.height or (CONFIG5_RES if replica else HEIGHT)

    # identical seeded model on every rank (replicated: 24.4 MB table + its 4.6 GB of cell-major gather copies + 256 KB occupancy bits).
    # `depth` steps are in flight: one context + stream + output buffers per slot, so the tail of one
    # batch (a few long-lived tiles) overlaps the head of the next.
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    depth = args.frames_in_flight or (2 if dist_on else DEFAULT_DEPTH)
    V, V_step, shard_index, shard_count, scaling = step_plan(args, rank, world, replica)
    opts = nh.default_options()
    opts.shard_index, opts.shard_count = shard_index, shard_count
    # one rank alone renders row-major frames; with --force-dist its single shard takes the shard layout, as every rank's does at N > 1
    tiled = not replica and (shard_count > 1 or dist_on)
    opts.tile_major = int(tiled and shard_count == 1)
    tps = nh.tiles_per_shard(W, H, shard_count)
    cam = syn.default_camera(W, H)
    if replica:  # 64 distinct cameras around the object (three elevations)
        poses = [syn.orbit_pose(360.0 * i / CONFIG5_REQUESTS, (10.0, 30.0, 50.0)[i % 3]) for i in range(CONFIG5_REQUESTS)]
    else:
        poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
    n_px = tps * 64 if tiled else W * H

    def step_poses(i):
        """(global pose indices of step i rendered by THIS rank)"""
        return step_pose_indices(i, rank, V, V_step, replica, len(poses))

    class Slot:
        pass

    # N > 1: the renders of all slots go to ONE stream.  The persistent render kernel owns every compute unit it runs on, so
    # a kernel of another stream gets a wave slot only if it becomes eligible together with a render: the exchange of step i
    # waits for render i's event and render i + 1 for render i on its stream -- both start when render i ends, RCCL's few
    # workgroups take their slots and the render the rest.  With a stream per slot render i + 1 is resident before render i
    # has ended, and the exchange of step i waits for the next gap (scripts/overlap_test.py: 17.2 against 14.2 ms per step).
    render_stream = torch.cuda.Stream(dev) if dist_on else None
    packed = dist_on and args.gather_format == "rgbd8"
    slots = []
    for _ in range(depth):
        sl = Slot()
        sl.ctx = nh.NerfHip(dev.index)
        sl.ctx.load_model(desc)
        sl.ctx.set_options(opts)
        sl.ctx.set_resolution(W, H)
        # a real (non-NULL) stream: NULL would make the ABI synchronise per call
        sl.stream = render_stream if render_stream is not None else torch.cuda.Stream(dev)
        if packed:
            # the kernel writes the reference's 8-bit pixels (r, g, b, depth: nerf_render.cu:345-359) itself: the shard goes
            # onto the wire as rendered, 4 B/px instead of 20 B/px of float planes and no quantise pass
            sl.send = torch.zeros((V, n_px), dtype=torch.int32, device=dev)
            sl.ctx.bind_output_rgbd8(sl.send.data_ptr())
        else:
            sl.rgba = torch.zeros((V, n_px, 4), device=dev)
            sl.depth = torch.zeros((V, n_px), device=dev)
            sl.ctx.bind_output(sl.rgba.data_ptr(), sl.depth.data_ptr())
        sl.rendered = torch.cuda.Event()
        sl.gathered = torch.cuda.Event()
        if dist_on:
            is_sink = rank == 0 or args.gather_root == "rotate"  # this rank assembles (some of) the steps' frames
            if packed:  # 4-byte pixels: one int32 "channel"
                sl.all = torch.empty((world, V, n_px), dtype=torch.int32, device=dev) if is_sink else None
                sl.frame = (sl.all.view(world * V, H, W) if replica else
                            torch.empty((V, H, W), dtype=torch.int32, device=dev)) if is_sink else None
            else:
                sl.send = sl.rgba
                sl.all = torch.empty((world, V, n_px, 4), device=dev) if is_sink else None
                sl.frame = (sl.all.view(world * V, H, W, 4) if replica else
                            torch.empty((V, H, W, 4), device=dev)) if is_sink else None
            sl.parts = [sl.all[r] for r in range(world)] if is_sink else None
        slots.append(sl)
    ctx = slots[0].ctx
    comm = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)

    launch_events = []  # (start, end) HIP events around every render launch of the timed region
    cams_step = np.stack([cam] * V)

    def step(i, timed=False):
        sl = slots[i % depth]
        if dist_on:
            sl.stream.wait_event(sl.gathered)  # the slot's previous shard has left the building
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sl.stream)
        sl.ctx.render_views(cams_step, [poses[j] for j in step_poses(i)], stream=sl.stream.cuda_stream)
        if timed:
            e1.record(sl.stream)
            launch_events.append((e0, e1))
        if dist_on:
            # the one exchange of the path: every rank's shard / frames -> the step's sink rank (direct xGMI sends: xGMI
            # is point-to-point, so a gather moves 1/N-th of what an all-gather would), untile there
            sl.rendered.record(sl.stream)
            # the rank that assembles this step's frames: step % N -- the receive of N - 1 shards (link-bound: 930 MB over
            # seven xGMI links at N = 8) and the untile of the step's frames are then every rank's duty once in N steps
            # instead of rank 0's in every step (scripts/overlap_test.py: 14.4 -> 13.5 ms per step on the sink)
            root = step_root(i, world, args.gather_root)
            with torch.cuda.stream(comm):
                comm.wait_event(sl.rendered)
                dist.gather(sl.send, sl.parts if rank == root else None, dst=root)
                if rank == root and not replica:
                    sl.ctx.untile_views(sl.all.data_ptr(), world, tps, 4 if args.gather_format == "f32" else 1, V,
                                        sl.frame.data_ptr(), stream=comm.cuda_stream)
                sl.gathered.record(comm)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if dist_on:
        # RCCL sets up its peer-to-peer channels on the first send/recv of every pair: open them now, so that the
        # timed region never pays for connection set-up whatever --warmup is
        with torch.cuda.stream(comm):
            probe = torch.zeros((256,), dtype=torch.int32, device=dev)
            for root in (range(world) if args.gather_root == "rotate" else (0,)):
                dist.gather(probe, [torch.empty_like(probe) for _ in range(world)] if rank == root else None, dst=root)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sample counts of THIS rank's share, from the launches themselves: every distinct step composition is replayed once,
    # untimed, and the kernel's counters are read (evaluated samples depend -- a little -- on how rays are batched into
    # rounds, which differs between a view rendered alone and in a batch; the composited ones do not), together with
    # the duration of one step's launch when it has the chip to itself; and one view alone: what a render_frame costs
    stream = slots[0].stream
    single_ms, kern_ms = [], []
    for j in sorted({j for i in range(args.steps) for j in step_poses(i)}):
        ctx.render(cam, poses[j], stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        single_ms.append(float(ctx.stats().render_ms))
    per_step = {}
    clock_mhz = []
    for i in range(args.steps):
        key = tuple(step_poses(i))
        if key not in per_step:
            ctx.render_views(cams_step, [poses[j] for j in key], stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            st = ctx.stats()
            per_step[key] = (int(st.n_samples), int(st.n_network_evals), int(st.n_composited))
            kern_ms.append(float(st.render_ms))
            if st.shader_clock_mhz > 0:
                clock_mhz.append(float(st.shader_clock_mhz))
    # lane addresses a sample sends into the texture path with the loaded model: 8 per level, 2 for a level gathered from its
    # cell-major quad copy (nrf_stats, ABI 6)
    gather_addr_sample = int(ctx.stats().gather_addresses_per_sample) or GATHER_ADDR_PER_SAMPLE
    grid_device_mb = round(int(ctx.stats().grid_device_bytes) / 1e6, 1)
    step_counts = [per_step[tuple(step_poses(i))] for i in range(args.steps)]
    # the sample count of the metric and of the roofline is the COMPOSITED one: the samples that reach a ray's compositing sum,
    # equal to the reference's own per-ray count (and the oracle's) and independent of timing; the kernel also evaluates the
    # samples a ray queues behind its terminating one (n_samples: +2-3 %, timing-dependent) -- reported beside it, not in `value`
    step_samples = [c[2] for c in step_counts]
    step_evaluated = [c[0] for c in step_counts]
    local_evals, local_composited = sum(c[1] for c in step_counts), sum(c[2] for c in step_counts)
    local_samples = sum(step_evaluated)
    # which physical device every rank sits on: two ranks on one GPU would make an N-GPU line out of fewer GPUs
    props = torch.cuda.get_device_properties(dev_index)
    dev_id = str(getattr(props, "uuid", "")) or f"pci {getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}"
    devices = [f"rank {rank}: cuda:{dev_index} {torch.cuda.get_device_name(dev_index)} [{dev_id}]"]
    # this rank's row of the peer-access matrix (hipDeviceCanAccessPeer towards every visible device): what RCCL's P2P transport
    # and nrf_group's hipMemcpyPeerAsync rely on -- so that the first real N-GPU run records it
    n_vis = torch.cuda.device_count()
    peer_rows = [[int(j == dev_index or torch.cuda.can_device_access_peer(dev_index, j)) for j in range(n_vis)]]
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001  (a build without the nccl bindings)
        rccl_version = None
    if dist_on:
        t = torch.tensor([local_samples, local_evals, local_composited], device=dev, dtype=torch.int64)
        dist.all_reduce(t)
        total_samples, total_evals, total_composited = (int(v) for v in t.tolist())
        gathered = [None] * world
        dist.all_gather_object(gathered, (devices[0], dev_id, peer_rows[0]))
        devices = [g[0] for g in gathered]
        ids = [g[1] for g in gathered]
        peer_rows = [g[2] for g in gathered]
        if len(set(ids)) != world and not args.single_device:
            sys.exit(f"bench.py: {world} ranks on {len(set(ids))} distinct device(s) ({devices}): this would not be an "
                     f"{world}-GPU measurement (rehearsals on one GPU: --backend gloo --single-device)")
    else:
        total_samples, total_evals, total_composited = local_samples, local_evals, local_composited

    check = None
    if (args.check if args.check is not None else dist_on) and dist_on:
        # the gathered (+ untiled) frames of a step must equal unsharded single renders of the same poses
        torch.cuda.synchronize(dev)
        step(0)
        torch.cuda.synchronize(dev)
        if rank == 0:
            solo = nh.NerfHip(dev.index)
            solo.load_model(desc)
            solo.set_resolution(W, H)
            # tile-sharded: view 0 of the step; replica: the first view of every rank's block
            # (a step of more than NRF_MAX_VIEWS views is several launches: the last view covers the last of them)
            pairs = ([(r * V, r * V_step // world) for r in range(world)] if replica else
                     sorted({(0, step_poses(0)[0]), (V - 1, step_poses(0)[V - 1])}))
            check = True
            for fi, pj in pairs:
                frame = slots[0].frame[fi]
                solo.render(cam, poses[pj])
                if args.gather_format == "rgbd8":
                    rgb8, d8 = solo.read_u8()
                    want = (rgb8[..., 0].astype(np.uint32) | (rgb8[..., 1].astype(np.uint32) << 8) |
                            (rgb8[..., 2].astype(np.uint32) << 16) | (d8.astype(np.uint32) << 24))
                    check = check and bool(np.array_equal(frame.cpu().numpy().view(np.uint32), want))
                else:
                    want, _ = solo.read_f32()
                    check = check and bool(np.array_equal(frame.cpu().numpy(), want))
            solo.close()
        dist.barrier()

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    msamples_s = total_composited / elapsed / 1e6
    # the kernel's average launch duration over the timed region (HIP events on the launch streams;
    # launches of different slots overlap, so this is what rocprofv3 --stats reports for the same command)
    mean_kern_s = float(np.mean([a.elapsed_time(b) for a, b in launch_events])) * 1e-3
    iso_kern_s = float(np.mean(kern_ms)) * 1e-3  # the same launches replayed one at a time
    mean_samples_launch = float(np.mean(step_samples))
    gather_gbs = mean_samples_launch * BYTES_PER_SAMPLE / mean_kern_s / 1e9
    # what binds the kernel, from this run alone: lane addresses per second into the texture addressers (every EVALUATED sample
    # sends 128: 16 levels x 8 corners, one address per lane whatever the entry size) against their rate at the clock the
    # launch itself measured (s_memtime / s_memrealtime inside the kernel, nrf_stats.shader_clock_mhz)
    clock_hz = float(np.mean(clock_mhz)) * 1e6 if clock_mhz else None
    addr_per_s = float(np.mean(step_evaluated)) * gather_addr_sample / iso_kern_s
    ta_peak = TA_UNITS * TA_ADDR_PER_CLK * clock_hz if clock_hz else None
    in_flight = mean_kern_s * 1e3 / ms_per_step
    single_view_ms = float(np.mean(single_ms))
    # PMC passes cannot run inside this process: the counters of the same command are read from the committed
    # summary -- only when it was collected for exactly this launch shape AND these kernel sources
    pmc = None
    if not dist_on and not replica and (W, H) == (WIDTH, HEIGHT) and V == DEFAULT_VIEWS and PMC_FILE.exists():
        cand = json.loads(PMC_FILE.read_text())
        if cand.get("views_per_launch") == V and cand.get("kernel_source_sha16") == kernel_source_sha16():
            pmc = cand
    traffic = pmc["hbm_bytes_per_launch"] if pmc else None
    valu = pmc.get("valu_insts_per_launch") if pmc else None
    res = f"{W}x{H}"
    out = {
        "metric": f"megasamples/s (march samples composited into a ray = the reference's per-ray sample count), Lego-like NeRF render @{res}",
        "value": round(msamples_s, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "frames_per_s": round(V_step * 1e3 / ms_per_step, 2),
        # network evaluations including the padding of the 16-sample MFMA tiles, reported separately
        "network_evaluations_per_s_M": round(total_evals / elapsed / 1e6, 2),
        # `value` counts the samples that reach a ray's compositing sum (deterministic; what the reference's per-ray schedule
        # emits).  The kernel also evaluates the ones a ray queues behind its terminating sample (never used, timing-dependent):
        "useful_msamples_s": round(total_composited / elapsed / 1e6, 2),
        "evaluated_msamples_s": round(total_samples / elapsed / 1e6, 2),
        "evaluated_over_composited": round(total_samples / max(total_composited, 1), 5),
        "ms_per_frame": round(ms_per_step / V_step, 4),
        # one render_frame call of the reference's API = one view per launch, nothing else on the chip
        "single_view_ms": round(single_view_ms, 4),
        "frames_per_s_single": round(1e3 / single_view_ms, 2),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config {config}: synthetic Lego-like scene {res}, hash grid L=16 F=2 T=2^19 base 16, "
                               "density MLP 32-64-16 + rgb MLP 32-64-64-16, SH-4, "
                               + (f"{CONFIG5_REQUESTS} camera requests per step" if replica else "8 orbit cameras"),
                   "samples_per_frame": None,
                   "parallelism": (f"replica{world}" if replica else f"tile{world}"), "views_per_step": V_step,
                   "step": (f"{V_step} frames per step, each tile-sharded over the {world} rank(s)" if not replica else
                            f"{V_step} requests per step, {V} whole frames per rank"),
                   "views_per_rank_and_step": V, "steps_in_flight": depth,
                   "gather": (args.gather_format if dist_on else None),
                   "gather_root": (("step % N" if args.gather_root == "rotate" else "rank 0") if dist_on else None)},
        "distributed": {"world_size": (dist.get_world_size() if dist_on else 1),
                        "backend": (dist.get_backend() if dist_on else None),
                        # True: ONE rank running the whole exchange leg (--force-dist): what a one-GPU box can execute of N > 1
                        "forced_single_rank": bool(dist_on and world == 1),
                        "launcher": ("bench.py self-launch" if os.environ.get("NRF_BENCH_SELF_LAUNCHED") else
                                     ("external" if world > 1 else None)),
                        "devices": devices,
                        # row r = rank r's device against every visible device (1: hipDeviceCanAccessPeer or itself)
                        "peer_access": peer_rows, "visible_devices": n_vis,
                        "rccl_version": rccl_version},
        "roofline": {
            "kernel": "render_persistent_kernel",
            # the contract's figure: ALGORITHMIC gather bytes / kernel time against the HBM peak.  The table
            # is served mostly from L2 / Infinity Cache (the 24 MB reference-order part and the copies of the coarse levels; the 4.5 GB of
            # copies of levels 8..11 come from HBM), so this is a cache-gather rate: see hbm_gbs_measured and limiter
            "bound": "hbm",
            "bound_note": "the contract's figure: ALGORITHMIC gather bytes (512 B per composited sample) over the launch time, against the HBM "
                          "peak; most of it is served from L2 / Infinity Cache (hbm_gbs_measured), what binds the kernel is in `limiter`",
            "binding_unit": ((pmc.get("limiter") or {}).get("binding_unit") if pmc else None),
            "achieved": round(gather_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(gather_gbs / HBM_PEAK_GBS, 5),
            # Gather lane-addresses per second over what 256 texture addressers take at the measured clock (no PMC needed): the bound
            # that bound the kernel through round 5 (0.76 with 128 addresses per sample); the cell-major quad copies of round 6 read a
            # level as two 16-byte entries instead of eight 4-byte ones (56 addresses per sample), and what limits the kernel now is
            # the VALU issue port and the overlap of memory latency (`binding_unit`).  `frac` above can exceed what HBM could deliver -- 0.99 for config 4, 1.2 for an F = 8 grid
            # (`configs`) -- because the table is served from L2 / Infinity Cache and an aligned 8- or 16-byte entry costs one
            # address like a 4-byte one: bytes per address change, addresses per second do not.
            "gather_addr_frac": round(addr_per_s / ta_peak, 5) if ta_peak else None,
            "gather_addr_per_s": round(addr_per_s / 1e9, 2), "gather_addr_unit": f"G lane-addresses/s (evaluated samples x {gather_addr_sample} / isolated launch time)",
            "gather_addresses_per_sample": gather_addr_sample,
            "grid_device_mb": grid_device_mb,
            "gather_addresses_note": "8 per level read corner by corner (4-byte entries), 2 per level read as two aligned 16-byte quads from its "
                                     "cell-major copy; 128 = no copies (the reference's table alone)",
            "gather_addr_peak": round(ta_peak / 1e9, 2) if ta_peak else None,
            "gather_addr_peak_is": f"{TA_UNITS} texture addressers x {TA_ADDR_PER_CLK:g} lane addresses per clock x the measured clock",
            "shader_clock_mhz_measured": round(clock_hz / 1e6, 1) if clock_hz else None,
            # boxes of one pool run the same launch at different sustained clocks (2.02-2.25 GHz seen in round 6) and `value` follows:
            # the clock-normalised figure is what compares across boxes and rounds
            "msamples_s_per_ghz": round(msamples_s / (clock_hz / 1e9), 1) if clock_hz else None,
            "traffic": traffic,
            "from_committed_profile": bool(pmc),
            "traffic_source": (f"{PMC_FILE.relative_to(ROOT)} (rocprofv3 --pmc passes of this command, kernel sources "
                               f"{pmc['kernel_source_sha16']})") if pmc else None,
            "hbm_gbs_measured": round(traffic / mean_kern_s / 1e9, 1) if traffic else None,
            "hbm_frac_measured": round(traffic / mean_kern_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "algorithmic_bytes_per_launch": int(mean_samples_launch * BYTES_PER_SAMPLE),
            "kernel_ms": round(mean_kern_s * 1e3, 4),
            "samples_per_launch": int(mean_samples_launch),
            # a step of more than NRF_MAX_VIEWS views per rank is several back-to-back launches, timed as one
            "launches_per_step": -(-V // nh.NRF_MAX_VIEWS),
            # `achieved` is per launch while ~launches_in_flight launches share the chip; the chip-level figures:
            "launches_in_flight": round(in_flight, 2),
            "aggregate_achieved": round(gather_gbs * in_flight, 2),
            "aggregate_frac": round(gather_gbs * in_flight / HBM_PEAK_GBS, 5),
            "isolated_kernel_ms": round(iso_kern_s * 1e3, 4),
            "isolated_frac": round(mean_samples_launch * BYTES_PER_SAMPLE / iso_kern_s / 1e9 / HBM_PEAK_GBS, 5),
            # MFMA share of the render kernel: from this run's sample count (composited samples x 20 480 FLOP: the padding of the
            # 16-sample tiles and the samples evaluated behind a ray's end are not in it) ...
            "mfma_tflops_aggregate": round(msamples_s * 1e6 * FLOP_PER_SAMPLE / 1e12 / max(world, 1), 3),
            # ... and from the committed counters of the same command (north_star: "rocprof showing achieved MFMA utilisation"):
            # SQ_VALU_MFMA_BUSY_CYCLES / all SIMD cycles, FLOPs from SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 over the profiled kernel time
            "mfma_busy_frac": ((pmc.get("mfma") or {}).get("mfma_busy_frac") if pmc else None),
            "mfma_tflops_from_counters": ((pmc.get("mfma") or {}).get("mfma_tflops_from_counters") if pmc else None),
            "mfma_frac_of_peak_from_counters": ((pmc.get("mfma") or {}).get("mfma_frac_of_2p5_pflops") if pmc else None),
            # what the counters say binds the kernel (profiles/r02: the issue-rate microbenchmark + PMC passes)
            "limiter": (pmc.get("limiter") if pmc else None),
            "valu_insts_per_launch": valu,
        },
    }
    if check is not None:
        out["sharded_frame_equals_unsharded"] = check
    if world == 1:
        out["config"]["samples_per_frame"] = int(mean_samples_launch / V)
        if not replica and not args.no_extras and not dist_on:
            out["api"] = api_bench(nh, torch, dev, desc, cam, poses, W, H)
            out["fast_interp"] = fast_interp_bench(nh, torch, dev, desc, cams_step, [poses[j] for j in step_poses(0)], W, H, V, ms_per_step / V_step)
            out["march_fast_forward"] = march_ff_bench(nh, torch, dev, desc, cams_step, [poses[j] for j in step_poses(0)], W, H, V)
            with torch.cuda.stream(stream):
                out["mlp_kernel"] = mlp_microbench(ctx, torch, dev)
            if pmc and pmc.get("mlp_forward_kernel"):  # counter side of the same kernel at 2^24 samples (committed --pmc pass)
                m = pmc["mlp_forward_kernel"]
                out["mlp_kernel"]["counters"] = {k: m.get(k) for k in ("mfma_busy_frac", "mfma_tflops_from_counters", "mfma_frac_of_2p5_pflops",
                                                                      "mfma_flops_from_counters", "algorithmic_flops", "effective_clock_ghz",
                                                                      "kernel_ms_profiled", "mfma_flops_counter")}
                out["mlp_kernel"]["counters"]["source"] = str(PMC_FILE.relative_to(ROOT))
            if pmc and pmc.get("mlp_forward_kernel_register_resident"):  # ... and of its register-resident loop (2^22 samples x 64)
                m = pmc["mlp_forward_kernel_register_resident"]
                cr = {k: m.get(k) for k in ("mfma_busy_frac", "mfma_tflops_from_counters", "mfma_frac_of_2p5_pflops", "effective_clock_ghz",
                                            "kernel_ms_profiled", "mfma_busy_cycles_per_mfma")}
                # The matrix pipe's share of the SIMD cycles -- what "MFMA utilisation" means per clock -- and the same as a rate: the
                # 2.5 PFLOP/s peak is 1024 SIMDs x 1024 FLOP per cycle at 2.4 GHz, a clock the chip does not hold under this load
                clk = m.get("effective_clock_ghz") or 0.0
                cr["peak_tflops_at_measured_clock"] = round(1024 * 1024 * clk * 1e9 / 1e12, 1) if clk else None
                cr["reading"] = ("per clock the register-resident loop keeps the matrix pipe mfma_busy_frac busy (north_star's 0.70 is exceeded "
                                 "there); against the 2.4 GHz datasheet peak it is mfma_frac_of_2p5_pflops, because the chip runs this loop at "
                                 "effective_clock_ghz.  The HBM-fed launch (`counters`) is the product form: target_met refers to it")
                cr["source"] = str(PMC_FILE.relative_to(ROOT))
                out["mlp_kernel"]["counters_register_resident"] = cr
            t0 = time.perf_counter()
            out["configs"] = configs_bench(nh, torch, dev, desc)
            out["configs"]["wall_s"] = round(time.perf_counter() - t0, 1)
            out["server"] = server_bench()
            if not args.no_cpu_baseline:
                out["cpu_baseline"], out["parity"] = cpu_baseline(nh, dev, desc, cam, poses[0], W, H, args.cpu_sample_div)
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    print(json.dumps(out), flush=True)
    if dist_on:
        os.dup2(2, 1)  # (and whatever the teardown prints)
        dist.destroy_process_group()


def configs_bench(nh, torch, dev, desc2):
    """BASELINE.json configs[3..4] on this GPU, in the driver's own run (they are parity-test cases, not the headline): per
    entry one launch shape, device time per launch from the launch's own HIP events (nrf_stats.render_ms), the composited
    and evaluated sample counts, and the contract's roofline fraction (512 B per composited sample over the launch time
    against the HBM peak -- a cache-gather rate like the headline's, except where the table leaves the caches)."""
    import tempfile

    import numpy as np

    import models
    import synthetic as syn

    def run(desc, W, H, V, opts=None, reps=4, radius=4.0311):
        c = nh.NerfHip(dev.index)
        c.load_model(desc)
        if opts is not None:
            c.set_options(opts)
        c.set_resolution(W, H)
        c.set_max_views(V)
        cams = np.stack([syn.default_camera(W, H)] * V)
        poses = np.stack([syn.orbit_pose(360.0 * i / V, 25.0, radius=radius) for i in range(V)])
        st = torch.cuda.Stream(dev)
        ms = []
        for i in range(reps + 2):
            c.render_views(cams, poses, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append(float(c.stats().render_ms))
        stt = c.stats()
        c.close()
        t = float(np.mean(ms[2:])) * 1e-3
        comp, ev = int(stt.n_composited), int(stt.n_samples)
        # gathered bytes and lane addresses per sample of THIS model: levels x 8 corners x (F fp16 values | one address)
        addrs, bytes_ = int(stt.gather_addresses_per_sample), int(desc.n_levels) * 8 * 2 * int(desc.n_features_per_level)
        clk = float(stt.shader_clock_mhz) * 1e6
        return {"views_per_launch": V, "resolution": f"{W}x{H}", "ms_per_launch": round(t * 1e3, 4), "ms_per_view": round(t * 1e3 / V, 4),
                "frames_per_s": round(V / t, 1), "msamples_s": round(comp / t / 1e6, 1), "evaluated_msamples_s": round(ev / t / 1e6, 1),
                "samples_per_view": comp // V, "evaluated_over_composited": round(ev / max(comp, 1), 4),
                "gather_bytes_per_sample": bytes_, "gather_addresses_per_sample": addrs,
                "frac": round(comp * bytes_ / t / 1e9 / HBM_PEAK_GBS, 4),
                "gather_addr_frac": round(ev * addrs / t / (TA_UNITS * TA_ADDR_PER_CLK * clk), 4) if clk > 0 else None,
                "shader_clock_mhz": round(clk / 1e6, 1) if clk > 0 else None}

    out = {"what": "BASELINE.json configs[3] (real-captured-scene SHAPE: bound 16, five cascades, 1024 samples per ray; synthetic stand-in, "
                   "the reference ships no scene) and configs[4] (64 requests of 800x800) on ONE GPU; device time per launch, `frac` = "
                   "gathered bytes per sample x composited samples / time / 8 TB/s as in `roofline` (a cache-gather rate: it may exceed "
                   "what HBM could deliver), `gather_addr_frac` = lane addresses per second / (256 texture addressers x 4 per clock x "
                   "the clock the launch measured): the bound that binds, and the one that stays below 1"}
    o4 = nh.default_options()
    o4.max_steps = 1024
    desc4, keep4, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
    out["config4_bound16_5cascades_1024steps"] = run(desc4, WIDTH, HEIGHT, DEFAULT_VIEWS, o4)
    del desc4, keep4
    # the same shape from a snapshot in instant-ngp's own layout (aabb_scale 32: Morton-ordered fp16 density grid of six cascades,
    # params_binary, per_level_scale derived from aabb_scale), written to a temp file and read back through the loader
    with tempfile.TemporaryDirectory() as td:
        pls = nh.default_per_level_scale(32.0, 16, 16)
        dn, kn, cfgn = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0, per_level_scale=pls)
        f = os.path.join(td, "scene_ngp.msgpack")
        syn.write_ngp_snapshot(f, cfgn, kn[0], kn[1], 32)
        del dn, kn
        t0 = time.perf_counter()
        desc_ngp, keep_ngp = nh.desc_from_config(syn.read_snapshot(f))
        load_s = time.perf_counter() - t0
        size_mb = os.path.getsize(f) / 1e6
    e = run(desc_ngp, WIDTH, HEIGHT, DEFAULT_VIEWS, o4)
    e.update({"snapshot_mb": round(size_mb, 1), "snapshot_load_s": round(load_s, 3)})
    out["config4_instant_ngp_layout_snapshot"] = e
    del desc_ngp, keep_ngp
    # T = 2^22: a 158 MB table -- the case that leaves the L2s and leans on Infinity Cache / HBM ("stresses hash-grid HBM path")
    desc22, keep22, _ = models.build_model(log2_hashmap_size=22, H=128)
    e = run(desc22, WIDTH, HEIGHT, DEFAULT_VIEWS)
    e["table_mb"] = 158
    out["config2_scene_table_2p22"] = e
    del desc22, keep22
    # the GRID instances: the same number of features (32) in wider entries -- an F = 8 entry is ONE aligned 16-byte gather, so a
    # sample needs 32 lane addresses instead of 128 for the same 512 bytes: bytes per second ("frac") go up past the HBM peak,
    # addresses per second do not.  (T/.../grid.h:1403-1411: n_features_per_level in {1, 2, 4, 8})
    for name, kw in (("grid_F8x4_levels", dict(n_features_per_level=8, n_levels=4)),
                     ("grid_F4x8_levels", dict(n_features_per_level=4, n_levels=8))):
        dg, kg, _ = models.build_model(log2_hashmap_size=19, H=128, **kw)
        out[name] = run(dg, WIDTH, HEIGHT, DEFAULT_VIEWS)
        del dg, kg
    # config 5: 64 camera requests of 800x800 in ONE launch (the render_server's batch)
    out["config5_64_requests_800x800"] = run(desc2, CONFIG5_RES, CONFIG5_RES, CONFIG5_REQUESTS, reps=3)
    return out


def server_bench():
    """BASELINE configs[4] through the REAL render_server (nerf-cuda_amd/host/render_server: the reference's wire protocol on TCP,
    one queue + worker per GPU, 64-view batches, two in flight): 64 concurrent clients, each with one 800x800 request in
    flight, 40 requests per client after two warm-up ones -- scripts/server_bench.py as a child process (its own server and
    client threads; this process is idle meanwhile).  Never fails the bench: a problem is reported in the object."""
    try:
        r = subprocess.run([sys.executable, str(ROOT / "scripts" / "server_bench.py"), "--clients", "64", "--requests", "40", "--port", "23611"],
                           capture_output=True, text=True, timeout=150)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:400]}


def api_bench(nh, torch, dev, desc, cam, poses, W, H):
    """The host end of the path: what a caller of the reference's API sees.  NerfRender::render_frame returns HOST memory
    (8-bit rgb + depth, R/src/nerf_render.cu:345-359); its counterpart here is nrf_render_host_u8 (the kernel writes the
    8-bit Image, the copy engine brings the rows of the region of interest into pinned host memory, the calling thread
    fills the background rows meanwhile).  Wall-clock around the C-ABI call, through ctypes."""
    import numpy as np

    g = nh.NerfHip(dev.index)
    g.load_model(desc)
    g.set_resolution(W, H)
    V = DEFAULT_VIEWS
    g.set_max_views(V)
    cam1 = np.ascontiguousarray(cam, np.float32).reshape(1, 4)
    p1 = [np.ascontiguousarray(p, np.float32).reshape(1, 16) for p in poses]
    camV = np.ascontiguousarray(np.stack([cam] * V), np.float32)
    pV = [np.ascontiguousarray(np.stack([poses[(i * V + v) % len(poses)] for v in range(V)]), np.float32).reshape(V, 16)
          for i in range(2)]
    for i in range(6):
        g.render_host_u8_raw(cam1, p1[i % len(p1)])
    n1 = 48
    wall, devms, copied = [], [], []
    for i in range(n1):
        t0 = time.perf_counter()
        f = g.render_host_u8_raw(cam1, p1[i % len(p1)])
        wall.append((time.perf_counter() - t0) * 1e3)
        devms.append(float(f.render_ms))
        copied.append(int(f.copied_bytes))
    for i in range(2):
        g.render_host_u8_raw(camV, pV[i % 2])
    nb = 12
    t0 = time.perf_counter()
    bdev, bcopied = [], []
    for i in range(nb):
        f = g.render_host_u8_raw(camV, pV[i % 2])
        bdev.append(float(f.render_ms))
        bcopied.append(int(f.copied_bytes))
    batch_ms = (time.perf_counter() - t0) * 1e3 / nb
    # two batches in flight: the copy (and the caller) of batch k under the render of batch k + 1
    tickets = [g.submit_host_u8(camV, pV[0])]
    t0 = time.perf_counter()
    for i in range(1, nb + 1):
        tickets.append(g.submit_host_u8(camV, pV[i % 2]))
        g.lib.nrf_wait_host_u8(g.h, tickets[i - 1], None)
    pipe_ms = (time.perf_counter() - t0) * 1e3 / nb
    g.lib.nrf_wait_host_u8(g.h, tickets[-1], None)
    # the after-the-fact path of round 2 for comparison: float planes, then nrf_read_u8 (quantise launch + blocking copies)
    old = []
    for i in range(8):
        t0 = time.perf_counter()
        g.render(cam, poses[i % len(poses)])
        g.read_u8()
        old.append((time.perf_counter() - t0) * 1e3)
    g.close()
    # a plain pinned D2H copy of the same size as one frame's planes, for the link's rate
    src = torch.empty((W * H * 4,), dtype=torch.uint8, device=dev)
    dst = torch.empty((W * H * 4,), dtype=torch.uint8, pin_memory=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(10):
        dst.copy_(src, non_blocking=True)
    e1.record()
    torch.cuda.synchronize(dev)
    link_gbs = 10 * W * H * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    w, d = float(np.mean(wall)), float(np.mean(devms))
    bd = float(np.mean(bdev))
    return {"what": "the reference's render_frame() ends in host memory (8-bit rgb + depth): nrf_render_host_u8, wall-clock per C-ABI call",
            "render_frame_host_u8_ms": round(w, 4),
            "render_frame_host_u8_ms_median": round(float(np.median(wall)), 4),
            "render_frame_device_ms": round(d, 4),
            "render_frame_host_tail_ms": round(w - d, 4),
            "frames_per_s_render_frame": round(1e3 / w, 2),
            "host_copy_bytes_per_frame": int(np.mean(copied)),
            "frame_bytes": W * H * 4,
            "render_frames_host_u8_ms_per_frame": round(batch_ms / V, 4),
            "render_frames_device_ms_per_frame": round(bd / V, 4),
            "render_frames_host_u8_pipelined_ms_per_frame": round(pipe_ms / V, 4),
            "frames_per_s_host_u8_batched": round(V * 1e3 / batch_ms, 2),
            "frames_per_s_host_u8_pipelined": round(V * 1e3 / pipe_ms, 2),
            "views_per_batch": V,
            # the link's rate (a plain pinned device-to-host copy of one frame's size); the path's own copies run beside the
            # render (finished strip rows are copied while the rest renders), so what a batch still waits for is the tail:
            "d2h_gbs": round(link_gbs, 2),
            "host_tail_ms_per_batch": round(batch_ms - bd, 4),
            "copy_ms_per_batch_at_link_rate": round(float(np.mean(bcopied)) / (link_gbs * 1e9) * 1e3, 4),
            "render_plus_read_u8_ms": round(float(np.mean(old[2:])), 4)}


def fast_interp_bench(nh, torch, dev, desc, cams, poses, W, H, V, base_ms_per_frame):
    """nrf_options::fast_interp (OPT-IN, not the headline: the default stays bit-exact): the step's launch with the
    single-rounding interpolation -- ms per frame, samples/s, and its distance from the default frame."""
    import numpy as np

    g = nh.NerfHip(dev.index)
    g.load_model(desc)
    g.set_resolution(W, H)
    g.set_max_views(V)
    st = torch.cuda.Stream(dev)
    res = {}
    for fast in (0, 1):
        o = nh.default_options()
        o.fast_interp = fast
        g.set_options(o)
        ms = []
        for i in range(6):
            g.render_views(cams, poses, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append(float(g.stats().render_ms))
        stt = g.stats()
        res[fast] = (float(np.mean(ms[2:])), int(stt.n_samples), g.read_view_f32(0)[0])
    g.close()
    mse = float(np.mean((res[1][2].astype(np.float64) - res[0][2].astype(np.float64)) ** 2))
    return {"what": "OPT-IN nrf_options.fast_interp = 1 (one rounding per corner of the hash-grid interpolation instead of the reference's "
                    "three: not bit-exact, default off); same launch as the headline step, device time",
            "ms_per_frame": round(res[1][0] / V, 4), "default_ms_per_frame_same_run": round(res[0][0] / V, 4),
            "msamples_s": round(res[1][1] / (res[1][0] * 1e-3) / 1e6, 2),
            "speedup": round(res[0][0] / res[1][0], 4),
            "psnr_db_vs_default_frame": round(99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse), 2),
            "max_abs_vs_default_frame": float(np.abs(res[1][2] - res[0][2]).max()),
            "tolerance": "features within 4 x 2^-11 of the bit-exact ones, frames <= 2/255 (tests/test_parity_gpu.py)"}


def march_ff_bench(nh, torch, dev, desc, cams, poses, W, H, V):
    """The barrier fast-forward of the march (nrf_device.h; on by default, part of the headline) against a context created
    with NRF_MARCH_FF=0, which simulates every trip ahead of a ray's first possible sample as rounds 1-2 did: the same
    launch as the headline step, device time, and whether the frames are the same bits."""
    import os

    import numpy as np

    res = {}
    for ff in ("0", "1"):
        saved = os.environ.get("NRF_MARCH_FF")
        os.environ["NRF_MARCH_FF"] = ff
        try:
            g = nh.NerfHip(dev.index)  # (the switch is read at nrf_create)
        finally:
            if saved is None:
                os.environ.pop("NRF_MARCH_FF", None)
            else:
                os.environ["NRF_MARCH_FF"] = saved
        g.load_model(desc)
        g.set_resolution(W, H)
        g.set_max_views(V)
        st = torch.cuda.Stream(dev)
        ms = []
        for i in range(6):
            g.render_views(cams, poses, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append(float(g.stats().render_ms))
        frames = [g.read_view_f32(v) for v in (0, V - 1)]
        res[ff] = (float(np.mean(ms[2:])), int(g.stats().n_composited), frames)
        g.close()
    same = res["0"][1] == res["1"][1] and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(res["0"][2], res["1"][2]))
    return {"what": "exact shortcut of the march, default ON (the headline includes it): a ray steps straight to its last barrier plane "
                    "ahead of its first possible sample instead of simulating the reference's cell trips there; against NRF_MARCH_FF=0 "
                    "on the same launch",
            "ms_per_frame": round(res["1"][0] / V, 4), "ms_per_frame_every_trip_simulated": round(res["0"][0] / V, 4),
            "speedup": round(res["0"][0] / res["1"][0], 4), "frames_bit_identical": bool(same),
            "tests": "tests/test_persistent_gpu.py (both schedulings, 1-5 cascades), tests/test_barrier_lemma.py (CPU, against the reference's trip loop)"}


def mlp_microbench(ctx, torch, dev):
    """The fused-MLP stage kernel alone on resident fp16 inputs: the MFMA-roofline figure (north star: >= 0.70 -- not met:
    see target_frac / the ceilings below)."""
    st = torch.cuda.current_stream(dev)
    assert st.cuda_stream != 0

    def timed(f, reps, warm=3):
        for _ in range(warm):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            f()
        e1.record(st)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    sizes = {}
    for log2n in (22, 24):  # 2^22: what rounds 1-2 quoted (436 MB of rows, partly cache-resident); 2^24: steady state from HBM
        n = 1 << log2n
        feat = (torch.rand((n, 32), device=dev) - 0.5).half()
        dirf = (torch.rand((n, 16), device=dev) - 0.5).half()
        out = torch.empty((n, 4), dtype=torch.float16, device=dev)
        torch.cuda.synchronize(dev)
        ms = timed(lambda: ctx.mlp_forward(feat.data_ptr(), dirf.data_ptr(), n, out.data_ptr(), stream=st.cuda_stream), 20 if log2n == 22 else 8)
        sizes[log2n] = (n, ms)
        if log2n == 22:
            # the same kernel with every chunk evaluated 64 times from registers: the MFMA chain (with its fp32 -> fp16
            # re-packing between layers) without the HBM stream
            rep = 64
            core_ms = timed(lambda: ctx.mlp_forward_repeat(feat.data_ptr(), dirf.data_ptr(), n, out.data_ptr(), rep, stream=st.cuda_stream), 4, 1)
            core = n * rep * FLOP_PER_SAMPLE / (core_ms * 1e-3) / 1e12
        del feat, dirf, out
    n, ms = sizes[22]
    n24, ms24 = sizes[24]
    tflops = n * FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12
    tflops24 = n24 * FLOP_PER_SAMPLE / (ms24 * 1e-3) / 1e12
    return {"kernel": "mlp_forward_kernel", "samples": n, "ms": round(ms, 4), "bound": "mfma",
            "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / MFMA_PEAK_TFLOPS, 4), "hbm_gbs": round(n * 104 / (ms * 1e-3) / 1e9, 1),
            "steady_state": {"samples": n24, "ms": round(ms24, 4), "achieved": round(tflops24, 2), "frac": round(tflops24 / MFMA_PEAK_TFLOPS, 4),
                             "hbm_gbs": round(n24 * 104 / (ms24 * 1e-3) / 1e9, 1)},
            "register_resident_tflops": round(core, 2), "register_resident_frac": round(core / MFMA_PEAK_TFLOPS, 4),
            # the north star's target and why it is out of reach for this network shape on this ISA (DESIGN.md mlp_forward_kernel):
            "target_frac": 0.70, "target_met": bool(core / MFMA_PEAK_TFLOPS >= 0.70),
            "ceiling_hbm_fed_frac": 0.50,          # 104 B/sample at ~6.3 TB/s achievable: 1.24 PFLOP/s
            "ceiling_no_repack_frac": 0.70,        # the MFMA chain with NO re-packing work between layers (scripts/mlp_probe, profiles/r02/mlp_probe.txt)
            "ceiling_pure_mix_frac": 0.62,         # one MFMA + four half-rate conversions, the shape's instruction mix (profiles/r02/issue_rate.txt)
            "interleave_experiment": "forced MFMA / VALU issue patterns (sched_group_barrier): none beats the compiler's schedule, profiles/r03/mlp_interleave.txt"}


def usable_cpus(cgroup_root="/sys/fs/cgroup"):
    """The CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (the one-GPU box shows 256
    logical CPUs and grants 16: more threads than the quota only take turns -- 128 threads measured 1.9 Msamples/s where 16
    give 2.9, profiles/r05/cpu_threads.txt).  cgroup v2: `cpu.max` = "<quota> <period>" or "max <period>"; v1:
    cpu/cpu.cfs_quota_us (-1: none) and cpu/cpu.cfs_period_us."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    root = Path(cgroup_root)
    try:
        quota, period = (root / "cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:  # noqa: BLE001  (cgroup v1, or none)
        try:
            q = int((root / "cpu" / "cpu.cfs_quota_us").read_text())
            p = int((root / "cpu" / "cpu.cfs_period_us").read_text())
            if q > 0 and p > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


def cpu_baseline(nh, dev, desc, cam, pose, W, H, div):
    """The CPU oracle on this host's cores, on a (W/div)x(H/div) frame of the same view -- and, since the oracle's
    frame is there anyway, the image-quality leg of the metric: PSNR / max |d| of the HIP frame against it.
    The timed form is the oracle's PER_RAY schedule with every ray run to its end on its own (dynamic OpenMP schedule over
    rays, no per-round barrier, F16C conversions): the schedule the HIP kernel implements, the same sample count as `value`
    (composited samples), and what a CPU implementation of the path would do.  The reference's own global round schedule
    (nerf_render.cu:269-338: a barrier and a serial compaction per round, 8.6 % of the evaluations behind a ray's end) is
    timed beside it on a quarter-size frame, and so is the per-ray form on 8 threads -- the core scaling."""
    import numpy as np
    import oracle_py as op

    w, h = max(8, W // div), max(8, H // div)
    c = np.array(cam, np.float32) / np.float32(div)
    o = op.Oracle(desc)
    threads = min(int(op.lib().nrfo_max_threads()), usable_cpus())  # what OpenMP would start, capped by what the cgroup grants
    t0 = time.perf_counter()
    want, want_depth, st, counts, hashes = o.render_rays(c, pose, w, h, schedule=op.SCHED_PER_RAY, n_threads=threads)
    dt = time.perf_counter() - t0
    n = int(st.n_samples)
    # a quarter-size frame of the same camera: 8 threads (the core scaling) and the reference's round schedule on all cores
    w2, h2 = max(8, w // 2), max(8, h // 2)
    c2 = c / np.float32(2)
    t0 = time.perf_counter()
    _, _, st8 = o.render(c2, pose, w2, h2, schedule=op.SCHED_PER_RAY, n_threads=min(8, threads))
    dt8 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, st2 = o.render(c2, pose, w2, h2, schedule=op.SCHED_PER_RAY, n_threads=threads)
    dt2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, str_ = o.render(c2, pose, w2, h2, schedule=op.SCHED_REFERENCE, n_threads=threads)
    dtr = time.perf_counter() - t0
    t8 = min(8, threads)
    rate, rate8, rate2 = n / dt, int(st8.n_samples) / dt8, int(st2.n_samples) / dt2
    frac = "quarter" if div == 2 else f"1/{div * div}"
    base = {"value": round(rate / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "host_logical_cpus": os.cpu_count(), "cores_note": "threads used = min(OpenMP's default, the cgroup's CPU quota)",
            "sample": f"one {w}x{h} frame ({frac} of the {W}x{H} frame's pixels) of the same camera "
                      f"({n} composited samples, {dt:.1f} s), every ray run to its end on its own (the per-ray schedule the HIP "
                      f"kernel implements), OpenMP dynamic over rays",
            "frames_per_s_1080p_equiv": round(1.0 / (dt * div * div), 5),
            "fp16_conversions": op.fp16_backend(),
            # what one core does: thread-seconds per sample (the oracle is a bit-exact restatement, not a tuned renderer:
            # scalar code, 128 table gathers + ~10 k multiply-adds per sample)
            "us_per_sample_and_core": round(threads * dt / n * 1e6, 3),
            "core_scaling": {"frame": f"{w2}x{h2}", "threads_lo": t8, "msamples_s_lo": round(rate8 / 1e6, 4),
                             "us_per_sample_and_core_lo": round(t8 * dt8 / max(int(st8.n_samples), 1) * 1e6, 3),
                             "threads_hi": threads, "msamples_s_hi": round(rate2 / 1e6, 4),
                             "parallel_efficiency": round((rate2 / threads) / (rate8 / t8), 3)},
            "reference_round_schedule": {"frame": f"{w2}x{h2}", "msamples_s": round(int(str_.n_samples) / dtr / 1e6, 4),
                                         "evaluated_samples": int(str_.n_samples), "rounds": int(str_.n_rounds),
                                         "note": "nerf_render.cu:269-338 as written: a barrier + serial compaction per round"}}
    g = nh.NerfHip(dev.index)
    g.load_model(desc)
    g.set_resolution(w, h)
    g.render(c, pose)
    got, got_depth = g.read_f32()
    n_hip = int(g.stats().n_composited)

    def dist_of(a, b):
        mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        return {"psnr_db": round(99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse), 2), "max_abs": float(np.abs(a - b).max())}

    parity = {"against": f"CPU oracle (port of the reference path), same {w}x{h} frame, float RGBA",
              **dist_of(got, want), "max_abs_depth": float(np.abs(got_depth - want_depth).max()),
              "composited_samples_hip": n_hip, "composited_samples_oracle": n,
              "tolerance": "max_abs <= 2/255 and PSNR >= 45 dB (tests/test_parity_gpu.py)"}
    # The reference accumulates its MLP products in fp16 WMMA fragments (T/src/fully_fused_mlp.cu:69,334,437); HIP path and
    # oracle accumulate in fp32.  The distance to the reference's own arithmetic, measured against the oracle's emulation of
    # that accumulator: K16 = fp16 running sum rounded once per 16-wide K block (one mma_sync, the reference's granularity) on
    # the SAME frame; STEP = rounded after every product (the pessimistic bound) on a 1/16-size frame of the same camera.
    t0 = time.perf_counter()
    k16, _, _ = op.Oracle(desc, accumulate=op.ACC_FP16_K16).render(c, pose, w, h, schedule=op.SCHED_PER_RAY, n_threads=threads)
    w4, h4 = max(8, w // 4), max(8, h // 4)
    c4 = c / np.float32(4)
    step4, _, _ = op.Oracle(desc, accumulate=op.ACC_FP16_STEP).render(c4, pose, w4, h4, schedule=op.SCHED_PER_RAY, n_threads=threads)
    fp32_4, _, _ = o.render(c4, pose, w4, h4, schedule=op.SCHED_PER_RAY, n_threads=threads)
    g.set_resolution(w4, h4)
    g.render(c4, pose)
    got4, _ = g.read_f32()
    g.close()
    parity["vs_fp16_accumulate"] = {
        "what": "HIP frame (fp32 MFMA accumulation) against the oracle emulating the reference's fp16 WMMA accumulators",
        "k16": {**dist_of(got, k16), "frame": f"{w}x{h}", "tolerance": "max_abs <= 1/255 and PSNR >= 72 dB"},
        "step": {**dist_of(got4, step4), "frame": f"{w4}x{h4}", "tolerance": "max_abs <= 2/255 and PSNR >= 65 dB"},
        "oracle_fp32_vs_oracle_k16": dist_of(want, k16),
        "hip_vs_oracle_fp32_small_frame": dist_of(got4, fp32_4),
        "tests": "tests/test_accumulate_modes.py (fixtures: tests/golden/accumulate_modes.npz)",
        "cpu_s": round(time.perf_counter() - t0, 1)}
    # nvcc fuses `a * b + c` by default and the reference's build does not turn it off (R/CMakeLists.txt:71-79); HIP path and
    # oracle round every operation.  This gap moves sample POSITIONS (ox + t * dx, the cell index, pos_fract), not only values:
    # the same frame from the oracle in contraction mode (nrfo_set_contract) -- frame distance, composited-sample delta, and the
    # rays whose sample set (the (dt, t - last_t) bits of every sample) differs between the two arithmetics.
    t0 = time.perf_counter()
    fused, fdepth, stf, fcounts, fhashes = op.Oracle(desc, contract=True).render_rays(c, pose, w, h, schedule=op.SCHED_PER_RAY,
                                                                                      n_threads=threads)
    sampling = int((counts > 0).sum())
    dpx = np.abs(got - fused).max(axis=-1)  # per pixel
    # which product of two a compiler fuses, and whether it fuses a product that has other uses, is its own choice: the same frame
    # with the OTHER choice at every such site (nrfo_set_contract(2)) -- the spread is the uncertainty of this emulated distance
    fused2, _, stf2, fcounts2, fhashes2 = op.Oracle(desc, contract=2).render_rays(c, pose, w, h, schedule=op.SCHED_PER_RAY, n_threads=threads)
    sensitivity = {"what": "the same comparison with the other fusion choice at every ambiguous site (right product of two fused; "
                           "`alpha * T`, which has other uses, not fused)",
                   **dist_of(got, fused2), "rays_with_other_sample_set_frac": round(int((hashes != fhashes2).sum()) / max(sampling, 1), 6),
                   "choice_1_vs_choice_2": dist_of(fused, fused2),
                   "rays_differing_between_the_choices": int((fhashes != fhashes2).sum())}
    parity["vs_fma_contract"] = {
        "what": "HIP frame (every fp32 operation rounded) against the oracle with a * b + c fused wherever the reference's device "
                "source has it in one expression (nvcc's default contraction).  AN EMULATION of the reference binary's arithmetic "
                "(parity unpinned: the reference cannot run here); `sensitivity` gives the spread over the compiler's own choices",
        "sensitivity": sensitivity,
        **dist_of(got, fused), "frame": f"{w}x{h}", "max_abs_depth": float(np.abs(got_depth - fdepth).max()),
        # a ray whose sample set changes (one occupied cell more or less at a grazing hit) changes its pixel by whatever that
        # sample weighs: max_abs is not bounded by rounding here, the share of such pixels is
        "pixels_over_1_255": int((dpx > 1.0 / 255.0).sum()), "pixels_over_1_255_frac": round(float((dpx > 1.0 / 255.0).mean()), 7),
        "abs_p9999": float(np.quantile(dpx, 0.9999)),
        "tolerance": "PSNR >= 80 dB and |d| <= 1/255 for >= 99.9 % of the pixels (max_abs: a changed sample set, not rounding)",
        "composited_samples_contracted": int(stf.n_samples), "composited_sample_delta": int(stf.n_samples) - n_hip,
        "rays_sampling": sampling, "rays_with_other_sample_count": int((counts != fcounts).sum()),
        "rays_with_other_sample_set": int((hashes != fhashes).sum()),
        "rays_with_other_sample_set_frac": round(int((hashes != fhashes).sum()) / max(sampling, 1), 6),
        "oracle_contract_vs_oracle_default": dist_of(want, fused),
        "tests": "tests/test_contract_modes.py (fixtures: tests/golden/contract_modes.npz)",
        "cpu_s": round(time.perf_counter() - t0, 1)}
    return base, parity


if __name__ == "__main__":
    main()
