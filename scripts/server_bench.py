#!/usr/bin/env python3
"""BASELINE config 5 through the real render_server: N concurrent clients, each keeping one 800x800 camera request in
flight over its own TCP connection (the reference's wire protocol: 64-byte pose in, 3*W*H bytes of rgb out), against
nerf-cuda_amd/host/render_server with one queue + worker per GPU.

Prints one JSON line: requests/s, mean batch size (views per launch), GPU-busy fraction (device time of the launches /
wall time, from the server's STAT hook), client-side latency percentiles, and a byte-equality check of sampled replies
against the binding's own render of the same pose.
  usage: python scripts/server_bench.py [--clients 64] [--requests 12] [--res 800] [--devices 0] [--port 23600]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import models
import nerfhip as nh
import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--clients", type=int, default=64)
ap.add_argument("--requests", type=int, default=12, help="requests per client (after 2 warm-up requests)")
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--devices", default="0", help='NERF_DEVICES of the server, e.g. "0" or "0,0" (two workers on one GPU)')
ap.add_argument("--mode", default="replica", choices=("replica", "tile"))
ap.add_argument("--port", type=int, default=23600)
ap.add_argument("--log2-hashmap-size", type=int, default=19)
ap.add_argument("--check", type=int, default=4, help="sampled replies compared with the binding's render")
args = ap.parse_args()

W = H = args.res
frame_bytes = 3 * W * H
desc, keep, cfg = models.build_model(log2_hashmap_size=args.log2_hashmap_size, H=128)
tmp = tempfile.mkdtemp(prefix="nrf_server_bench_")
snap = Path(tmp) / "scene.msgpack"
syn.write_snapshot(snap, cfg, keep[0], keep[1], binary="__half")
env = dict(os.environ, NRF_SERVER_TEST_HOOKS="1", NRF_SERVER_BIND="127.0.0.1", NERF_DEVICES=args.devices, NERF_SERVER_MODE=args.mode,
           NRF_SERVER_MAX_CLIENTS=str(max(256, args.clients + 8)))
srv = subprocess.Popen([os.environ.get("NRF_SERVER_BIN", str(ROOT / "nerf-cuda_amd" / "host" / "render_server")), str(args.port), str(snap), str(W), str(H)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)


def connect():
    for _ in range(600):
        try:
            s = socket.create_connection(("127.0.0.1", args.port), timeout=2.0)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4 << 20)
            s.settimeout(120)
            return s
        except OSError:
            if srv.poll() is not None:
                raise RuntimeError("render_server exited: " + (srv.stdout.read() or ""))
            time.sleep(0.1)
    raise RuntimeError("render_server did not come up")


def hook(sock, word):
    m = np.zeros(16, np.float32)
    m[:1] = np.frombuffer(word, np.float32)
    sock.sendall(m.tobytes())


def stat(sock):
    hook(sock, b"STAT")
    buf = bytearray()
    while len(buf) < 256:
        buf += sock.recv(256 - len(buf))
    w = buf.split(b"\0", 1)[0].decode().split()
    return {w[i]: float(w[i + 1]) for i in range(0, len(w), 2)}


try:
    ctl = connect()
    poses = [[syn.orbit_pose(360.0 * ((c * 7 + i * 13) % 64) / 64.0, (10.0, 30.0, 50.0)[(c + i) % 3]) for i in range(args.requests + 2)]
             for c in range(args.clients)]
    socks = [connect() for _ in range(args.clients)]
    lat = [[] for _ in range(args.clients)]
    keep_frames = {}
    errors = []
    start = threading.Barrier(args.clients + 1)

    def client(c):
        try:
            buf = bytearray(frame_bytes)
            mv = memoryview(buf)
            s = socks[c]

            def one(i):
                t0 = time.perf_counter()
                s.sendall(np.ascontiguousarray(poses[c][i], np.float32).tobytes())
                got = 0
                while got < frame_bytes:
                    n = s.recv_into(mv[got:], frame_bytes - got)
                    if n == 0:
                        raise RuntimeError("connection closed early")
                    got += n
                return time.perf_counter() - t0

            for i in range(2):
                one(i)
            start.wait()
            for i in range(2, args.requests + 2):
                lat[c].append(one(i))
                if c < args.check and i == 2 + c % max(args.requests, 1):
                    keep_frames[c] = (i, bytes(buf))
            start.wait()
        except Exception as e:  # noqa: BLE001
            errors.append((c, repr(e)))
            try:
                start.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=client, args=(c,), daemon=True) for c in range(args.clients)]
    for t in threads:
        t.start()
    start.wait()
    s0 = stat(ctl)
    t0 = time.perf_counter()
    start.wait()
    wall = time.perf_counter() - t0
    s1 = stat(ctl)
    for t in threads:
        t.join(30)
    assert not errors, errors
    frames = s1["frames"] - s0["frames"]
    batches = s1["batches"] - s0["batches"]
    all_lat = np.array([x for l in lat for x in l]) * 1e3
    # sampled replies against the binding's own render (host frames of a single view), byte for byte
    equal = None
    if keep_frames:
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        equal = True
        for c, (i, data) in keep_frames.items():
            rgb, _ = ctx.render_host_u8([cam], [poses[c][i]])
            equal = equal and bool(np.array_equal(np.frombuffer(data, np.uint8).reshape(H, W, 3), rgb[0]))
        ctx.close()
    out = {"what": f"render_server, {args.clients} concurrent clients x {W}x{H} (BASELINE config 5), devices {args.devices} ({args.mode})",
           "requests": int(frames), "wall_s": round(wall, 4), "requests_per_s": round(frames / wall, 2),
           "mean_batch_size": round(frames / max(batches, 1), 2), "launches": int(batches),
           "gpu_busy_frac": round((s1["gpu_ms"] - s0["gpu_ms"]) / (s1["wall_ms"] - s0["wall_ms"]) / max(s1["workers"], 1), 4),
           "gpu_ms_per_request": round((s1["gpu_ms"] - s0["gpu_ms"]) / max(frames, 1), 4),
           "reply_MB_per_s": round(frames * frame_bytes / wall / 1e6, 1),
           "latency_ms": {"p50": round(float(np.percentile(all_lat, 50)), 2), "p90": round(float(np.percentile(all_lat, 90)), 2),
                          "max": round(float(all_lat.max()), 2)},
           "workers": int(s1["workers"]), "sampled_replies_equal_binding": equal}
    print(json.dumps(out), flush=True)
    hook(ctl, b"QUIT")
    srv.wait(timeout=30)
finally:
    if srv.poll() is None:
        srv.kill()
