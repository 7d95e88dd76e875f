#!/usr/bin/env python3
"""Single-view host frames with two calls in flight (submit(k + 1) before wait(k)) under NRF_HOST_COLS=1 (the region of
interest's COLUMNS travel: pitched hipMemcpy2DAsync, offsets not 4-byte aligned) and =0 (whole rows): the pitched copy of
call k runs beside the persistent render of call k + 1 here, which profiles/r04/host_cols_ab.txt (serial submit + wait) did
not measure -- a pitched copy the runtime moves with a blit KERNEL finds no compute unit free beside a resident render.
Alternates the two settings on one box; prints ms per frame, serial and pipelined.
usage: python scripts/host_cols_pipelined.py [reps]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import numpy as np  # noqa: E402

import models  # noqa: E402
import nerfhip as nh  # noqa: E402
import synthetic as syn  # noqa: E402

W, H, N = 1920, 1080, 96


def make(cols):
    os.environ["NRF_HOST_COLS"] = cols
    g = nh.NerfHip(0)
    os.environ.pop("NRF_HOST_COLS", None)
    return g


def run(g, cam1, poses):
    for i in range(6):
        g.render_host_u8_raw(cam1, poses[i % len(poses)])
    t0 = time.perf_counter()
    for i in range(N):
        g.render_host_u8_raw(cam1, poses[i % len(poses)])
    serial = (time.perf_counter() - t0) * 1e3 / N
    tickets = [g.submit_host_u8(cam1, poses[0])]
    t0 = time.perf_counter()
    for i in range(1, N + 1):
        tickets.append(g.submit_host_u8(cam1, poses[i % len(poses)]))
        g.lib.nrf_wait_host_u8(g.h, tickets[i - 1], None)
    piped = (time.perf_counter() - t0) * 1e3 / N
    g.lib.nrf_wait_host_u8(g.h, tickets[-1], None)
    return serial, piped


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
    cam1 = np.ascontiguousarray(syn.default_camera(W, H), np.float32).reshape(1, 4)
    poses = [np.ascontiguousarray(syn.orbit_pose(45.0 * i, 30.0), np.float32).reshape(1, 16) for i in range(8)]
    ctxs = {}
    for cols in ("1", "0"):
        g = make(cols)
        g.load_model(desc)
        g.set_resolution(W, H)
        ctxs[cols] = g
    print(f"single 1920x1080 views to host bytes, {N} calls per figure; ms per frame")
    for rep in range(reps):
        for cols in ("1", "0"):
            s, p = run(ctxs[cols], cam1, poses)
            print(f"NRF_HOST_COLS={cols}  serial {s:.4f}  two in flight {p:.4f}", flush=True)


if __name__ == "__main__":
    main()
