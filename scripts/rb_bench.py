#!/usr/bin/env python3
"""The render-buffer chain as a stream: nrf_rb_accumulate + nrf_rb_tonemap (two passes, the reference's call shape,
R/src/render_buffer.cu:590-627) against nrf_rb_present (one pass), device time from HIP events on the launch stream, and the
rate against the ~6.3 TB/s a plain stream reaches on this chip (MI355X_MICROARCH.md).  Bytes per pixel: two passes
32 + 16 and 16 + 16 = 80 (+ a 16-byte clear for the first sample); one pass 16 + 16 read, 16 + 16 written = 64 (+ 4 with
the packed 8-bit plane).
usage: python scripts/rb_bench.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "nerf-cuda_amd"), str(ROOT / "tests")]
import torch  # noqa: E402

import nerfhip as nh  # noqa: E402

STREAM_PEAK = 6300.0  # GB/s achievable by a plain copy stream


def timed(f, st, reps=40, warm=5):
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        f()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    st = torch.cuda.Stream()
    s = st.cuda_stream
    bg = [0.3, 0.6, 0.9, 1.0]
    print("resolution  colour space / curve      two passes: us  GB/s |  one pass: us  GB/s  of 6.3 TB/s |  + rgba8: us  GB/s")
    for W, H in ((1920, 1080), (3840, 2160)):
        n = W * H
        rgba8 = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        for cs, ocs, curve, name in ((0, 0, 1, "linear / ACES / linear"), (0, 1, 1, "linear / ACES / sRGB"), (1, 1, 2, "sRGB / Hable / sRGB")):
            rb = nh.RenderBuffer(0)
            rb.resize(W, H)
            rb.set_color_space(cs)
            rb.set_tonemap_curve(curve)
            rb.accumulate(0.0)  # spp 1: steady state (the mean plane is read)

            def two():
                rb.accumulate(0.0, s)
                rb.tonemap(0.5, bg, ocs, s)

            t2 = timed(two, st)
            t1 = timed(lambda: rb.present(0.5, bg, ocs, None, s), st)
            t8 = timed(lambda: rb.present(0.5, bg, ocs, rgba8.data_ptr(), s), st)
            g2, g1, g8 = n * 80 / t2 / 1e6, n * 64 / t1 / 1e6, n * 68 / t8 / 1e6
            print(f"{W}x{H}  {name:24s}  {t2 * 1e3:8.1f} {g2:7.0f} | {t1 * 1e3:8.1f} {g1:7.0f}  {g1 / STREAM_PEAK:5.2f}       | {t8 * 1e3:8.1f} {g8:7.0f}")
            rb.close()


if __name__ == "__main__":
    main()
