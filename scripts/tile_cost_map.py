"""Diagnostic build (make prof) + NRF_MARCH_BUDGET=4095: per-tile cost (Mcycles) and start time (Mcycles after the wave entered
the tile loop) of the persistent kernel, as maps over the 8x8 tiles of one 1080p view."""
import os, pathlib, sys
os.environ["NRF_MARCH_BUDGET"] = "4095"
sys.path[:0] = ["nerf-cuda_amd", "tests"]
import numpy as np
import models, nerfhip as nh, synthetic as syn
nh.LIB_PATH = pathlib.Path("nerf-cuda_amd/libnerfhip_prof.so").resolve()
desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
c = nh.NerfHip(0); c.load_model(desc)
W, H = 1920, 1080
c.set_resolution(W, H)
cam = syn.default_camera(W, H)
for az in (0, 45):
    for _ in range(2):
        c.render(cam, syn.orbit_pose(az, 30))
    rgba, depth = c.read_f32()
    import ctypes as C
    wt = (C.c_ulonglong * 8192)()
    c.lib.nrf_debug_wave_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    if c.lib.nrf_debug_wave_times(c.h, wt, 4096) == 0:
        w = np.array(wt, dtype=np.uint64).reshape(4096, 2).astype(np.float64)
        # s_memtime is not one clock for the whole device: waves whose entry stamps lie within 0.3 Mcycles of each other
        # (sorted) share a clock domain; every domain is normalised to its own first entry
        order = np.argsort(w[:, 0])
        dom = np.zeros(4096, dtype=int)
        dom[order] = np.cumsum(np.concatenate([[0], np.diff(w[order, 0]) > 3e5]))
        t0 = np.array([w[dom == x, 0].min() for x in range(dom.max() + 1)])[dom]
        print(f"az {az}: {dom.max() + 1} clock domains, workgroups per domain {np.bincount(dom) // 16}")
        b, e = (w[:, 0] - t0) / 1e6, (w[:, 1] - t0) / 1e6
        print(f"az {az}: waves enter the tile loop at {b.min():.3f} .. {b.max():.3f} Mcycles (mean {b.mean():.3f}), leave it at "
              f"{e.min():.3f} .. {e.max():.3f} (mean {e.mean():.3f}); percentiles of the exit: " +
              " ".join(f"{q}%:{np.percentile(e, q):.2f}" for q in (5, 25, 50, 75, 90, 95, 99)))
        wg_end = e.reshape(256, 16).max(axis=1)
        print("   workgroup exit (last wave), sorted, every 16th:", " ".join(f"{v:.2f}" for v in np.sort(wg_end)[::16]))
    cost = depth[::8, 1::8]      # lane 1 of each tile: cost; lane 0: start; lane 2: samples / 1000; lane 3: index in its block
    start = depth[::8, 0::8]
    cost, start = cost[:, :240], start[:, :240]
    live = cost > 0
    print(f"az {az}: render {c.stats().render_ms:.3f} ms; live tiles {live.sum()}, cost Mcycles: mean {cost[live].mean():.3f} max {cost.max():.3f}; "
          f"latest start {start.max():.3f}, latest end {(start + cost).max():.3f}")
    end = start + cost
    order = np.argsort(end.ravel())[::-1][:12]
    for i in order:
        ty, tx = divmod(i, cost.shape[1])
        print(f"   tile ({tx:3d},{ty:3d}) start {start[ty, tx]:.3f} cost {cost[ty, tx]:.3f} end {end[ty, tx]:.3f}")
    samples = depth[::8, 2::8][:, :240] * 1e3
    bt = depth[::8, 3::8][:, :240]
    for b in range(16):
        m = live & (bt == b)
        print(f"   tile {b:2d} of its block: {m.sum():5d} tiles, {samples[m].sum()/1e6:6.3f} M samples, {cost[m].sum():7.1f} Mcycles, "
              f"{1e6*cost[m].sum()/max(samples[m].sum(),1):6.1f} cycles/sample, mean start {start[m].mean():.3f}")
    simd, wv = depth[::8, 4::8][:, :240], depth[::8, 5::8][:, :240]
    for name, key, rng in (("SIMD", simd, range(4)), ("wave of the workgroup", wv, range(16))):
        for b in rng:
            m = live & (key == b)
            print(f"   {name} {b:2d}: {m.sum():5d} tiles, {samples[m].sum()/1e6:6.3f} M samples, "
                  f"{1e6*cost[m].sum()/max(samples[m].sum(),1):6.1f} cycles/sample, mean tile index in block {bt[m].mean():.1f}")
    rows = cost.sum(axis=1)
    print("   cost per tile row (Mcycles), rows 60-90:", " ".join(f"{i}:{v:.1f}" for i, v in enumerate(rows) if 60 <= i < 90))
