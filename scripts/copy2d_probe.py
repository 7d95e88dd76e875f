"""Device-to-pinned-host copies of a frame's region of interest: whole rows (one linear copy per plane) against the columns of
the region only (hipMemcpy2DAsync, pitch = the frame's row).  1920x1080 rgb u8 + depth u8 planes, rows 100..980, 80 % / 60 % /
40 % of the width; HIP events around the copies on a non-blocking stream."""
import ctypes as C
import sys

hip = C.CDLL("libamdhip64.so")
vp, sz = C.c_void_p, C.c_size_t
hip.hipMalloc.argtypes = [C.POINTER(vp), sz]
hip.hipHostMalloc.argtypes = [C.POINTER(vp), sz, C.c_uint]
hip.hipMemcpyAsync.argtypes = [vp, vp, sz, C.c_int, vp]
hip.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, C.c_int, vp]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipEventCreate.argtypes = [C.POINTER(vp)]
hip.hipEventRecord.argtypes = [vp, vp]
hip.hipEventSynchronize.argtypes = [vp]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]


def ck(e):
    assert e == 0, e


W, H = 1920, 1080
d, h, st, e0, e1 = vp(), vp(), vp(), vp(), vp()
ck(hip.hipMalloc(C.byref(d), W * H * 4)); ck(hip.hipHostMalloc(C.byref(h), W * H * 4, 0))
ck(hip.hipStreamCreateWithFlags(C.byref(st), 1)); ck(hip.hipEventCreate(C.byref(e0))); ck(hip.hipEventCreate(C.byref(e1)))
D2H = 2
lo, hi = 100, 980


def timed(f, reps=20):
    for _ in range(3):
        f()
    ck(hip.hipEventRecord(e0, st))
    for _ in range(reps):
        f()
    ck(hip.hipEventRecord(e1, st)); ck(hip.hipEventSynchronize(e1))
    ms = C.c_float()
    ck(hip.hipEventElapsedTime(C.byref(ms), e0, e1))
    return ms.value / reps


def rows():
    ck(hip.hipMemcpyAsync(h.value + lo * W * 3, d.value + lo * W * 3, (hi - lo) * W * 3, D2H, st))
    ck(hip.hipMemcpyAsync(h.value + W * H * 3 + lo * W, d.value + W * H * 3 + lo * W, (hi - lo) * W, D2H, st))


t = timed(rows)
nb = (hi - lo) * W * 4
print(f"whole rows {lo}..{hi}: {t * 1e3:.1f} us for {nb / 1e6:.2f} MB = {nb / t / 1e6:.1f} GB/s", flush=True)
for frac in (0.8, 0.6, 0.4):
    x0 = int(W * (1 - frac) / 2) & ~3
    x1 = W - x0

    def cols():
        ck(hip.hipMemcpy2DAsync(h.value + (lo * W + x0) * 3, W * 3, d.value + (lo * W + x0) * 3, W * 3, (x1 - x0) * 3, hi - lo, D2H, st))
        ck(hip.hipMemcpy2DAsync(h.value + W * H * 3 + lo * W + x0, W, d.value + W * H * 3 + lo * W + x0, W, x1 - x0, hi - lo, D2H, st))

    t2 = timed(cols)
    nb2 = (hi - lo) * (x1 - x0) * 4
    print(f"columns {x0}..{x1} ({frac:.0%}): {t2 * 1e3:.1f} us for {nb2 / 1e6:.2f} MB = {nb2 / t2 / 1e6:.1f} GB/s ({t2 / t:.2f} of the row copy's time)", flush=True)
