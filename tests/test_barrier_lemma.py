"""The barrier lemma behind the HIP path's march fast-forward (nerf-cuda_amd/csrc/nrf_device.h, "barrier fast-forward"),
checked on the CPU against the reference's own trip loop.

The oracle (`nrfo_march_trip_starts`) runs kernel_march_rays' loop (render_utils.h:593-653) along a ray of an EMPTY volume and
records the t at which every trip begins.  The kernel claims: for the barrier time e it computes ahead of any t_skip, the
first member of the step sequence t_{k+1} = t_k + clamp(t_k * dt_gamma, dt_min, dt_max) that is >= e IS one of those trip
starts -- whatever the hops before it did.  Here the kernel's barrier arithmetic is restated in numpy float32 (same
operations, same margins) and that claim is tested for random rays, step sizes, grid sizes, one to five cascades (an ARBITRARY member of the sequence is a
trip start only 60-88 % of the time: the test would notice a wrong certificate).
CPU only: no GPU, no HIP library."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as op

f = np.float32
NONE = f(-3.402823466e+38)
DT_MIN = f(2 * 1.7320508075688772 / 1024)


def _lib():
    L = op.lib()
    L.nrfo_march_trip_starts.restype = C.c_uint32
    L.nrfo_march_trip_starts.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                         C.c_void_p, C.c_uint32]
    return L


def clamp3(x, lo, hi):
    return f(min(max(f(x), f(lo)), f(hi)))


def ctab(level, n, H, bound):
    """cell_bound[level][n] = ((n / (H-1)) * 2 - 1) * mip_bound in the reference's operation order (render_utils.h:643)"""
    mb = f(min(f(2.0 ** level), f(bound)))
    return f(f(f(f(n) / f(H - 1)) * f(2) - f(1)) * mb)


def barrier_unit(H, o, d, rd, t_skip, mb=1.0, bound=1.0):
    """fast_forward_to_barrier (one cascade, mip_bound mb = min(1, bound)): max over the negative axes of barrier_before"""
    best = NONE
    mb = f(mb)
    for a in range(3):
        if not (d[a] < 0) or not (rd[a] > f(-3.0e38)):
            continue
        xs = f(o[a] + f(t_skip * d[a]))
        v = clamp3(np.ceil(f(f(f(xs / mb) + f(1)) * f(f(0.5) * f(H - 1)))), 0, H)
        n = int(v)
        if n > H - 1:
            continue  # above the plane of the last slab (positions beyond the grid clamp into it): no plane passed yet
        eps = f(f(f(4.0e-6) * f(f(t_skip + f(bound)) + f(2))) * f(f(1) + abs(rd[a])))
        e = f(f(f(ctab(0, n, H, mb) - o[a]) * rd[a]) + eps)
        if not (e <= t_skip):
            n += 1
            if n > H - 1:
                continue
            e = f(f(f(ctab(0, n, H, mb) - o[a]) * rd[a]) + eps)
        if e <= t_skip:
            best = max(best, e)
    return best


def cube_interval(s, o, rd):
    t_in, t_out, ok = NONE, f(3.402823466e+38), True
    with np.errstate(invalid="ignore", over="ignore"):
        for a in range(3):
            u, v = f(f(-s - o[a]) * rd[a]), f(f(s - o[a]) * rd[a])
            ok = ok and not np.isnan(u) and not np.isnan(v)
            t_in = f(np.fmax(t_in, np.fmin(u, v)))
            t_out = f(np.fmin(t_out, np.fmax(u, v)))
    return ok, t_in, t_out


def barrier_pow2(C_, H, bound, o, d, rd, t, t_skip):
    """fast_forward_to_barrier_pow2: the search over shells, the window conditions (W) and (O)"""
    mag = f(f(t_skip + f(bound)) + f(2))
    pad = f(f(1.0e-4) * mag)
    slab0 = f(f(2) / f(H))
    t_hi, best = f(t_skip), NONE
    for _ in range(C_ + 1):
        if best > NONE or not (t_hi > t):
            break
        q = [clamp3(f(o[a] + f(t_hi * d[a])), -bound, bound) for a in range(3)]
        _, ex = np.frexp(f(max(abs(q[0]), abs(q[1]), abs(q[2]))))
        L = min(max(int(ex), 0), C_ - 1)
        mb = f(min(f(2.0 ** L), f(bound)))
        slab_next = f(slab0 * f(min(f(2.0 ** (L + 1)), f(bound))))
        has_in, has_out = L >= 1, L <= C_ - 2
        ok_in, ai, bi = cube_interval(f(f(2.0 ** (L - 1)) + pad), o, rd) if has_in else (True, f(0), f(0))
        ok_out, ao, bo = cube_interval(f(f(2.0 ** L) - pad), o, rd) if has_out else (True, f(0), f(0))
        for a in range(3):
            if not (d[a] < 0) or not (rd[a] > f(-3.0e38)):
                continue
            ard = abs(rd[a])
            eps = f(f(f(4.0e-6) * mag) * f(f(1) + ard))
            xs = f(o[a] + f(t_hi * d[a]))
            n = int(clamp3(np.ceil(f(f(f(xs / mb) + f(1)) * f(f(0.5) * f(H - 1)))), 0, H))
            if n > H - 1:
                continue
            T = f(f(ctab(L, n, H, bound) - o[a]) * rd[a])
            e = f(T + eps)
            if not (e <= t_hi):
                n += 1
                if n > H - 1:
                    continue
                T = f(f(ctab(L, n, H, bound) - o[a]) * rd[a])
                e = f(T + eps)
            if not (e <= t_hi):
                continue
            w_lo = f(f(T - f(slab_next * ard)) - eps)
            ok = ok_in and ok_out
            if has_out:
                ok = ok and ao <= w_lo and e <= bo
            if has_in:
                ok = ok and (ai > bi or e <= ai or w_lo >= bi)
            Lp = L + 2
            while Lp <= C_ - 1 and ok:
                ok, a2, b2 = cube_interval(f(f(2.0 ** (Lp - 1)) - pad), o, rd)
                ok = ok and a2 <= f(f(T - f(f(slab0 * f(min(f(2.0 ** Lp), f(bound)))) * ard)) - eps) and e <= b2
                Lp += 1
            if ok:
                best = max(best, e)
        if best > NONE:
            break
        nxt = NONE
        lim = f(t_hi - f(f(1.0e-6) * mag))
        if has_in and ok_in:
            for v in (ai, bi):
                if v < lim:
                    nxt = max(nxt, v)
        if has_out and ok_out:
            for v in (ao, bo):
                if v < lim:
                    nxt = max(nxt, v)
        t_hi = f(nxt - f(f(4) * pad))
    return best


def first_member_at_or_after(t0, target, far, dt_gamma, dt_max):
    t = f(t0)
    while t < target and t < far:
        t = f(t + clamp3(f(t * f(dt_gamma)), DT_MIN, dt_max))
    return t


def random_ray(rng, bound, r_max):
    r = float(rng.uniform(0.2, r_max)) * bound
    v = rng.normal(size=3)
    pos = v / np.linalg.norm(v) * r
    target = rng.uniform(-0.8, 0.8, size=3) * bound * (0.25 if rng.random() < 0.6 else 1.0)
    dd = target - pos
    dd /= np.linalg.norm(dd)
    o, d = pos.astype(np.float32), dd.astype(np.float32)
    with np.errstate(divide="ignore"):
        rd = (f(1) / d).astype(np.float32)
    # the ray inside the aabb (any start would do: the lemma does not depend on where the sequence begins)
    lo, hi = (-bound - pos) / dd, (bound - pos) / dd
    near = max(float(np.minimum(lo, hi).max()), 0.05)
    far = float(np.maximum(lo, hi).min())
    return o, d, rd, near, far


CASES = [  # cascades, bound, H, dt_gamma, camera distance up to (in bounds), rays
    (1, 1.0, 128, 1.0 / 128.0, 5.0, 700), (1, 1.0, 64, 0.0, 4.0, 300), (1, 1.0, 32, 1.0 / 64.0, 6.0, 500), (1, 1.0, 128, 1.0 / 256.0, 5.0, 400),
    (2, 2.0, 64, 1.0 / 128.0, 4.0, 500), (3, 4.0, 128, 1.0 / 128.0, 4.0, 500), (5, 16.0, 64, 1.0 / 128.0, 3.0, 700), (5, 16.0, 128, 1.0 / 32.0, 3.0, 500),
    (4, 8.0, 32, 1.0 / 64.0, 3.0, 400), (2, 2.0, 64, 0.0, 3.0, 200),
    # positions beyond the grid (one cascade with bound > 1, fewer cascades than the bound needs), grids / bounds that are not powers of two
    (1, 2.0, 64, 1.0 / 128.0, 4.0, 1200), (1, 4.0, 128, 1.0 / 64.0, 3.0, 600), (2, 16.0, 64, 1.0 / 128.0, 3.0, 600), (1, 0.75, 100, 1.0 / 128.0, 5.0, 400),
    (3, 3.0, 96, 1.0 / 128.0, 4.0, 500), (1, 1.0, 100, 0.0, 4.0, 300),
]


@pytest.mark.parametrize("cascade,bound,H,dt_gamma,r_max,n_rays", CASES)
def test_first_member_behind_the_barrier_is_a_trip_start(cascade, bound, H, dt_gamma, r_max, n_rays):
    L = _lib()
    rng = np.random.default_rng(1234 + cascade * 100 + H)
    dt_max = f(f(2) * f(bound) / f(H))
    cap = 1 << 16
    starts = np.empty(cap, np.float32)
    checked = with_barrier = 0
    for _ in range(n_rays):
        o, d, rd, near, far = random_ray(rng, bound, r_max)
        if not (near < far):
            continue
        near, far = f(near), f(far)
        n = L.nrfo_march_trip_starts(C.c_float(bound), cascade, H, C.c_float(dt_gamma), o.ctypes.data, d.ctypes.data, C.c_float(near),
                                     C.c_float(far), starts.ctypes.data, cap)
        assert 0 < n <= cap
        trip_starts = set(starts[:n].tolist())
        for t_skip in rng.uniform(float(near), float(far), size=3).astype(np.float32):
            if cascade == 1:
                e = barrier_unit(H, o, d, rd, f(t_skip), min(1.0, bound), bound)
            else:
                e = barrier_pow2(cascade, H, bound, o, d, rd, near, f(t_skip))
            checked += 1
            if not (e > NONE):
                continue  # no negative axis / no qualifying plane: the kernel keeps its trips
            assert e <= t_skip
            if not (e > near):
                continue  # the barrier lies before the start: nothing is skipped
            with_barrier += 1
            m = first_member_at_or_after(near, min(e, far), far, dt_gamma, dt_max)
            # the fast-forward resumes the simulation at m: the reference must begin a trip exactly there (or the ray has ended)
            assert m >= far or float(m) in trip_starts, (cascade, bound, H, dt_gamma, o.tolist(), d.tolist(), float(near), float(t_skip), float(e), float(m))
    assert with_barrier >= checked // 4  # the certificate exists for most rays (all but those without a negative axis)
