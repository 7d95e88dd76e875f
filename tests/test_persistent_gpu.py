"""The two schedulings of the fused kernel must produce the same frames bit for bit: `render_persistent_kernel` (one
workgroup per CU, waves pull strips from per-XCD work queues; the default for the base.json shape) against
`render_kernel` (one workgroup per strip; `NRF_PERSISTENT=0`).  Every render goes into poisoned caller-owned planes,
so a pixel that neither the queues nor the background sweep of the persistent kernel covers shows up.  The cases are
the ones its queue arithmetic has to get right: frames whose region of interest is the whole image, a strip of it,
empty; images smaller than a workgroup's share; odd sizes; shards of 2 / 3 / 8 ranks; more views than one launch
takes; several cascades; queue settings (one queue, row order).  Sizes are small: the point is coverage, the speed
is bench.py's business."""
import os

import numpy as np
import pytest

import models
import nerfhip as nh
import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _poses(kind, n):
    if kind == "orbit":
        return [syn.orbit_pose(37.0 * i + 5.0, (25.0, -10.0, 60.0)[i % 3]) for i in range(n)]
    if kind == "inside":      # camera inside the volume: the region of interest is the whole image
        return [syn.orbit_pose(50.0 * i, 20.0, radius=0.4 / 0.33) for i in range(n)]
    if kind == "away":        # looking away from the object: empty region of interest -> all background
        out = []
        for i in range(n):
            m = syn.orbit_pose(40.0 * i, 15.0).copy()
            m[:3, 0] *= -1.0  # turn the camera round (x and z axes flipped: still right-handed)
            m[:3, 2] *= -1.0
            out.append(m)
        return out
    if kind == "far":         # the object is a few pixels wide: a region of interest of one or two strips
        return [syn.orbit_pose(70.0 * i, 35.0, radius=30.0 / 0.33) for i in range(n)]
    raise ValueError(kind)


def _render(desc, W, H, poses, env, shard=(0, 1), opts_kw=None):
    """Frames of `poses` (one nrf_render_views call) with the environment `env` in force at context creation."""
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = nh.NerfHip(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.load_model(desc)
    o = nh.default_options()
    o.shard_index, o.shard_count = shard
    for k, v in (opts_kw or {}).items():
        setattr(o, k, v)
    ctx.set_options(o)
    ctx.set_resolution(W, H)
    n = len(poses)
    n_px = nh.tiles_per_shard(W, H, shard[1]) * 64 if shard[1] > 1 else W * H
    rgba = torch.full((n, n_px, 4), 7.0, device="cuda")
    depth = torch.full((n, n_px), 7.0, device="cuda")
    torch.cuda.synchronize()
    ctx.bind_output(rgba.data_ptr(), depth.data_ptr())
    ctx.render_views(np.stack([syn.default_camera(W, H)] * n), np.stack(poses))
    st = ctx.stats()
    assert st.n_samples >= st.n_composited > 0 or st.n_samples == st.n_composited == 0
    # (n_samples -- evaluated samples -- depends on how rays are batched into rounds: tail splitting hands rays to idle waves;
    #  the samples that reach a ray's compositing sum do not)
    out = rgba.cpu().numpy(), depth.cpu().numpy(), int(st.n_composited), int(st.n_rays)
    ctx.close()
    return out


def _same(a, b, what):
    assert a[2] == b[2] and a[3] == b[3], (what, a[2:], b[2:])          # composited samples, rays
    # every pixel the per-strip kernel writes is written (what stays poisoned in both: the padding tiles at the end of a
    # shard's tile-major buffer when the shards are uneven -- nobody's pixels); unsharded frames have no such padding
    assert np.array_equal(a[0] == 7.0, b[0] == 7.0) and np.array_equal(a[1] == 7.0, b[1] == 7.0), what
    if a[0].shape[1] == b[0].shape[1] and what is not None and not np.any(b[0] == 7.0):
        assert not np.any(a[0] == 7.0) and not np.any(a[1] == 7.0), what
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), what
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), what


STRIP = {"NRF_PERSISTENT": "0"}
PERSISTENT = {"NRF_PERSISTENT": "1"}


@pytest.fixture(scope="module")
def model():
    desc, keep, _ = models.build_model(log2_hashmap_size=15, H=64)
    return desc, keep


@pytest.mark.parametrize("W,H", [(333, 211), (64, 40), (8, 8), (1000, 24), (40, 600)])
@pytest.mark.parametrize("kind", ["orbit", "inside", "away", "far"])
def test_persistent_kernel_equals_strip_kernel(model, W, H, kind):
    desc, _ = model
    poses = _poses(kind, 5)
    ref = _render(desc, W, H, poses, STRIP)
    _same(_render(desc, W, H, poses, PERSISTENT), ref, (W, H, kind))


@pytest.mark.parametrize("count", [2, 3, 8])
def test_persistent_kernel_shards(model, count):
    desc, _ = model
    W, H = 501, 283  # 63 x 36 tiles: 16 strip columns, the last one partly outside the image
    poses = _poses("orbit", 3) + _poses("inside", 2) + _poses("far", 1)
    for index in range(count):
        ref = _render(desc, W, H, poses, STRIP, shard=(index, count))
        _same(_render(desc, W, H, poses, PERSISTENT, shard=(index, count)), ref, (count, index))


def test_more_views_than_one_launch_takes(model):
    desc, _ = model
    n = nh.NRF_MAX_VIEWS + 5
    poses = _poses("orbit", n)
    ref = _render(desc, 96, 64, poses, STRIP)
    _same(_render(desc, 96, 64, poses, PERSISTENT), ref, "two launches")
    _same(_render(desc, 96, 64, poses, PERSISTENT, shard=(1, 2)), _render(desc, 96, 64, poses, STRIP, shard=(1, 2)), "two launches, sharded")


def test_queue_settings_do_not_change_the_picture(model):
    desc, _ = model
    poses = _poses("orbit", 4) + _poses("inside", 2)
    ref = _render(desc, 400, 300, poses, STRIP)
    for env in ({"NRF_QUEUE_CLASSES": "1"}, {"NRF_QUEUE_CLASSES": "3"}, {"NRF_CENTRE_OUT": "0"},
                {"NRF_QUEUE_CLASSES": "8", "NRF_CENTRE_OUT": "0"}):
        _same(_render(desc, 400, 300, poses, dict(PERSISTENT, **env)), ref, env)


@pytest.mark.parametrize("W,H,kind,n,shard", [(1920, 1080, "orbit", 1, (0, 1)), (800, 800, "inside", 1, (0, 1)), (640, 360, "orbit", 6, (0, 1)),
                                                (1280, 720, "orbit", 2, (1, 3)), (1000, 24, "far", 3, (0, 1))])
def test_planned_queue_order_does_not_change_the_picture(model, W, H, kind, n, shard):
    """plan_price_kernel / plan_sort_kernel permute the order in which the queues hand the strips out (dearest estimated strip
    first); with NRF_PLAN_MAX_POS=0 the positions come in order.  A strip the permutation lost would leave its pixels
    poisoned, one it held twice would count its rays twice."""
    desc, _ = model
    poses = _poses(kind, n)
    ref = _render(desc, W, H, poses, dict(PERSISTENT, NRF_PLAN_MAX_POS="0"), shard=shard)
    _same(_render(desc, W, H, poses, PERSISTENT, shard=shard), ref, (W, H, kind, "planned"))
    _same(_render(desc, W, H, poses, dict(PERSISTENT, NRF_QUEUE_CLASSES="3"), shard=shard), ref, (W, H, kind, "planned, 3 classes"))


def _octant_poses(radius=4.0311):
    """Cameras above, below and level with the object from all sides: every sign pattern of the ray direction."""
    return [syn.orbit_pose(az, el, radius=radius) for el in (55.0, 12.0, -35.0, -70.0) for az in (10.0, 100.0, 190.0, 280.0)]


@pytest.mark.parametrize("sched", [PERSISTENT, STRIP])
@pytest.mark.parametrize("W,H,dt_gamma,hash_log2,grid_H", [(333, 211, 1.0 / 128.0, 15, 64), (256, 192, 0.0, 15, 64), (200, 160, 1.0 / 64.0, 15, 64),
                                                            (320, 200, 1.0 / 128.0, 14, 128), (160, 120, 1.0 / 256.0, 14, 32)])
def test_barrier_fast_forward_does_not_change_the_picture(sched, W, H, dt_gamma, hash_log2, grid_H):
    """fast_forward_to_barrier steps a ray from its start to the last barrier plane ahead of t_skip without simulating the
    trips in between (nrf_device.h).  With NRF_MARCH_FF=0 every trip is simulated: frames, composited samples and ray counts
    must be identical -- for every sign pattern of the direction (negative axes carry the barriers, rays without one keep
    their trips), step sizes at dt_min / growing / at dt_max, cameras near and far, three grid resolutions."""
    desc, _ = models.build_model(log2_hashmap_size=hash_log2, H=grid_H)[:2]
    poses = _octant_poses() + _octant_poses(radius=1.9)[::3] + _poses("inside", 2) + _poses("far", 1)
    kw = {"dt_gamma": dt_gamma}
    ref = _render(desc, W, H, poses, dict(sched, NRF_MARCH_FF="0"), opts_kw=kw)
    _same(_render(desc, W, H, poses, sched, opts_kw=kw), ref, (W, H, dt_gamma, grid_H, "fast-forward"))


@pytest.mark.parametrize("sched", [PERSISTENT, STRIP])
@pytest.mark.parametrize("cascade,bound,grid_H,dt_gamma", [(5, 16.0, 64, 1.0 / 128.0), (3, 4.0, 128, 1.0 / 128.0), (2, 2.0, 32, 0.0), (5, 16.0, 32, 1.0 / 32.0)])
def test_barrier_fast_forward_with_cascades(sched, cascade, bound, grid_H, dt_gamma):
    """Several cascades (fast_forward_to_barrier_pow2): a level-L plane is a barrier only while no trip of another level can
    reach past it.  Cameras outside the aabb, in the outer shells, close to shell boundaries and inside the innermost cube,
    looking in every octant; identical frames with and without the fast-forward."""
    desc, _ = models.build_model(log2_hashmap_size=14, H=grid_H, cascade=cascade, bound=bound)[:2]
    poses = []
    for radius in (1.2, 2.05, 3.9, 8.3, 17.0, 40.0):
        poses += [syn.orbit_pose(az, el, radius=radius / 0.33) for az, el in ((15.0, 40.0), (140.0, -25.0), (250.0, 8.0), (320.0, -60.0))]
    poses += _poses("inside", 2)
    kw = {"dt_gamma": dt_gamma, "max_steps": 1024}
    ref = _render(desc, 240, 160, poses, dict(sched, NRF_MARCH_FF="0"), opts_kw=kw)
    _same(_render(desc, 240, 160, poses, sched, opts_kw=kw), ref, (cascade, bound, grid_H, dt_gamma, "fast-forward, cascades"))


@pytest.mark.parametrize("sched", [PERSISTENT, STRIP])
@pytest.mark.parametrize("kw", [
    dict(H=96),                                                       # a grid that is no power of two: the generic march
    dict(H=64, cascade=1, bound=2.0),                                 # one cascade, bound 2: positions beyond the grid clamp into its last slabs
    dict(H=48, cascade=3, bound=3.0),                                 # a bound that is no power of two, several cascades
    dict(H=64, n_neurons=32, n_features_per_level=4, n_levels=8),     # the generic network instance (always the generic march)
    dict(H=64, cascade=3, bound=4.0, dir_otype="Frequency", n_frequencies=12),  # wide instance, cascades
], ids=["H96", "bound2-1cascade", "bound3-3cascades", "generic-net", "wide-cascades"])
def test_barrier_fast_forward_other_march_instances(sched, kw):
    """The fast-forward in the generic march (any grid size / bound, the generic network instance) and next to the other
    network instances; identical frames with NRF_MARCH_FF=0."""
    desc, _ = models.build_model(log2_hashmap_size=13, **kw)[:2]
    poses = _octant_poses()[::2] + _octant_poses(radius=2.4 / 0.33)[1::3] + _poses("inside", 2)
    opts = {"dt_gamma": 1.0 / 128.0, "max_steps": 1024}
    ref = _render(desc, 200, 144, poses, dict(sched, NRF_MARCH_FF="0"), opts_kw=opts)
    _same(_render(desc, 200, 144, poses, sched, opts_kw=opts), ref, (kw, "fast-forward, other instances"))


@pytest.mark.parametrize("sched", [PERSISTENT, STRIP])
@pytest.mark.parametrize("kw", [
    dict(H=64, density_hidden_layers=2, rgb_hidden_layers=3),            # other depths: generic networks behind a standard grid
    dict(H=64, n_neurons=32, n_levels=13),                               # fewer levels than 16 (lanes without a fourth level), 32 neurons
    dict(H=64, log2_hashmap_size=12, sh_degree=6, n_levels=5),           # dense levels only + a few hashed ones
    dict(H=32, cascade=2, bound=2.0, activation="Sigmoid"),              # another activation, two cascades
    dict(H=64, interpolation="Smoothstep"),                               # Smoothstep: the fractions are transformed, the gathers are the same
], ids=["depths", "13levels-32n", "5levels-sh6", "sigmoid-2cascades", "smoothstep"])
def test_generic_instance_fast_grid_equals_literal_grid(sched, kw):
    """gen_encode_rows takes the register-resident instance's gathers (level_gather / level_interp) when the grid is of its
    kind (F = 2, Linear, dense / power-of-two levels); NRF_GEN_FAST_GRID=0 keeps gen_level's literal arithmetic.  Same
    features, hence the same frames, bit for bit."""
    kw = dict(kw)
    log2T = kw.pop("log2_hashmap_size", 14)
    env = dict(sched, NRF_WIDTH_INSTANCES="0")  # (16 / 32 / 128-neuron and SH models: the generic instance, not their own ones)
    desc, _ = models.build_model(log2_hashmap_size=log2T, **kw)[:2]
    poses = _poses("orbit", 3) + _poses("inside", 1)
    ref = _render(desc, 240, 160, poses, dict(env, NRF_GEN_FAST_GRID="0"))
    _same(_render(desc, 240, 160, poses, env), ref, (kw, "fast grid"))


def test_persistent_kernel_with_cascades_and_sample_cap():
    """BASELINE config 4 shape (bound 16, five cascades: per-cascade visibility walks on the workgroup's own copy of the
    dilated table, 44 KB of march tables in LDS) and a small max_steps."""
    desc, keep, _ = models.build_model(log2_hashmap_size=15, H=64, cascade=5, bound=16.0)
    poses = [syn.orbit_pose(30.0, 20.0), syn.orbit_pose(200.0, -15.0, radius=1.5 / 0.33), syn.orbit_pose(120.0, 70.0, radius=9.0 / 0.33)]
    for kw in ({"max_steps": 1024}, {"max_steps": 7}):
        ref = _render(desc, 320, 200, poses, STRIP, opts_kw=kw)
        _same(_render(desc, 320, 200, poses, PERSISTENT, opts_kw=kw), ref, kw)


@pytest.mark.parametrize("kw", [
    dict(dir_otype="Frequency", n_frequencies=12),                                        # the wide form of the register-resident instance
    dict(n_neurons=32, n_features_per_level=4, n_levels=8, interpolation="Smoothstep"),   # generic instance
    dict(dir_otype="SphericalHarmonics", sh_degree=6, density_hidden_layers=2, n_neurons=128),  # generic, wide rows
], ids=["wide-frequency12", "generic-32x4x8", "generic-sh6-128"])
def test_persistent_kernel_other_instances(kw):
    """The wide and generic instances run in the persistent form as well (12 and 8 waves per workgroup)."""
    desc, keep, _ = models.build_model(log2_hashmap_size=14, H=64, **kw)
    poses = _poses("orbit", 3) + _poses("inside", 1) + _poses("away", 1)
    for shard in ((0, 1), (2, 3)):
        ref = _render(desc, 300, 180, poses, STRIP, shard=shard)
        _same(_render(desc, 300, 180, poses, PERSISTENT, shard=shard), ref, (kw, shard))


@pytest.mark.parametrize("env", [STRIP, PERSISTENT], ids=["per-strip", "persistent"])
@pytest.mark.parametrize("shard", [(0, 1), (1, 3)])
def test_packed_8bit_output_equals_quantised_float_planes(model, env, shard):
    """nrf_bind_output_rgbd8: the kernel writes the reference's 8-bit pixels itself; bit-identical to nrf_quantize_rgbd8 of
    the float planes, for whole frames and for a shard's tile-major buffer (padding pixels zero), in both schedulings."""
    desc, _ = model
    W, H = 333, 211
    poses = _poses("orbit", 3) + _poses("inside", 1) + _poses("away", 1)
    n = len(poses)
    cams = np.stack([syn.default_camera(W, H)] * n)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = nh.NerfHip(0)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    ctx.load_model(desc)
    o = nh.default_options()
    o.shard_index, o.shard_count = shard
    ctx.set_options(o)
    ctx.set_resolution(W, H)
    n_px = nh.tiles_per_shard(W, H, shard[1]) * 64 if shard[1] > 1 else W * H
    rgba = torch.zeros((n, n_px, 4), device="cuda")
    depth = torch.zeros((n, n_px), device="cuda")
    want = torch.zeros((n, n_px), dtype=torch.int32, device="cuda")
    got = torch.full((n, n_px), 0x07070707, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.bind_output(rgba.data_ptr(), depth.data_ptr())
    ctx.render_views(cams, np.stack(poses))
    ctx.quantize_rgbd8(rgba.data_ptr(), depth.data_ptr(), n * n_px, want.data_ptr())
    ctx.bind_output_rgbd8(got.data_ptr())
    f = ctx.render_views(cams, np.stack(poses))
    torch.cuda.synchronize()
    assert not f.rgba and not f.depth  # the packed frame is the caller's
    with pytest.raises(nh.NerfHipError):
        ctx.read_f32()
    g, w = got.cpu().numpy(), want.cpu().numpy()
    if shard[1] > 1:  # padding tiles at the end of an uneven shard's buffer are nobody's pixels
        written = g != 0x07070707
        assert written.mean() > 0.97 and np.array_equal(g[written], w[written])
    else:
        assert np.array_equal(g, w)
    ctx.bind_output_rgbd8(0)      # back to the context's own float planes
    ctx.set_max_views(n)
    ctx.render_views(cams, np.stack(poses))
    ctx.close()


@pytest.mark.parametrize("env", [STRIP, PERSISTENT], ids=["per-strip", "persistent"])
def test_shard_without_a_strip_renders_nothing(model, env):
    """11 x 42 pixels are 6 strips: rank 7 of 8 owns none of them -- the launch is a no-op, not an error."""
    desc, _ = model
    out = _render(desc, 11, 42, _poses("orbit", 2), env, shard=(7, 8))
    assert out[2] == 0 and np.all(out[0] == 7.0)  # no sample, no pixel touched


@pytest.mark.parametrize("W,H,n", [(333, 211, 5), (640, 360, 3)])
def test_transmittance_sample_cap_never_changes_a_pixel(W, H, n):
    """The per-round sample queue by transmittance (FrameParams::sample_cap; the default 2 for launches of three views and
    more, 0 = the full queue of eight for one or two): a ray queues fewer samples per round the less it can still absorb.
    Per-ray semantics must not depend on it -- launches of 3+ views with NRF_SAMPLE_CAP = 0, 1, 2 (read at nrf_create:
    separate contexts) give bit-identical float planes and equal composited-sample counts, at float-bit level (the golden
    hashes and single-view references only ever exercise cap 0)."""
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
    poses = _poses("orbit", n)
    ref = _render(desc, W, H, poses, {"NRF_SAMPLE_CAP": "0"})
    for cap in ("1", "2"):
        _same(_render(desc, W, H, poses, {"NRF_SAMPLE_CAP": cap}), ref, (W, H, "cap " + cap))
    # and the default context (cap 2 for this launch) is the same again
    _same(_render(desc, W, H, poses, {}), ref, (W, H, "default"))
