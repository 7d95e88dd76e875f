"""The host end of render_frame: the reference's API returns HOST memory -- 8-bit rgb + depth
(R/src/nerf_render.cu:345-359).  Here the kernel writes those bytes itself (OUT_U8: whole dwords assembled inside the
tile's wavefront), the rows of the view's region of interest travel by asynchronous copies into pinned host memory and
the calling thread fills the other rows with the background.  Every byte must equal the after-the-fact path
(nrf_render into float planes + nrf_read_u8), which tests/test_parity_gpu.py ties to the oracle's quantisation -- and
the oracle is compared directly as well.  Cases: frame widths that are / are not multiples of 4 and of 8 (dword path,
byte path, ragged tiles), regions of interest from empty to the whole image, cameras that change between calls on the
same slot (the background bookkeeping of the pinned planes), a changing background colour, rgb-only frames, two calls
in flight, batches larger than the slot was sized for, device groups, caller-bound 8-bit planes in both scheduling
forms of the kernel, and two launches of ONE context overlapping on different streams (the per-call ring of work
queues)."""
import os

import numpy as np
import pytest

import models
import nerfhip as nh
import synthetic as syn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def model():
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    return desc, keep, cfg


def _away(pose):
    m = pose.copy()
    m[:3, 0] *= -1.0
    m[:3, 2] *= -1.0
    return m


def _reference_u8(desc, W, H, cams, poses, opts=None):
    """nrf_render into float planes + nrf_read_u8, one view at a time."""
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    if opts is not None:
        ctx.set_options(opts)
    ctx.set_resolution(W, H)
    out = []
    for c, p in zip(cams, poses):
        ctx.render(c, p)
        out.append(ctx.read_u8())
    ctx.close()
    return out


@pytest.mark.parametrize("W,H", [(64, 48), (100, 60), (101, 77), (36, 20), (8, 8), (1920 // 4, 1080 // 4)])
def test_host_frames_equal_read_u8(model, W, H):
    desc, keep, cfg = model
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(30, 30), syn.orbit_pose(200, -15), _away(syn.orbit_pose(40, 15)),
             syn.orbit_pose(50, 20, radius=0.4 / 0.33), syn.orbit_pose(70, 35, radius=30.0 / 0.33)]
    want = _reference_u8(desc, W, H, [cam] * len(poses), poses)
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    # one view per call: the slots alternate, every slot sees a different camera (and region of interest) each time
    for rep in range(2):
        for i, p in enumerate(poses):
            rgb, depth = ctx.render_host_u8([cam], [p])
            np.testing.assert_array_equal(rgb[0], want[i][0], err_msg=f"rgb, pose {i}, pass {rep}")
            np.testing.assert_array_equal(depth[0], want[i][1], err_msg=f"depth, pose {i}, pass {rep}")
    # all views in one call (the slot grows), then fewer views again
    rgb, depth = ctx.render_host_u8([cam] * len(poses), poses)
    for i in range(len(poses)):
        np.testing.assert_array_equal(rgb[i], want[i][0])
        np.testing.assert_array_equal(depth[i], want[i][1])
    rgb, depth = ctx.render_host_u8([cam] * 2, poses[3:1:-1])
    np.testing.assert_array_equal(rgb[0], want[3][0])
    np.testing.assert_array_equal(rgb[1], want[2][0])
    np.testing.assert_array_equal(depth[0], want[3][1])
    assert want[0][0].min() < 250 and (want[2][0] == 255).all()  # the object is in view / the away view is all background
    st = ctx.stats()
    assert st.n_samples >= 0
    ctx.close()


def test_host_frames_against_the_oracle(model):
    import oracle_py as op

    desc, keep, cfg = model
    W, H = 96, 64
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
    rgba, depth, _ = op.Oracle(desc).render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    want_rgb, want_depth = op.quantize_u8(rgba, depth)
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    rgb, d8 = ctx.render_host_u8([cam], [pose])
    ctx.close()
    for got, want, what in ((rgb[0], want_rgb, "rgb"), (d8[0], want_depth, "depth")):
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (what, int(d.max()), float((d > 0).mean()))


def test_pipelined_submits_rgb_only_and_background_changes(model):
    desc, keep, cfg = model
    W, H = 128, 72
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(25.0 * i, 10.0 + 5 * (i % 4)) for i in range(12)]
    want = _reference_u8(desc, W, H, [cam] * len(poses), poses)
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    # two calls in flight: submit k + 1 before waiting for k (views of the pinned planes, no copies)
    tickets = [ctx.submit_host_u8([cam] * 3, poses[0:3])]
    for k in range(1, 4):
        tickets.append(ctx.submit_host_u8([cam] * 3, poses[3 * k:3 * k + 3], flags=nh.NRF_HOST_RGB_ONLY if k % 2 else 0))
        rgb, depth = ctx.wait_host_u8(tickets[k - 1], copy=False)
        for v in range(3):
            np.testing.assert_array_equal(rgb[v], want[3 * (k - 1) + v][0])
        assert (depth is None) == ((k - 1) % 2 == 1)
        if depth is not None:
            for v in range(3):
                np.testing.assert_array_equal(depth[v], want[3 * (k - 1) + v][1])
    rgb, depth = ctx.wait_host_u8(tickets[3])
    np.testing.assert_array_equal(rgb[2], want[11][0])
    # another background colour: every row of the pinned planes has to follow
    o = nh.default_options()
    o.bg_color = 0.25
    ctx.set_options(o)
    want_bg = _reference_u8(desc, W, H, [cam] * 2, [poses[0], _away(poses[1])], opts=o)
    for _ in range(2):
        rgb, depth = ctx.render_host_u8([cam] * 2, [poses[0], _away(poses[1])])
        np.testing.assert_array_equal(rgb[0], want_bg[0][0])
        np.testing.assert_array_equal(rgb[1], want_bg[1][0])
        np.testing.assert_array_equal(depth[1], want_bg[1][1])
    assert (want_bg[1][0] == 63).all()  # (unsigned char)(255.0 * 0.25)
    ctx.close()


@pytest.mark.parametrize("persistent", ["1", "0"])
def test_bound_u8_planes_in_both_kernel_forms(model, persistent):
    """nrf_bind_output_u8: caller-owned planes (here torch tensors), poisoned first; persistent and per-strip kernel."""
    desc, keep, cfg = model
    saved = os.environ.get("NRF_PERSISTENT")
    os.environ["NRF_PERSISTENT"] = persistent
    try:
        ctx = nh.NerfHip(0)
    finally:
        if saved is None:
            os.environ.pop("NRF_PERSISTENT", None)
        else:
            os.environ["NRF_PERSISTENT"] = saved
    ctx.load_model(desc)
    import ctypes as C
    ctx.lib.nrf_debug_instance.argtypes = [C.c_void_p]
    assert (ctx.lib.nrf_debug_instance(ctx.h) >= 16) == (persistent == "1")
    for W, H in ((96, 64), (75, 41)):
        cam = syn.default_camera(W, H)
        poses = [syn.orbit_pose(30, 30), syn.orbit_pose(140, 5), _away(syn.orbit_pose(10, 10))]
        want = _reference_u8(desc, W, H, [cam] * 3, poses)
        ctx.set_resolution(W, H)
        rgb = torch.full((3, H, W, 3), 77, dtype=torch.uint8, device="cuda")
        depth = torch.full((3, H, W), 78, dtype=torch.uint8, device="cuda")
        ctx.bind_output_u8(rgb.data_ptr(), depth.data_ptr())
        ctx.render_views([cam] * 3, poses)
        with pytest.raises(nh.NerfHipError):
            ctx.read_u8()  # the frame is the caller's to read
        for v in range(3):
            np.testing.assert_array_equal(rgb[v].cpu().numpy(), want[v][0])
            np.testing.assert_array_equal(depth[v].cpu().numpy(), want[v][1])
        ctx.bind_output_u8(0, 0)
        ctx.render(cam, poses[0])
        np.testing.assert_array_equal(ctx.read_u8()[0], want[0][0])
    ctx.close()


def test_group_host_frames(model):
    """Members render packed 8-bit shards, the first device untiles into the Image layout (nrf_untile_views_u8)."""
    desc, keep, cfg = model
    for W, H in ((96, 64), (101, 77)):
        cam = syn.default_camera(W, H)
        poses = [syn.orbit_pose(33.0 * i, 20.0) for i in range(5)]
        want = _reference_u8(desc, W, H, [cam] * 5, poses)
        for devices in ([0], [0, 0], [0, 0, 0]):
            g = nh.NerfGroup(devices)
            g.load_model(desc)
            g.set_resolution(W, H)
            t0 = g.submit_host_u8([cam] * 2, poses[0:2])
            t1 = g.submit_host_u8([cam] * 3, poses[2:5])
            rgb, depth = g.wait_host_u8(t0)
            for v in range(2):
                np.testing.assert_array_equal(rgb[v], want[v][0])
                np.testing.assert_array_equal(depth[v], want[v][1])
            rgb, depth = g.wait_host_u8(t1)
            for v in range(3):
                np.testing.assert_array_equal(rgb[v], want[2 + v][0])
                np.testing.assert_array_equal(depth[v], want[2 + v][1])
            # the float path of the group and its u8 readback (no host loop any more) still agree
            g.render_views([cam], [poses[4]])
            r8, d8 = g.read_view_u8(0)
            np.testing.assert_array_equal(r8, want[4][0])
            np.testing.assert_array_equal(d8, want[4][1])
            g.close()


def test_two_streams_on_one_context(model):
    """two launches of ONE context that overlap on different streams must not share a work queue (each
    would render only part of its strips and leave stale pixels).  Every render call takes its own slot of the
    context's ring of counters + queues."""
    desc, keep, cfg = model
    W, H = 256, 192
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(20.0 * i, 25.0) for i in range(8)]
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    want = []
    for p in poses:
        ctx.render(cam, p)
        want.append(ctx.read_f32())
    streams = [torch.cuda.Stream() for _ in range(4)]
    planes = [(torch.full((H, W, 4), float("nan"), device="cuda"), torch.full((H, W), float("nan"), device="cuda")) for _ in poses]
    for rep in range(3):
        for i, p in enumerate(poses):  # eight launches back to back on four streams, nothing waits for anything
            rgba, depth = planes[i]
            rgba.fill_(float("nan"))
            depth.fill_(float("nan"))
        torch.cuda.synchronize()
        for i, p in enumerate(poses):
            ctx.bind_output(planes[i][0].data_ptr(), planes[i][1].data_ptr())
            ctx.render(cam, p, stream=streams[i % 4].cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(poses)):
            np.testing.assert_array_equal(planes[i][0].cpu().numpy(), want[i][0], err_msg=f"launch {i}, pass {rep}")
            np.testing.assert_array_equal(planes[i][1].cpu().numpy(), want[i][1])
    ctx.bind_output(0, 0)
    ctx.close()


def test_quantize_and_untile_u8_entry_points(model):
    """nrf_quantize_u8 / nrf_untile_views_u8 against numpy on random data (NaN / inf / out-of-range included)."""
    desc, keep, cfg = model
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    rng = np.random.default_rng(5)
    n = 10007
    rgba = rng.uniform(-0.2, 1.3, (n, 4)).astype(np.float32)
    depth = rng.uniform(-0.2, 1.3, n).astype(np.float32)
    rgba[5, 0], rgba[6, 1], rgba[7, 2], depth[8] = np.nan, np.inf, -np.inf, np.nan

    def q(v):
        s = 255.0 * v.astype(np.float64)
        out = np.where(s > 0.0, np.minimum(s, 255.0), 0.0)
        return np.nan_to_num(out, nan=0.0).astype(np.uint8)

    t_rgba, t_depth = torch.from_numpy(rgba).cuda(), torch.from_numpy(depth).cuda()
    rgb8 = torch.zeros((n, 3), dtype=torch.uint8, device="cuda")
    d8 = torch.zeros((n,), dtype=torch.uint8, device="cuda")
    ctx.quantize_u8(t_rgba.data_ptr(), t_depth.data_ptr(), n, rgb8.data_ptr(), d8.data_ptr())
    np.testing.assert_array_equal(rgb8.cpu().numpy(), q(rgba[:, :3]))
    np.testing.assert_array_equal(d8.cpu().numpy(), q(depth))
    for W, H, world, views in ((96, 64, 2, 3), (101, 77, 3, 2), (44, 20, 8, 1)):
        ctx.set_resolution(W, H)
        tps = nh.tiles_per_shard(W, H, world)
        packed = rng.integers(0, 2 ** 32, (world, views, tps * 64), dtype=np.uint32)
        g = torch.from_numpy(packed.view(np.int32)).cuda()
        o_rgb = torch.zeros((views, H, W, 3), dtype=torch.uint8, device="cuda")
        o_d = torch.zeros((views, H, W), dtype=torch.uint8, device="cuda")
        ctx.untile_views_u8(g.data_ptr(), world, tps, views, o_rgb.data_ptr(), o_d.data_ptr())
        for v in range(views):
            img = nh.untile_numpy(packed[:, v, :, None], W, H)[..., 0]
            np.testing.assert_array_equal(o_rgb[v].cpu().numpy(), np.stack([img & 255, (img >> 8) & 255, (img >> 16) & 255], -1).astype(np.uint8))
            np.testing.assert_array_equal(o_d[v].cpu().numpy(), (img >> 24).astype(np.uint8))
    ctx.close()


def test_host_frames_1080p_size_independent_properties():
    """BASELINE config 2 size: the 16-view batch of bench.py through the host-frame path equals read_u8 of the float
    planes byte for byte, repeated calls are identical, and pixels outside the region of interest are the background."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H = 1920, 1080
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(45.0 * i, 30.0) for i in range(8)]
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    ctx.set_max_views(8)
    first = ctx.render_host_u8([cam] * 8, poses)
    again = ctx.render_host_u8([cam] * 8, poses)   # the other slot
    third = ctx.render_host_u8([cam] * 8, poses[::-1])  # the first slot again, other cameras
    ctx.render_views([cam] * 8, poses)
    for v in (0, 3, 7):
        r8, d8 = ctx.read_view_u8(v)
        np.testing.assert_array_equal(first[0][v], r8)
        np.testing.assert_array_equal(first[1][v], d8)
        np.testing.assert_array_equal(again[0][v], r8)
        np.testing.assert_array_equal(third[0][7 - v], r8)
        np.testing.assert_array_equal(third[1][7 - v], d8)
    assert (first[0][0][:8] == 255).all() and first[0][0].min() < 200
    ctx.close()


@pytest.mark.parametrize("W,H,n_views", [(96, 56, 1), (96, 56, 16), (480, 272, 3)])
def test_resolution_change_on_a_live_context_keeps_no_state_of_the_old_geometry(model, W, H, n_views):
    """the host-frame slots were keyed on W * H, so a portrait / landscape flip (same pixel count) kept the old
    width's row bookkeeping -- stale pixels came back as background -- and, when H grew, strip-row arrays sized for the
    old tiles_y.  WxH, then HxW, then WxH again on ONE context, each against nrf_render + nrf_read_u8 of a fresh context."""
    desc, keep, cfg = model
    poses = [syn.orbit_pose(25.0 * i, 20 + (i % 3) * 10, radius=(4.0311 if i % 4 else 9.0)) for i in range(n_views)]
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_max_views(n_views)
    for w, h in ((W, H), (H, W), (W, H), (H, W)):
        cam = syn.default_camera(w, h)
        want = _reference_u8(desc, w, h, [cam] * n_views, poses)
        ctx.set_resolution(w, h)
        for rep in range(2):  # both slots
            rgb, depth = ctx.render_host_u8([cam] * n_views, poses)
            for i in range(n_views):
                np.testing.assert_array_equal(rgb[i], want[i][0], err_msg=f"rgb {w}x{h} view {i} pass {rep}")
                np.testing.assert_array_equal(depth[i], want[i][1], err_msg=f"depth {w}x{h} view {i} pass {rep}")
    ctx.close()


def test_region_of_interest_narrower_than_the_frame_travels_by_columns(model):
    """A frame that is copied after its render (one view per call) sends only the COLUMNS of its region of interest (pitched
    copies); the pinned planes hold the background beside them.  The object moves across the frame from call to call on the
    same two slots (principal point shifted left / right / up, far and near cameras, an all-background view in between, a
    changed background colour, rgb-only frames): every byte must equal nrf_render + nrf_read_u8 of a fresh context, and a
    narrow region must have moved fewer bytes than the frame has."""
    desc, keep, cfg = model
    W, H = 640, 360
    base = syn.default_camera(W, H)
    far, near = syn.orbit_pose(40, 20, radius=14.0 / 0.33), syn.orbit_pose(40, 20, radius=3.0 / 0.33)
    seq = []
    for dx, dy, pose in ((-200, 0, far), (200, 0, far), (0, 0, near), (0, -100, far), (250, 120, far), (0, 0, _away(far)), (-250, -120, far),
                         (0, 0, syn.orbit_pose(10, 40)), (120, 60, far)):
        cam = base.copy(); cam[2] += dx; cam[3] += dy
        seq.append((cam, pose))
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    narrow = 0
    for bg, flags in ((1.0, 0), (0.25, 0), (0.25, nh.NRF_HOST_RGB_ONLY)):
        o = nh.default_options(); o.bg_color = bg
        ctx.set_options(o)
        want = _reference_u8(desc, W, H, [c for c, _ in seq], [p for _, p in seq], o)
        for i, (cam, pose) in enumerate(seq):
            f = ctx.render_host_u8_raw(np.ascontiguousarray(cam, np.float32).reshape(1, 4), np.ascontiguousarray(pose, np.float32).reshape(1, 16), flags)
            rgb, depth = nh._host_frame_arrays(f, False)
            np.testing.assert_array_equal(rgb[0], want[i][0], err_msg=f"rgb, bg {bg}, flags {flags}, call {i}")
            if not flags:
                np.testing.assert_array_equal(depth[0], want[i][1], err_msg=f"depth, bg {bg}, call {i}")
            obj = np.any(want[i][0] != want[i][0][0, 0], axis=2)
            if obj.any() and obj.any(axis=0).sum() < W // 3:  # the object covers less than a third of the width
                assert f.copied_bytes < W * H * (3 if flags else 4) * 0.6, (i, f.copied_bytes)
                narrow += 1
    assert narrow >= 6
    ctx.close()


@pytest.mark.parametrize("members", [1, 2])
def test_failed_group_submit_commits_nothing(model, members):
    """nrf_group_submit_host_u8 used to advance its slot and hand out the ticket before anything was validated; a
    submit that failed midway left a slot that a later wait reported as OK with stale frames.  A submit on a group WITHOUT a
    model fails; waiting on either ticket afterwards is NRF_E_STATE ("nothing was submitted"), and once a model is loaded the
    next submit takes ticket 0 again and returns the right bytes."""
    desc, keep, cfg = model
    W, H = 96, 64
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
    g = nh.NerfGroup([0] * members)
    g.set_resolution(W, H) if members == 1 else None
    with pytest.raises(nh.NerfHipError):
        g.submit_host_u8([cam], [pose])  # no model (and for several members: no resolution either)
    for t in (0, 1):
        with pytest.raises(nh.NerfHipError) as e:
            g.wait_host_u8(t)
        assert e.value.code == nh.NRF_E_STATE
    g.load_model(desc)
    g.set_resolution(W, H)
    want = _reference_u8(desc, W, H, [cam], [pose])
    t = g.submit_host_u8([cam], [pose])
    assert t == 0
    rgb, depth = g.wait_host_u8(t)
    np.testing.assert_array_equal(rgb[0], want[0][0])
    np.testing.assert_array_equal(depth[0], want[0][1])
    g.close()
