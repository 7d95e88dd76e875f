"""Render-buffer presentation chain (reference src/render_buffer.cu accumulate / tonemap): oracle known
answers on the CPU, HIP kernels against the oracle on the GPU, and the device-side hand-off
render -> frame buffer -> accumulate -> tonemap that replaces the reference's host round trip."""
import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn


def test_oracle_known_answers():
    acc = np.array([[0.2, 0.4, 0.6, 0.5], [1.0, 0.0, 0.25, 1.0]], np.float32)
    # Identity curve, linear in/out, white background with alpha 1: c + (1-a), a -> 1
    out = op.rb_tonemap(acc, 0.0, [1, 1, 1, 1], nh.CS_LINEAR, nh.CS_LINEAR, nh.TM_IDENTITY)
    np.testing.assert_allclose(out, [[0.7, 0.9, 1.1, 1.0], [1.0, 0.0, 0.25, 1.0]], rtol=1e-6)
    # exposure +1 doubles; sRGB output of 0.5 with the reference's 0.41666 exponent
    out = op.rb_tonemap(acc[1:], 1.0, [0, 0, 0, 0], nh.CS_LINEAR, nh.CS_SRGB, nh.TM_IDENTITY)
    want = 1.055 * (0.5 ** 0.41666) - 0.055
    assert out[0, 2] == pytest.approx(want, rel=1e-5) and out[0, 1] == 0.0
    # Reinhard: x / (1 + Y)
    out = op.rb_tonemap(acc[:1], 0.0, [0, 0, 0, 0], nh.CS_LINEAR, nh.CS_LINEAR, nh.TM_REINHARD)
    Y = 0.2126 * 0.2 + 0.7152 * 0.4 + 0.0722 * 0.6
    np.testing.assert_allclose(out[0, :3], np.array([0.2, 0.4, 0.6]) / (1 + Y), rtol=1e-6)
    # ACES and Hable map 0 -> 0 and are increasing, bounded near 1 for large inputs
    x = np.zeros((5, 4), np.float32); x[:, :3] = np.array([0, 0.1, 0.5, 2, 50], np.float32)[:, None]; x[:, 3] = 1
    for curve in (nh.TM_ACES, nh.TM_HABLE):
        y = op.rb_tonemap(x, 0.0, [0, 0, 0, 0], nh.CS_LINEAR, nh.CS_LINEAR, curve)[:, 0]
        assert y[0] == 0 and np.all(np.diff(y) > 0) and 0.8 < y[-1] < 1.3
    # accumulate: running mean over spp
    f1 = np.array([[1, 0, 0, 1]], np.float32); f2 = np.array([[0, 1, 0, 0]], np.float32)
    a = op.rb_accumulate(f1, np.zeros((1, 4), np.float32), 0, nh.CS_LINEAR)
    a = op.rb_accumulate(f2, a, 1, nh.CS_LINEAR)
    np.testing.assert_allclose(a, [[0.5, 0.5, 0, 0.5]])
    v = op.rb_accumulate(np.array([[0.2, 0.7, 0, 1]], np.float32), np.zeros((1, 4), np.float32), 0, nh.CS_VISPOSNEG)
    np.testing.assert_allclose(v[0, :2], [0.0, 0.5], atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cs,ocs,curve,exposure", [(0, 1, 1, 0.5), (0, 0, 0, 0.0), (1, 1, 2, -1.0), (0, 1, 3, 2.0), (2, 0, 0, 0.0)])
def test_hip_matches_oracle(cs, ocs, curve, exposure):
    torch = pytest.importorskip("torch")
    W, H = 67, 41
    rb = nh.RenderBuffer(0)
    rb.resize(W, H)
    rb.set_color_space(cs)
    rb.set_tonemap_curve(curve)
    frame_p, depth_p, acc_p, sur_p = rb.buffers()
    rng = np.random.default_rng(cs * 10 + curve)
    acc_ref = np.zeros((H * W, 4), np.float32)
    for spp in range(3):
        frame = rng.random((H * W, 4), dtype=np.float32) * np.float32(1.5)
        t = torch.from_numpy(frame).cuda()
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        torch.cuda.synchronize()
        assert hip.hipMemcpy(frame_p, t.data_ptr(), frame.nbytes, 3) == 0  # device-to-device: NOT host-synchronous
        torch.cuda.synchronize()
        rb.accumulate(0.0)
        acc_ref = op.rb_accumulate(frame, acc_ref, spp, cs)
    assert rb.spp() == 3
    bg = [0.3, 0.6, 0.9, 0.8]
    rb.tonemap(exposure, bg, ocs)
    acc, sur = rb.read()
    np.testing.assert_allclose(acc.reshape(-1, 4), acc_ref, rtol=2e-6, atol=1e-7)  # powf: device vs libm
    want = op.rb_tonemap(acc_ref, exposure, bg, cs, ocs, curve)
    np.testing.assert_allclose(sur.reshape(-1, 4), want, rtol=2e-5, atol=2e-6)
    rb.reset_accumulation()
    assert rb.spp() == 0
    rb.close()


@pytest.mark.gpu
def test_render_into_render_buffer_on_device():
    """nrf_render composites straight into the render buffer's frame plane (no host round trip),
    then accumulate + tonemap run on the device; equals the oracle chain on the oracle frame."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    W, H = 72, 56
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    rb = nh.RenderBuffer(0)
    rb.resize(W, H)
    rb.set_tonemap_curve(nh.TM_ACES)
    frame_p, depth_p, _, _ = rb.buffers()
    ctx.bind_output(frame_p, depth_p)
    cam = syn.default_camera(W, H)
    o = op.Oracle(desc)
    acc_ref = np.zeros((H * W, 4), np.float32)
    for spp, az in enumerate((20.0, 20.0)):
        rb.clear_frame()
        ctx.render(cam, syn.orbit_pose(az, 30))
        rb.accumulate(0.0)
        want, _, _ = o.render(cam, syn.orbit_pose(az, 30), W, H, schedule=op.SCHED_PER_RAY)
        acc_ref = op.rb_accumulate(want.reshape(-1, 4), acc_ref, spp, nh.CS_LINEAR)
    rb.tonemap(0.0, [1, 1, 1, 1], nh.CS_SRGB)
    acc, sur = rb.read()
    assert np.abs(acc.reshape(-1, 4) - acc_ref).max() <= 2.0 / 255.0
    want = op.rb_tonemap(acc_ref, 0.0, [1, 1, 1, 1], nh.CS_LINEAR, nh.CS_SRGB, nh.TM_ACES)
    assert np.abs(sur.reshape(-1, 4) - want).max() <= 3.0 / 255.0
    ctx.bind_output(None, None)
    ctx.close()
    rb.close()


def test_oracle_turbo_known_answers():
    """colormap_turbo end points and mid point (render_buffer.cu:413-429): source-derived known answers."""
    depth = np.array([[0.0, 0.5, 1.0, 7.0, -3.0]], np.float32)
    out = op.rb_overlay_depth(np.zeros((1, 5, 4), np.float32), 1.0, depth, 1.0, 0, 1.0, (0.5, 0.5))
    k = dict(r=(0.13572138, 4.61539260, -42.66032258, 132.13108234, -152.94239396, 59.28637943),
             g=(0.09140261, 2.19418839, 4.84296658, -14.18503333, 4.27729857, 2.82956604),
             b=(0.10667330, 12.64194608, -60.58204836, 110.36276771, -89.90310912, 27.34824973))
    for col, x in ((0, 0.0), (1, 0.5), (2, 1.0), (3, 1.0), (4, 0.0)):  # saturation of out-of-range depths
        want = [sum(c * x ** p for p, c in enumerate(k[ch])) for ch in "rgb"]
        np.testing.assert_allclose(out[0, col, :3], want, rtol=0, atol=2e-5)
        assert out[0, col, 3] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("alpha,fov_axis,zoom,center", [(1.0, 0, 1.0, (0.5, 0.5)), (0.35, 1, 1.7, (0.42, 0.61)), (0.5, 0, 0.6, (0.5, 0.5))])
def test_overlay_depth_matches_oracle(alpha, fov_axis, zoom, center):
    """nrf_rb_overlay_depth = overlay_depth_kernel (turbo colours, nearest-neighbour resampling, alpha blend)."""
    import torch
    rb = nh.RenderBuffer(0)
    W, H, iw, ih = 96, 54, 80, 60
    rb.resize(W, H)
    rng = np.random.default_rng(3)
    depth = rng.random((ih, iw), dtype=np.float32) * 1.3 - 0.1
    d_d = torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    rb.set_color_space(nh.CS_LINEAR)
    rb.tonemap(0.0, (0.2, 0.4, 0.6, 1.0), nh.CS_LINEAR)  # surface = background colour (the accumulate plane is empty)
    _, before = rb.read()
    assert np.all(before == before[0, 0]) and before.max() > 0
    rb.overlay_depth(alpha, d_d.data_ptr(), 0.9, iw, ih, fov_axis, zoom, center)
    _, after = rb.read()
    want = op.rb_overlay_depth(before, alpha, depth, 0.9, fov_axis, zoom, center)
    np.testing.assert_allclose(after, want, rtol=0, atol=1e-6)
    rb.close()


def _upload(frame_p, frame):
    import ctypes as C
    import torch
    t = torch.from_numpy(frame).cuda()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    torch.cuda.synchronize()
    assert hip.hipMemcpy(frame_p, t.data_ptr(), frame.nbytes, 3) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("cs,ocs,curve,exposure", [(0, 1, 1, 0.5), (0, 0, 0, 0.0), (1, 1, 2, -1.0), (0, 1, 3, 2.0), (2, 0, 0, 0.0), (0, 0, 2, 1.25)])
def test_present_is_accumulate_plus_tonemap_in_one_pass(cs, ocs, curve, exposure):
    """nrf_rb_present (one pass: frame -> mean -> surface -> packed 8-bit) against nrf_rb_accumulate + nrf_rb_tonemap (the
    reference's two calls, R/src/render_buffer.cu:590-627): the mean plane and the surface must hold the same BITS after
    every sample -- the first one included, for which the fused pass neither clears nor reads the mean plane (it is poisoned
    here) -- and rgba8 must be the library's 8-bit rule applied to the surface."""
    torch = pytest.importorskip("torch")
    W, H = 131, 57
    two, one = nh.RenderBuffer(0), nh.RenderBuffer(0)
    for rb in (two, one):
        rb.resize(W, H)
        rb.set_color_space(cs)
        rb.set_tonemap_curve(curve)
    f2, _, _, _ = two.buffers()
    f1, _, a1, _ = one.buffers()
    _upload(a1, np.full((H * W, 4), np.nan, np.float32))  # whatever the mean plane holds before the first sample is ignored
    rgba8 = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(7 + cs + 3 * curve)
    bg = [0.3, 0.6, 0.9, 0.8]
    for spp in range(3):
        frame = (rng.random((H * W, 4), dtype=np.float32) * np.float32(1.6) - np.float32(0.1)).astype(np.float32)
        _upload(f2, frame)
        _upload(f1, frame)
        two.accumulate(0.0)
        two.tonemap(exposure, bg, ocs)
        one.present(exposure, bg, ocs, rgba8.data_ptr())
        assert one.spp() == two.spp() == spp + 1
        acc2, sur2 = two.read()
        acc1, sur1 = one.read()
        np.testing.assert_array_equal(acc1.view(np.uint32), acc2.view(np.uint32))
        np.testing.assert_array_equal(sur1.view(np.uint32), sur2.view(np.uint32))
        rgb8, a8 = op.quantize_u8(sur1.reshape(-1, 4), sur1.reshape(-1, 4)[:, 3].copy())
        want = (rgb8[:, 0].astype(np.uint32) | (rgb8[:, 1].astype(np.uint32) << 8) | (rgb8[:, 2].astype(np.uint32) << 16) |
                (a8.astype(np.uint32) << 24))
        np.testing.assert_array_equal(rgba8.cpu().numpy().view(np.uint32).reshape(-1), want)
    one.present(exposure, bg, ocs)  # without the 8-bit plane
    assert one.spp() == 4
    two.close()
    one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3])
def test_present_is_bit_exact_against_the_oracle_where_no_power_function_is_involved(curve):
    """Linear render and display spaces: every operation of the chain is an individually rounded fp32 add / multiply / divide /
    max on both sides, and what does not depend on the pixel (2^exposure, the background's sRGB decode, the curve's
    coefficients) is computed on the host with the oracle's own libm: bit-exact.  (With sRGB on either side the device's
    powf differs from libm's in the last place: test_hip_matches_oracle states that tolerance.)"""
    W, H = 96, 40
    rb = nh.RenderBuffer(0)
    rb.resize(W, H)
    rb.set_color_space(nh.CS_LINEAR)
    rb.set_tonemap_curve(curve)
    frame_p, _, _, _ = rb.buffers()
    rng = np.random.default_rng(curve)
    acc_ref = np.zeros((H * W, 4), np.float32)
    bg = [0.25, 0.5, 0.75, 0.6]
    for spp in range(3):
        frame = (rng.random((H * W, 4), dtype=np.float32) * np.float32(2.5)).astype(np.float32)
        _upload(frame_p, frame)
        rb.present(0.75, bg, nh.CS_LINEAR)
        acc_ref = op.rb_accumulate(frame, acc_ref, spp, nh.CS_LINEAR)
        acc, sur = rb.read()
        np.testing.assert_array_equal(acc.reshape(-1, 4), acc_ref)
        np.testing.assert_array_equal(sur.reshape(-1, 4), op.rb_tonemap(acc_ref, 0.75, bg, nh.CS_LINEAR, nh.CS_LINEAR, curve))
    rb.close()
