"""Parity of the gfx950 HIP path (through the C ABI) against the CPU oracle.

Tolerances (stated per test):
  * ray generation, near/far, marching, hash-grid encoding, SH encoding:
    BIT-EXACT (same individually-rounded fp32 / fp16 operations on both sides).
  * MLP outputs: the HIP path accumulates each layer in fp32 inside the MFMA
    (unspecified summation order), the oracle in fp32 ascending-k; both round
    to fp16 after every layer.  A pre-rounding difference of ~1e-7 flips an
    fp16 rounding now and then, so outputs agree to a few fp16 ulps:
    |d| <= 4*2^-11*|x| + 2e-3.
  * compositing: __expf (v_exp_f32) vs libm expf -> 2e-5 absolute.
  * rendered frame (float RGBA before u8): max |d| <= 2/255 and
    PSNR >= 45 dB against the oracle's PER_RAY schedule (the reference loop at n_step == 1, which
    is the per-ray semantics the kernel implements), and 2/255 against the other schedules.
"""
import ctypes as C

import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def sync():
    torch.cuda.synchronize()


def mlp_close(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), what
    tol = 4 * 2.0 ** -11 * np.abs(want[fin]) + 2e-3
    err = np.abs(got[fin] - want[fin])
    assert np.all(err <= tol), f"{what}: worst {float((err / tol).max()):.2f}x tolerance"


@pytest.fixture(scope="module")
def ctx():
    h = nh.NerfHip(0)
    yield h
    h.close()


@pytest.fixture(scope="module")
def small(ctx):
    """T = 2^12 table, H = 32 grid: the oracle renders small frames in well under a second."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    ctx.load_model(desc)
    return desc, keep, op.Oracle(desc)


def test_mfma_layout_identity_weights(ctx):
    """A = I style check of the fragment packing with an ASYMMETRIC second operand: a model whose
    layers route single inputs to single outputs must reproduce them exactly."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    p = keep[0].copy()
    D0 = np.zeros((64, 32), np.float32); D1 = np.zeros((16, 64), np.float32)
    R0 = np.zeros((64, 32), np.float32); R1 = np.zeros((64, 64), np.float32); R2 = np.zeros((16, 64), np.float32)
    for o_ in range(32):
        D0[o_, o_] = 1.0          # hidden[o] = relu(feat[o])
    for o_ in range(16):
        D1[o_, 2 * o_ + 1] = 1.0  # dens[o] = hidden[2o+1]   (asymmetric pick)
    for o_ in range(32):
        R0[o_, 31 - o_] = 1.0     # h1[o] = relu(rgbin[31-o]) (reversal catches k-order mistakes)
    for o_ in range(64):
        R1[o_, (o_ * 5 + 3) % 64] = 1.0
    R2[0, 7] = 1.0; R2[1, 20] = 1.0; R2[2, 41] = 1.0
    p[:syn.N_MLP] = np.concatenate([m.reshape(-1) for m in (D0, D1, R0, R1, R2)])
    desc2, keep2 = nh.desc_from_config({**cfg}, p, keep[1])
    ctx.load_model(desc2)
    rng = np.random.default_rng(5)
    n = 200
    feat = rng.uniform(0.0, 2.0, (n, 32)).astype(np.float16)       # positive: relu is the identity
    dirf = rng.uniform(0.0, 2.0, (n, 16)).astype(np.float16)
    out = torch.empty((n, 4), dtype=torch.float16, device="cuda")
    f_d, d_d = dev(feat.view(np.uint16).view(np.int16)), dev(dirf.view(np.uint16).view(np.int16))
    sync()
    ctx.mlp_forward(f_d.data_ptr(), d_d.data_ptr(), n, out.data_ptr())
    got = out.cpu().numpy().astype(np.float32)
    dens = feat[:, 1::2].astype(np.float32)[:, :16]               # dens[o] = feat[2o+1]
    rgbin = np.concatenate([dens, dirf.astype(np.float32)], axis=1)
    h1 = rgbin[:, ::-1]                                           # h1[o] = rgbin[31-o], o < 32; 0 above
    h1 = np.concatenate([h1, np.zeros((n, 32), np.float32)], axis=1)
    h2 = h1[:, [(o_ * 5 + 3) % 64 for o_ in range(64)]]
    np.testing.assert_array_equal(got[:, 0], h2[:, 7])
    np.testing.assert_array_equal(got[:, 1], h2[:, 20])
    np.testing.assert_array_equal(got[:, 2], h2[:, 41])
    want_sigma = np.exp(dens[:, 0].astype(np.float32)).astype(np.float16).astype(np.float32)
    np.testing.assert_allclose(got[:, 3], want_sigma, rtol=2e-3)
    want = op.Oracle(desc2).mlp_forward(feat.view(np.uint16), dirf.view(np.uint16)).view(np.float16).astype(np.float32)
    np.testing.assert_array_equal(got[:, :3], want[:, :3])


@pytest.mark.parametrize("log2T", [12, 19])
def test_encode_grid_bit_exact(ctx, log2T):
    desc, keep, cfg = models.build_model(log2_hashmap_size=log2T, H=32)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    rng = np.random.default_rng(1)
    pos = rng.random((5000, 3), dtype=np.float32)
    edge = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 1, 0], [0.25, 0.5, 0.75], [1 - 2 ** -24, 2 ** -24, 0.5],
                     [0.5, 0.5, 0.5], [1 / 3, 2 / 3, 1.0]], np.float32)
    pos = np.concatenate([edge, pos])
    want = o.encode_grid(pos)
    out = torch.empty((len(pos), 32), dtype=torch.int16, device="cuda")
    p_d = dev(pos)
    sync()
    ctx.encode_grid(p_d.data_ptr(), len(pos), out.data_ptr())
    got = out.cpu().numpy().view(np.uint16)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("kw", [dict(sh_degree=4), dict(sh_degree=3), dict(sh_degree=2),
                                dict(dir_otype="Frequency", n_frequencies=2), dict(dir_otype="Identity")])
def test_encode_dir(ctx, kw):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, **kw)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    rng = np.random.default_rng(2)
    # many directions: a compiler-fused v_fma_mixlo_f16 (one rounding instead of fp32-then-fp16) shows up in only
    # 1 of ~130 000 coefficients
    d = rng.normal(size=(400000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d01 = (d * np.float32(0.5) + np.float32(0.5)).astype(np.float32)
    want = o.encode_dir(d01)
    out = torch.empty((len(d01), 16), dtype=torch.int16, device="cuda")
    d_d = dev(d01)
    sync()
    ctx.encode_dir(d_d.data_ptr(), len(d01), out.data_ptr())
    got = out.cpu().numpy().view(np.uint16)
    if kw.get("dir_otype") == "Frequency":  # __sinf vs sinf: a few fp16 ulps near zero crossings
        np.testing.assert_allclose(got.view(np.float16).astype(np.float32), want.view(np.float16).astype(np.float32), atol=2e-3)
    else:
        np.testing.assert_array_equal(got, want)


def test_mlp_forward_and_network(ctx, small):
    desc, keep, o = small
    ctx.load_model(desc)
    rng = np.random.default_rng(3)
    n = 4099  # ragged: not a multiple of 64
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p01 = (np.float32(0.5) * xyz + np.float32(0.5)).astype(np.float32)
    d01 = (np.float32(0.5) * d + np.float32(0.5)).astype(np.float32)
    feat, dirf = o.encode_grid(p01), o.encode_dir(d01)
    want = o.mlp_forward(feat, dirf).view(np.float16).astype(np.float32)
    out = torch.empty((n, 4), dtype=torch.float16, device="cuda")
    f_d, d_d = dev(feat.view(np.int16)), dev(dirf.view(np.int16))
    sync()
    ctx.mlp_forward(f_d.data_ptr(), d_d.data_ptr(), n, out.data_ptr())
    got = out.cpu().numpy().astype(np.float32)
    mlp_close(got[:, :3], want[:, :3], "rgb (mlp_forward)")
    # sigma = exp(g0): compare in log space (g0 itself obeys the MLP tolerance)
    mlp_close(np.log(got[:, 3]), np.log(want[:, 3]), "log sigma (mlp_forward)")

    # the whole network from raw march output, through the render kernel's own code path
    sig_w, rgb_w = o.network(xyz, d)
    sig = torch.empty(n, dtype=torch.float32, device="cuda")
    rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    x_d, dd = dev(xyz), dev(d)
    sync()
    ctx.network(x_d.data_ptr(), dd.data_ptr(), n, sig.data_ptr(), rgb.data_ptr())
    mlp_close(rgb.cpu().numpy(), rgb_w, "rgb (network)")
    mlp_close(np.log(sig.cpu().numpy()), np.log(sig_w), "log sigma (network)")
    # empty input is a no-op, not an error
    ctx.network(x_d.data_ptr(), dd.data_ptr(), 0, sig.data_ptr(), rgb.data_ptr())


def _rays(ctx, o, W, H, cam, pose):
    ctx.set_resolution(W, H)
    n = W * H
    ro = torch.empty((n, 3), device="cuda"); rd = torch.empty((n, 3), device="cuda")
    nr = torch.empty(n, device="cuda"); fr = torch.empty(n, device="cuda")
    sync()
    ctx.generate_rays(cam, pose, ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr())
    return ro, rd, nr, fr


def test_generate_rays_bit_exact(ctx, small):
    desc, keep, o = small
    ctx.load_model(desc)
    W, H = 50, 37
    for pose in (syn.orbit_pose(10, 30), syn.orbit_pose(200, -15), syn.REFERENCE_MAIN_POSE):
        cam = syn.default_camera(W, H)
        ro, rd, nr, fr = _rays(ctx, o, W, H, cam, pose)
        wo, wd, wn, wf = o.generate_rays(cam, pose, W, H)
        np.testing.assert_array_equal(ro.cpu().numpy(), wo)
        np.testing.assert_array_equal(rd.cpu().numpy(), wd)
        np.testing.assert_array_equal(nr.cpu().numpy(), wn)
        np.testing.assert_array_equal(fr.cpu().numpy(), wf)


@pytest.mark.parametrize("bound,cascade", [(1.0, 1), (4.0, 3)])
def test_march_bit_exact_and_composite(ctx, bound, cascade):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, bound=bound, cascade=cascade)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 48, 40
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(45, 25)
    ro, rd, nr, fr = _rays(ctx, o, W, H, cam, pose)
    n = W * H
    for n_step in (1, 3, 8):
        xyzs = torch.empty((n, n_step, 3), device="cuda"); dirs = torch.empty((n, n_step, 3), device="cuda")
        deltas = torch.empty((n, n_step, 2), device="cuda")
        sync()
        ctx.march(ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr(), n, n_step, xyzs.data_ptr(), dirs.data_ptr(),
                  deltas.data_ptr())
        wx, wd, wdl = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), n_step)
        np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
        np.testing.assert_array_equal(dirs.cpu().numpy(), wd)
        np.testing.assert_array_equal(deltas.cpu().numpy(), wdl)
        assert (wdl[:, :, 0] > 0).sum() > 0
    # composite on those samples with synthetic sigma/rgb
    rng = np.random.default_rng(4)
    sig = rng.uniform(0, 400, (n, 8)).astype(np.float32)
    rgb = rng.random((n, 8, 3), dtype=np.float32)
    state = np.zeros((n, 5), np.float32)
    t0 = nr.cpu().numpy()
    wt, wst = op.composite(sig, rgb, wdl, t0, state)
    st_d, t_d = dev(state), dev(t0)
    s_d, r_d = dev(sig), dev(rgb)
    sync()
    ctx.composite(s_d.data_ptr(), r_d.data_ptr(), deltas.data_ptr(), n, 8, t_d.data_ptr(), st_d.data_ptr())
    np.testing.assert_allclose(st_d.cpu().numpy(), wst, atol=2e-5)
    got_t = t_d.cpu().numpy()
    dead_w, dead_g = wt < 0, got_t < 0
    assert (dead_w != dead_g).mean() < 1e-3  # a T ~ 1e-4 tie may fall either way
    same = ~dead_w & ~dead_g
    np.testing.assert_allclose(got_t[same], wt[same], rtol=1e-6)


def _render_both(ctx, o, W, H, cam, pose, opts=None):
    ctx.set_options(opts or nh.default_options())
    ctx.set_resolution(W, H)
    ctx.render(cam, pose)
    rgba, depth = ctx.read_f32()
    st = ctx.stats()
    want, wdepth, wst = o.render(cam, pose, W, H, opts=opts, schedule=op.SCHED_PER_RAY)
    return rgba, depth, st, want, wdepth, wst


@pytest.mark.parametrize("W,H,az,el", [(64, 64, 30, 30), (100, 52, 135, 10), (8, 8, 300, 45), (33, 70, 250, -20)])
def test_render_frame_matches_oracle(ctx, small, W, H, az, el):
    desc, keep, o = small
    ctx.load_model(desc)
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(az, el)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert np.all(np.isfinite(rgba)) and np.all(np.isfinite(depth))
    assert np.abs(rgba - want).max() <= 2.0 / 255.0
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0
    assert models.psnr(rgba, want) >= 45.0
    # the kernel batches up to 8 samples per ray and round, so it may evaluate a few samples past a ray's termination that
    # the one-sample-at-a-time oracle never emits: how many depends on timing (tail splitting hands rays to idle waves, and a
    # tiny frame is all tail -- which is why it keeps the transmittance-dependent queue since round 5).  Measured on these
    # frames: 0.7-2.9 % (profiles/r05/waste_small.txt); guarded at 15 % (+ 32 samples: the 8x8 frame composites 628).
    # (The full-size test below holds a 1080p view to 10 % and a 16-view launch to 4 %.)
    assert st.n_composited <= st.n_samples <= 1.15 * st.n_composited + 32, (st.n_samples, st.n_composited)
    # ... while the samples that reach a ray's compositing sum are the oracle's own (per-ray schedule), up to the rays whose
    # termination test falls the other way within the MLP tolerance
    assert abs(int(st.n_composited) - int(wst.n_composited)) <= 0.002 * wst.n_composited + 8 and wst.n_composited == wst.n_samples
    assert st.n_rays == (((W + 7) // 8 + 3) // 4) * 4 * ((H + 7) // 8) * 64  # whole strips of 4 tiles
    # the reference's own (global) schedule gives the same picture
    for sched in (op.SCHED_REFERENCE, op.SCHED_TILE64):
        ref, rdepth, _ = o.render(cam, pose, W, H, schedule=sched)
        assert np.abs(rgba - ref).max() <= 2.0 / 255.0 and np.abs(depth - rdepth).max() <= 2.0 / 255.0
    # u8 output = saturating quantisation of the float frame (nerf_render.cu:352-359)
    rgb8, d8 = ctx.read_u8()
    w8, wd8 = op.quantize_u8(rgba, depth)
    np.testing.assert_array_equal(rgb8, w8)
    np.testing.assert_array_equal(d8, wd8)


def test_render_options_and_multilevel_scene(ctx):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, bound=4.0, cascade=3)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 72, 48
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(80, 35)
    opts = nh.default_options()
    opts.bg_color, opts.density_scale, opts.max_steps, opts.min_near = 0.25, 0.5, 64, 0.05
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose, opts)
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0
    ctx.set_options(nh.default_options())


def test_model_without_occupied_cells_and_errors(ctx):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    empty = np.zeros_like(keep[1])
    d2, k2 = nh.desc_from_config(cfg, keep[0], empty)
    ctx.load_model(d2)
    ctx.set_resolution(16, 16)
    ctx.render(syn.default_camera(16, 16), syn.orbit_pose(0))
    rgba, depth = ctx.read_f32()
    assert np.all(rgba[..., :3] == 1.0) and np.all(rgba[..., 3] == 0) and np.all(depth == 0)
    assert ctx.stats().n_samples == 0
    # error behaviour of the reference: param-count and grid-size mismatches (nerf_network.h:425, nerf_render.cu:467)
    bad = nh.ModelDesc.from_buffer_copy(desc); bad.n_params -= 2
    with pytest.raises(nh.NerfHipError) as e:
        ctx.load_model(bad)
    assert e.value.code == nh.NRF_E_PARAMS
    bad = nh.ModelDesc.from_buffer_copy(desc); bad.cascade = 2
    with pytest.raises(nh.NerfHipError) as e:
        ctx.load_model(bad)
    assert e.value.code == nh.NRF_E_PARAMS
    bad = nh.ModelDesc.from_buffer_copy(desc); bad.n_neurons = 48  # FullyFusedMLP: 16, 32, 64 or 128 (fully_fused_mlp.cu:700-725)
    with pytest.raises(nh.NerfHipError) as e:
        ctx.load_model(bad)
    assert e.value.code == nh.NRF_E_INVALID
    fresh = nh.NerfHip(0)
    with pytest.raises(nh.NerfHipError) as e:
        fresh.render(syn.default_camera(8, 8), syn.orbit_pose(0))
    assert e.value.code == nh.NRF_E_STATE
    fresh.close()


def test_full_size_properties_1080p(ctx):
    """BASELINE config 2 (1920x1080, L=16 F=2 T=2^19, 64-wide MLPs): size-independent properties."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 1920, 1080
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.render(cam, pose)
    a, da = ctx.read_f32()
    sa = ctx.stats()
    # deterministic, bit for bit, over many launches.  (Regression test: with hipcc's SLP-packed
    # v_pk_*_f32 ops ~5 % of 1080p frames had one 8x8 tile with wrong colours in lanes 48-63.)
    for _ in range(40):
        ctx.render(cam, pose)
        b, db = ctx.read_f32()
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(da, db)
    assert np.all(np.isfinite(a)) and a[..., 3].min() >= 0 and a[..., 3].max() <= 1 + 1e-5
    assert da.min() >= 0 and sa.n_samples > 1_000_000
    # wasted network evaluations (samples a ray queued behind its terminating one) are guarded: a view alone <= 10 %,
    # the 16-view launch of bench.py <= 4 % (measured with the transmittance-dependent queue: < 1 %)
    assert sa.n_composited <= sa.n_samples <= 1.10 * sa.n_composited, (sa.n_samples, sa.n_composited)
    ctx.set_max_views(16)
    ctx.render_views(np.stack([cam] * 16), np.stack([syn.orbit_pose(45.0 * (i % 8), 30.0) for i in range(16)]))
    sb = ctx.stats()
    assert sb.n_composited <= sb.n_samples <= 1.04 * sb.n_composited, (sb.n_samples, sb.n_composited)
    ctx.render(cam, pose)  # (the single view is the frame the checks below read)
    # rays that miss the aabb are exactly background
    _, _, nr, fr = o.generate_rays(cam, pose, W, H)
    miss = (nr >= fr).reshape(H, W)
    if miss.any():
        assert np.all(a[miss][:, :3] == 1.0) and np.all(a[miss][:, 3] == 0)
    # a 128x64 crop rendered by the oracle (same rays via a shifted principal point) matches
    x0, y0, cw, ch = 896, 508, 128, 64
    ccam = cam.copy(); ccam[2] -= x0; ccam[3] -= y0
    want, wd, _ = o.render(ccam, pose, cw, ch, schedule=op.SCHED_PER_RAY)
    crop = a[y0:y0 + ch, x0:x0 + cw]
    assert np.abs(crop - want).max() <= 2.0 / 255.0 and models.psnr(crop, want) >= 45.0
    # tile sharding: every shard count reproduces the single-shard frame bit for bit after untile
    for count in (2, 8):
        tps = nh.tiles_per_shard(W, H, count)
        gathered = torch.zeros((count, tps * 64, 4), device="cuda")
        gdepth = torch.zeros((count, tps * 64, 1), device="cuda")
        total = 0
        for idx in range(count):
            opts = nh.default_options(); opts.shard_index, opts.shard_count = idx, count
            ctx.set_options(opts)
            f = ctx.render(cam, pose)
            assert f.tile_major == 1
            total += ctx.stats().n_composited
            n_px = f.n_tiles * 64
            shard = torch.empty((n_px, 4), device="cuda"); sdepth = torch.empty((n_px, 1), device="cuda")
            sync()
            _d2d(shard.data_ptr(), f.rgba, n_px * 16); _d2d(sdepth.data_ptr(), f.depth, n_px * 4)
            gathered[idx, :n_px] = shard; gdepth[idx, :n_px] = sdepth
        assert total == sa.n_composited  # (n_samples depends on the batching of rays into rounds, which sharding changes)
        out = torch.empty((H, W, 4), device="cuda"); outd = torch.empty((H, W, 1), device="cuda")
        sync()
        ctx.untile(gathered.data_ptr(), count, tps, 4, out.data_ptr())
        ctx.untile(gdepth.data_ptr(), count, tps, 1, outd.data_ptr())
        np.testing.assert_array_equal(out.cpu().numpy(), a)
        np.testing.assert_array_equal(outd.cpu().numpy()[..., 0], da)
    ctx.set_options(nh.default_options())


def _d2d(dst, src, nbytes):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst, src, nbytes, 3) == 0  # hipMemcpyDeviceToDevice (not host-synchronous)
    hip.hipDeviceSynchronize()


def test_generic_activation_kernel_instance(ctx):
    """rgb output Sigmoid + sigma... selects the GEN=true kernel instances (runtime activations)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, rgb_output_activation="Sigmoid")
    assert desc.rgb_output_activation == nh.ACT["sigmoid"]
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 56, 40
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(120, 25)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
    assert rgba[..., :3].max() <= 1.0 + 1e-3  # sigmoid colours + white background stay in range
    rng = np.random.default_rng(7)
    n = 1000
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sig_w, rgb_w = o.network(xyz, d)
    sig = torch.empty(n, dtype=torch.float32, device="cuda")
    rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    x_d, dd = dev(xyz), dev(d)
    sync()
    ctx.network(x_d.data_ptr(), dd.data_ptr(), n, sig.data_ptr(), rgb.data_ptr())
    mlp_close(rgb.cpu().numpy(), rgb_w, "rgb (generic network)")
    mlp_close(np.log(sig.cpu().numpy()), np.log(sig_w), "log sigma (generic network)")


def test_config4_large_bound_five_cascades(ctx):
    """BASELINE config 4 shape: bound 16, 5 density-grid cascades, T = 2^19, up to 1024 samples per ray
    (the generic march path: levels from frexp, cell tables read from LDS for 5 cascades)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128, bound=16.0, cascade=5)
    assert nh.level_table(desc).offset[16] == 6811592  # SURVEY Appendix C, bound 16
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 96, 64
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(60, 25)
    # bit-exact marching through the cascades
    ro, rd, nr, fr = _rays(ctx, o, W, H, cam, pose)
    n = W * H
    xyzs = torch.empty((n, 8, 3), device="cuda"); dirs = torch.empty((n, 8, 3), device="cuda")
    deltas = torch.empty((n, 8, 2), device="cuda")
    sync()
    ctx.march(ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr(), n, 8, xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr())
    wx, wd, wdl = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), 8)
    np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
    np.testing.assert_array_equal(deltas.cpu().numpy(), wdl)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert st.n_samples > 0
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0


def test_config4_full_size_properties_1080p():
    """BASELINE config 4 at FULL size (1920x1080; bound 16, five cascades, 1024 samples per ray -- the synthetic stand-in for
    the "real-captured 360 scene": the reference ships none): size-independent properties, for a camera outside the volume
    and one inside it.  Bit-identical over 5 launches, alpha in [0, 1], finite depth, rays that miss the aabb are
    background, a 128x64 crop against the oracle (stated tolerance: 2/255, PSNR >= 45 dB), the per-strip scheduling
    (NRF_PERSISTENT=0) bit-identical to the persistent one, a 16-view launch bit-identical to single renders, and the
    evaluated samples within 4 % (batch) / 10 % (alone) of the composited ones."""
    import os

    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)
    o = op.Oracle(desc)
    W, H = 1920, 1080
    cam = syn.default_camera(W, H)
    opts = nh.default_options(); opts.max_steps = 1024
    frames = {}
    for persistent in ("1", "0"):
        os.environ["NRF_PERSISTENT"] = persistent
        try:
            c = nh.NerfHip(0)
        finally:
            os.environ.pop("NRF_PERSISTENT", None)
        c.load_model(desc)
        c.set_options(opts)
        c.set_resolution(W, H)
        for name, pose in (("outside", syn.orbit_pose(60, 25)), ("inside", syn.orbit_pose(120, -15, radius=1.5 / 0.33))):
            c.render(cam, pose)
            a, da = c.read_f32()
            st = c.stats()
            frames[(persistent, name)] = (a, da)
            if persistent == "0":
                continue
            for _ in range(5):
                c.render(cam, pose)
                b, db = c.read_f32()
                np.testing.assert_array_equal(a, b)
                np.testing.assert_array_equal(da, db)
            assert np.all(np.isfinite(a)) and np.all(np.isfinite(da)) and a[..., 3].min() >= 0 and a[..., 3].max() <= 1 + 1e-5
            assert st.n_composited > 1_000_000 and st.n_composited <= st.n_samples <= 1.10 * st.n_composited
            _, _, nr, fr = o.generate_rays(cam, pose, W, H, opts)
            miss = (nr >= fr).reshape(H, W)
            if miss.any():
                assert np.all(a[miss][:, :3] == 1.0) and np.all(a[miss][:, 3] == 0)
            x0, y0, cw, ch = 896, 508, 128, 64
            ccam = cam.copy(); ccam[2] -= x0; ccam[3] -= y0
            want, wd, _ = o.render(ccam, pose, cw, ch, opts, schedule=op.SCHED_PER_RAY)
            crop = a[y0:y0 + ch, x0:x0 + cw]
            assert np.abs(crop - want).max() <= 2.0 / 255.0 and models.psnr(crop, want) >= 45.0, name
            assert np.abs(da[y0:y0 + ch, x0:x0 + cw] - wd).max() <= 2.0 / 255.0, name
        if persistent == "1":  # the 16-view launch of bench.py's `configs` object: views bit-identical to single renders
            c.set_max_views(16)
            poses = [syn.orbit_pose(360.0 * i / 16, 25.0) for i in range(16)]
            c.render_views([cam] * 16, poses)
            sb = c.stats()
            assert sb.n_composited <= sb.n_samples <= 1.04 * sb.n_composited, (sb.n_samples, sb.n_composited)
            batch = [c.read_view_f32(v) for v in (0, 5, 15)]
            for (rgba, depth), v in zip(batch, (0, 5, 15)):
                c.render(cam, poses[v])
                r1, d1 = c.read_f32()
                np.testing.assert_array_equal(rgba, r1)
                np.testing.assert_array_equal(depth, d1)
        c.close()
    for name in ("outside", "inside"):
        np.testing.assert_array_equal(frames[("0", name)][0], frames[("1", name)][0])
        np.testing.assert_array_equal(frames[("0", name)][1], frames[("1", name)][1])


def test_config5_batched_views_800x800(ctx):
    """BASELINE config 5 shape: independent 800x800 camera requests rendered back to back on one
    context must equal the same views rendered alone (no state leaks between requests), and a crop
    of each matches the oracle."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W = H = 800
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(az, el) for az, el in ((0, 30), (95, 10), (190, 45), (300, -10))]
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    batch = []
    for p in poses:  # queued on the context's stream without waiting in between
        f = nh.Frame()
        cam_c = np.ascontiguousarray(cam, np.float32); pose_c = np.ascontiguousarray(p, np.float32).reshape(16)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        rc = ctx.lib.nrf_render_async(ctx.h, fp(cam_c), fp(pose_c), C.byref(f))
        assert rc == 0
        rc = ctx.lib.nrf_sync(ctx.h)
        assert rc == 0
        batch.append(ctx.read_f32()[0].copy())
    for p, got in zip(poses, batch):
        fresh = nh.NerfHip(0)
        fresh.load_model(desc)
        fresh.set_resolution(W, H)
        fresh.render(cam, p)
        np.testing.assert_array_equal(fresh.read_f32()[0], got)
        fresh.close()
        x0, y0, cw, ch = 368, 376, 64, 48
        ccam = cam.copy(); ccam[2] -= x0; ccam[3] -= y0
        want, _, _ = o.render(ccam, p, cw, ch, schedule=op.SCHED_PER_RAY)
        assert np.abs(got[y0:y0 + ch, x0:x0 + cw] - want).max() <= 2.0 / 255.0


def test_render_views_one_launch_equals_single_renders(ctx):
    """nrf_render_views (BASELINE config 5, render_server batching): every view of a batched launch is
    bit-identical to nrf_render of that camera -- 11 views (two launches: 8 + 3), different intrinsics
    per view, unsharded and as a tile-major shard with the documented view stride."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    ctx.load_model(desc)
    W, H = 200, 120
    n = nh.NRF_MAX_VIEWS + 3
    cams = np.stack([syn.default_camera(W, H) * np.float32(1.0 + 0.03 * i) for i in range(n)])
    poses = np.stack([syn.orbit_pose(33.0 * i, 10.0 + 4.0 * i) for i in range(n)])
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    single = []
    for i in range(n):
        ctx.render(cams[i], poses[i])
        single.append(tuple(x.copy() for x in ctx.read_f32()))
    with pytest.raises(nh.NerfHipError):
        ctx.render_views(cams, poses)  # the context's own buffers hold one view until set_max_views
    ctx.set_max_views(n)
    f = ctx.render_views(cams, poses)
    assert f.n_views == n and f.view_stride_px == W * H and ctx.stats().n_samples > 0
    for i in range(n):
        rgba, depth = ctx.read_view_f32(i)
        np.testing.assert_array_equal(rgba, single[i][0])
        np.testing.assert_array_equal(depth, single[i][1])
    # sharded: views are tile-major shards, view_stride_px apart, in caller-owned memory
    opts = nh.default_options(); opts.shard_index, opts.shard_count = 1, 2
    ctx.set_options(opts)
    tps = nh.tiles_per_shard(W, H, 2)
    want = []
    for i in range(n):
        fr = ctx.render(cams[i], poses[i])
        assert fr.tile_major == 1 and fr.view_stride_px == tps * 64
        s = torch.empty((tps * 64, 4), device="cuda")
        sync(); _d2d(s.data_ptr(), fr.rgba, tps * 64 * 16)
        want.append(s.cpu().numpy())
    out = torch.zeros((n, tps * 64, 4), device="cuda"); outd = torch.zeros((n, tps * 64), device="cuda")
    sync()
    ctx.bind_output(out.data_ptr(), outd.data_ptr())
    fb = ctx.render_views(cams, poses)
    ctx.bind_output(None, None)
    assert fb.n_views == n and fb.view_stride_px == tps * 64
    got = out.cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(got[i], want[i])
    ctx.set_options(nh.default_options())
    ctx.set_max_views(1)


def test_quantize_rgbd8_matches_reference_packing(ctx):
    """nrf_quantize_rgbd8 = (unsigned char)(255.0 * x) of nerf_render.cu:352-359 (saturating, NaN -> 0) packed
    r | g << 8 | b << 16 | depth << 24, bit for bit the oracle's nrfo_quantize_u8."""
    rng = np.random.default_rng(5)
    n = 10007
    rgba = rng.uniform(-0.2, 1.2, (n, 4)).astype(np.float32)
    depth = rng.uniform(-0.2, 1.2, n).astype(np.float32)
    rgba[:8, 0] = [0.0, 1.0, np.nan, np.inf, -np.inf, 0.5, 1 / 255.0, 254.999 / 255.0]
    rgb8, d8 = op.quantize_u8(rgba, depth)
    want = (rgb8[:, 0].astype(np.uint32) | (rgb8[:, 1].astype(np.uint32) << 8) | (rgb8[:, 2].astype(np.uint32) << 16) |
            (d8.astype(np.uint32) << 24))
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    r_d, d_d = dev(rgba), dev(depth)
    sync()
    ctx.quantize_rgbd8(r_d.data_ptr(), d_d.data_ptr(), n, out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), want)


def test_device_group_equals_single_context(ctx):
    """nrf_group_*: the one-process multi-device form (the reference's NGPU).  Two and three members --
    all on device 0, the only one this box has -- render their strips, ship them device-to-device and
    untile; frames, u8 images and sample counts must equal a single context's, bit for bit."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H = 328, 200  # ragged strips (41 tiles per row) and rows (25 tile rows)
    n = 5
    cams = np.stack([syn.default_camera(W, H)] * n)
    poses = np.stack([syn.orbit_pose(50.0 * i, 20.0) for i in range(n)])
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.set_max_views(n)
    ctx.render_views(cams, poses)
    want = [ctx.read_view_f32(i) for i in range(n)]
    want_u8 = [ctx.read_view_u8(i) for i in range(n)]
    want_samples = ctx.stats().n_composited  # (evaluated samples depend on the batching of rays into rounds; these do not)
    ctx.set_max_views(1)
    for members in (1, 2, 3):
        g = nh.NerfGroup([0] * members)
        g.load_model(desc)
        g.set_resolution(W, H)
        f = g.render_views(cams, poses)
        assert f.n_views == n and f.tile_major == 0 and f.view_stride_px == W * H
        assert g.stats().n_composited == want_samples
        for i in range(n):
            rgba, depth = g.read_view_f32(i)
            np.testing.assert_array_equal(rgba, want[i][0])
            np.testing.assert_array_equal(depth, want[i][1])
            rgb8, d8 = g.read_view_u8(i)
            np.testing.assert_array_equal(rgb8, want_u8[i][0])
            np.testing.assert_array_equal(d8, want_u8[i][1])
        g.close()


def test_device_group_on_distinct_devices(ctx):
    """nrf_group_* with members on DIFFERENT devices: hipDeviceEnablePeerAccess + hipMemcpyPeerAsync over xGMI
    (csrc/nrf_group.hip), which the single-device rehearsal above cannot exercise.  Runs wherever at least two GPUs
    are visible (the driver's 8-GPU node); skipped on a one-GPU box."""
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs at least two visible GPUs")
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H = 328, 200
    n = 3
    cams = np.stack([syn.default_camera(W, H)] * n)
    poses = np.stack([syn.orbit_pose(70.0 * i, 25.0) for i in range(n)])
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.set_max_views(n)
    ctx.render_views(cams, poses)
    want = [ctx.read_view_f32(i) for i in range(n)]
    want_u8 = [ctx.read_view_u8(i) for i in range(n)]
    want_samples = ctx.stats().n_composited  # (evaluated samples depend on the batching of rays into rounds; these do not)
    ctx.set_max_views(1)
    for members in sorted({2, min(n_dev, 4), min(n_dev, 8)}):
        g = nh.NerfGroup(list(range(members)))
        g.load_model(desc)
        g.set_resolution(W, H)
        g.render_views(cams, poses)
        assert g.stats().n_composited == want_samples
        for i in range(n):
            rgba, depth = g.read_view_f32(i)
            np.testing.assert_array_equal(rgba, want[i][0])
            np.testing.assert_array_equal(depth, want[i][1])
        # the host end over peer copies (nrf_group_render_host_u8: packed shards by hipMemcpyPeerAsync, one untile, the copy to
        # pinned host memory): the bytes of the single-context render, twice (both host-frame slots)
        for rep in range(2):
            rgb8, d8 = g.render_host_u8(cams, poses)
            for i in range(n):
                np.testing.assert_array_equal(rgb8[i], want_u8[i][0], err_msg=f"{members} members, view {i}, pass {rep}")
                np.testing.assert_array_equal(d8[i], want_u8[i][1])
        g.close()


def _group_case(ctx, n):
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H = 328, 200  # ragged strips (41 tiles per row) and rows (25 tile rows)
    cams = np.stack([syn.default_camera(W, H)] * n)
    poses = np.stack([syn.orbit_pose(70.0 * i, 25.0) for i in range(n)])
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.set_max_views(n)
    ctx.render_views(cams, poses)
    want = [ctx.read_view_f32(i) for i in range(n)]
    want_u8 = [ctx.read_view_u8(i) for i in range(n)]
    want_samples = ctx.stats().n_composited
    ctx.set_max_views(1)
    return desc, keep, W, H, cams, poses, want, want_u8, want_samples


def _check_group(g, n, cams, poses, want, want_u8, want_samples, what):
    f = g.render_views(cams, poses)
    assert f.n_views == n and f.tile_major == 0
    assert g.stats().n_composited == want_samples
    for i in range(n):
        rgba, depth = g.read_view_f32(i)
        np.testing.assert_array_equal(rgba, want[i][0], err_msg=what)
        np.testing.assert_array_equal(depth, want[i][1], err_msg=what)
        rgb8, d8 = g.read_view_u8(i)
        np.testing.assert_array_equal(rgb8, want_u8[i][0], err_msg=what)
    for rep in range(3):  # both host-frame slots, and the first one again
        rgb8, d8 = g.render_host_u8(cams, poses)
        for i in range(n):
            np.testing.assert_array_equal(rgb8[i], want_u8[i][0], err_msg=f"{what}, host frame {i}, pass {rep}")
            np.testing.assert_array_equal(d8[i], want_u8[i][1])


def test_device_group_rccl_gather_one_member(ctx):
    """nrf_group_set_gather(NRF_GATHER_RCCL) (csrc/nrf_group.hip ship_to_first): ncclCommInitAll + one group of
    ncclSend / ncclRecv to devices[0] per call in place of hipMemcpyPeerAsync (the reference: cudaMemcpyAsync D2H per GPU,
    R/src/nerf_render.cu:345-359).  With ONE member -- all this box has -- the group still runs the whole exchange in this
    mode: the member renders its shard tile-major (nrf_options.tile_major), RCCL moves it (a send to self), devices[0]
    untiles.  Float planes, u8 images, host frames and sample counts must be a single context's, bit for bit; switching
    the transport back and forth between calls as well; a group that lists a device twice is refused."""
    n = 4
    desc, keep, W, H, cams, poses, want, want_u8, want_samples = _group_case(ctx, n)
    g = nh.NerfGroup([0])
    g.load_model(desc)
    g.set_resolution(W, H)
    assert g.gather() == (nh.GATHER_PEER_COPY, 0)
    _check_group(g, n, cams, poses, want, want_u8, want_samples, "peer copies (a lone member: no exchange)")
    g.set_gather(nh.GATHER_RCCL)
    mode, version = g.gather()
    assert mode == nh.GATHER_RCCL and version >= 20000, (mode, version)
    _check_group(g, n, cams, poses, want, want_u8, want_samples, "RCCL")
    g.set_gather(nh.GATHER_PEER_COPY)
    _check_group(g, n, cams, poses, want, want_u8, want_samples, "peer copies again")
    g.set_gather(nh.GATHER_RCCL)  # (a second communicator in one process)
    g.set_resolution(W, H)
    _check_group(g, n, cams, poses, want, want_u8, want_samples, "RCCL again")
    # the switch reallocates the host-frame slots: refused while a ticket has not been waited for
    ticket = g.submit_host_u8(cams, poses)
    with pytest.raises(nh.NerfHipError, match="outstanding"):
        g.set_gather(nh.GATHER_PEER_COPY)
    assert g.gather()[0] == nh.GATHER_RCCL
    rgb8, d8 = g.wait_host_u8(ticket)
    for i in range(n):
        np.testing.assert_array_equal(rgb8[i], want_u8[i][0])
    g.set_gather(nh.GATHER_PEER_COPY)
    _check_group(g, n, cams, poses, want, want_u8, want_samples, "peer copies after a waited ticket")
    g.close()
    g2 = nh.NerfGroup([0, 0])
    with pytest.raises(nh.NerfHipError, match="DISTINCT"):
        g2.set_gather(nh.GATHER_RCCL)
    g2.load_model(desc)
    g2.set_resolution(W, H)
    _check_group(g2, n, cams, poses, want, want_u8, want_samples, "two members on one device stay on peer copies")
    g2.close()


def test_device_group_rccl_gather_on_distinct_devices(ctx):
    """The RCCL transport across DIFFERENT devices (xGMI): runs wherever at least two GPUs are visible (the driver's 8-GPU
    node); skipped on a one-GPU box."""
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs at least two visible GPUs")
    n = 3
    desc, keep, W, H, cams, poses, want, want_u8, want_samples = _group_case(ctx, n)
    for members in sorted({2, min(n_dev, 4), min(n_dev, 8)}):
        g = nh.NerfGroup(list(range(members)))
        g.load_model(desc)
        g.set_resolution(W, H)
        g.set_gather(nh.GATHER_RCCL)
        _check_group(g, n, cams, poses, want, want_u8, want_samples, f"RCCL, {members} members")
        g.close()


def test_single_shard_in_the_shard_layout(ctx):
    """nrf_options.tile_major: one shard (shard_count 1) rendered tile-major + nrf_untile_views with shard_count 1 == the
    row-major frame, float planes and packed pixels (what bench.py --force-dist and a one-member RCCL group render)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H, n = 328, 200, 3
    cams = np.stack([syn.default_camera(W, H)] * n)
    poses = np.stack([syn.orbit_pose(100.0 * i, 15.0) for i in range(n)])
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.set_max_views(n)
    ctx.render_views(cams, poses)
    want = [ctx.read_view_f32(i) for i in range(n)]
    want_u8 = [ctx.read_view_u8(i) for i in range(n)]
    o = nh.default_options()
    o.tile_major = 1
    ctx.set_options(o)
    tps = nh.tiles_per_shard(W, H, 1)
    f = ctx.render_views(cams, poses)
    assert f.tile_major == 1 and f.n_tiles == tps and f.view_stride_px == tps * 64
    with pytest.raises(nh.NerfHipError):
        ctx.read_view_f32(0)  # a shard layout is not a row-major frame
    out = torch.empty((n, H, W, 4), device="cuda")
    outd = torch.empty((n, H, W), device="cuda")
    ctx.untile_views(f.rgba, 1, tps, 4, n, out.data_ptr())
    ctx.untile_views(f.depth, 1, tps, 1, n, outd.data_ptr())
    sync()
    for i in range(n):
        np.testing.assert_array_equal(out[i].cpu().numpy(), want[i][0])
        np.testing.assert_array_equal(outd[i].cpu().numpy(), want[i][1])
    packed = torch.zeros((n, tps * 64), dtype=torch.int32, device="cuda")
    ctx.bind_output_rgbd8(packed.data_ptr())
    ctx.render_views(cams, poses)
    ctx.bind_output_rgbd8(None)
    rgb8 = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    d8 = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
    ctx.untile_views_u8(packed.data_ptr(), 1, tps, n, rgb8.data_ptr(), d8.data_ptr())
    sync()
    for i in range(n):
        np.testing.assert_array_equal(rgb8[i].cpu().numpy(), want_u8[i][0])
        np.testing.assert_array_equal(d8[i].cpu().numpy(), want_u8[i][1])
    ctx.set_options(nh.default_options())
    ctx.set_max_views(1)


@pytest.mark.parametrize("radius,az,el,fl_scale", [(0.9, 40, 10, 1.0), (0.3, 200, 35, 0.4), (2.2, 310, 80, 3.0), (1.5, 0, -89, 1.0)])
def test_region_of_interest_cull_is_conservative(ctx, small, radius, az, el, fl_scale):
    """The per-view pixel rectangle outside of which strips are filled with the background without
    generating rays (nrf_api.hip view_roi): cameras inside the volume, next to the box of occupied
    cells (corners behind the image plane -> whole image), far away with a long lens (object spans
    the frame), looking straight up -- the frame must still match the oracle everywhere."""
    desc, keep, o = small
    ctx.load_model(desc)
    W, H = 136, 72
    cam = syn.default_camera(W, H)
    cam[:2] *= np.float32(fl_scale)
    pose = syn.orbit_pose(az, el, radius=radius / 0.33)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and np.abs(depth - wdepth).max() <= 2.0 / 255.0
    # pixels the oracle did not touch are exactly the background in both
    untouched = want[..., 3] == 0
    assert np.array_equal(rgba[untouched], want[untouched])


@pytest.mark.timeout(180)
def test_unusual_inputs_terminate_and_stay_finite(ctx):
    """NaN / inf poses, a zero focal length, 1x1 and 7x5 frames, the camera at the centre of the volume looking
    along an axis (zero direction components -> infinite reciprocals): every launch ends, every pixel is finite,
    and degenerate cameras yield the background."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())

    def go(W, H, cam, pose):
        ctx.set_resolution(W, H)
        ctx.render(cam, pose)
        rgba, depth = ctx.read_f32()
        assert np.isfinite(rgba).all() and np.isfinite(depth).all()
        return rgba, ctx.stats().n_samples

    cam, pose = syn.default_camera(64, 48), syn.orbit_pose(30, 30)
    for bad in ((0, 3, np.nan), (1, 1, np.inf), (2, 3, -np.inf)):
        p = pose.copy(); p[bad[0], bad[1]] = bad[2]
        rgba, n = go(64, 48, cam, p)
        assert n == 0 and np.all(rgba[..., :3] == 1.0) and np.all(rgba[..., 3] == 0.0)
    z = cam.copy(); z[0] = 0
    rgba, n = go(64, 48, z, pose)
    assert n == 0
    go(64, 48, cam, np.zeros((4, 4), np.float32))
    assert go(1, 1, syn.default_camera(1, 1), pose)[1] > 0
    assert go(7, 5, syn.default_camera(7, 5), pose)[1] > 0
    rgba, n = go(64, 48, cam, np.eye(4, dtype=np.float32))
    assert n > 0 and rgba[..., 3].max() <= 1.0 + 1e-5
    # a camera thousands of scene sizes away (t + dt == t in fp32 far enough out: the reference's march never ends): beyond
    # 4096 ngp units a view is background.  Tested where the march would still advance first, then where it would not.
    for radius in (2000.0 / 0.33, 5000.0 / 0.33, 3.0e6, 1.0e12):
        rgba, n = go(64, 48, cam, syn.orbit_pose(30, 30, radius=radius))
        if radius > 4500.0 / 0.33:
            assert n == 0 and np.all(rgba[..., :3] == 1.0) and np.all(rgba[..., 3] == 0.0), radius


@pytest.mark.parametrize("H,bound,cascade,dt_gamma", [(30, 1.0, 1, 1.0 / 128), (96, 1.0, 1, 1.0 / 128), (64, 1.5, 2, 0.0),
                                                      (48, 3.0, 3, 1.0 / 64), (128, 0.75, 1, 1.0 / 128)])
def test_every_march_instance_matches_oracle(ctx, H, bound, cascade, dt_gamma):
    """The render kernel picks its march instance from the model: tables in LDS or not (H % 4), the specialised
    single-cascade / power-of-two instance or the generic one, one visibility walk per cascade, bounds that are
    not powers of two (mip_bound = min(2^k, bound)), bound < 1, dt_gamma = 0 (constant step).  Frames must match
    the oracle and the stage march must be bit-exact for each."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=14, H=H, bound=bound, cascade=cascade)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, Hh = 88, 56
    cam, pose = syn.default_camera(W, Hh), syn.orbit_pose(140, 20)
    opts = nh.default_options()
    opts.dt_gamma = dt_gamma
    ro, rd, nr, fr = _rays(ctx, o, W, Hh, cam, pose)
    n = W * Hh
    xyzs = torch.empty((n, 4, 3), device="cuda"); dirs = torch.empty((n, 4, 3), device="cuda")
    deltas = torch.empty((n, 4, 2), device="cuda")
    ctx.set_options(opts)
    sync()
    ctx.march(ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr(), n, 4, xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr())
    wx, wd, wdl = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), 4, opts)
    np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
    np.testing.assert_array_equal(deltas.cpu().numpy(), wdl)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, Hh, cam, pose, opts)
    assert st.n_samples > 0
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and np.abs(depth - wdepth).max() <= 2.0 / 255.0
    ctx.set_options(nh.default_options())


@pytest.mark.parametrize("kw", [dict(dir_otype="Frequency", n_frequencies=2), dict(dir_otype="Identity"), dict(sh_degree=2)])
def test_render_with_other_direction_encodings(ctx, kw):
    """The fused kernel with the other direction encodings of the Composite block (frequency.h:46-93, Identity,
    lower SH degrees with their leading-ones padding): frames against the oracle."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=14, H=64, **kw)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 96, 64
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(215, 25)
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert st.n_samples > 0
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0


@pytest.mark.parametrize("kw", [dict(grid_type="Tiled", base_resolution=16), dict(grid_type="Dense", per_level_scale=1.13),
                                dict(grid_type="Hash", log2_hashmap_size=10, base_resolution=4)])
def test_other_grid_types_encode_and_render(ctx, kw):
    """GridEncoding `type` Tiled (every level folded into base^3 entries by `index % size`) and Dense (all levels
    dense), and a hash table so small that even the coarse levels are hashed: the generic level path of the kernels
    (stride loop + modulo, grid.h:100-117) -- bit-exact encoding and frames against the oracle."""
    desc, keep, cfg = models.build_model(H=32, **kw)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    rng = np.random.default_rng(11)
    pos = np.concatenate([rng.random((4000, 3), dtype=np.float32),
                          np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.999999, 1e-7, 0.5]], np.float32)])
    want = o.encode_grid(pos)
    out = torch.empty((len(pos), 32), dtype=torch.int16, device="cuda")
    p_d = dev(pos)
    sync()
    ctx.encode_grid(p_d.data_ptr(), len(pos), out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), want)
    W, H = 80, 56
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(20, 35)
    rgba, depth, st, wantf, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
    assert st.n_samples > 0
    assert np.abs(rgba - wantf).max() <= 2.0 / 255.0 and np.abs(depth - wdepth).max() <= 2.0 / 255.0


@pytest.mark.parametrize("bound,cascade,aabb_half", [(2.0, 1, 2.0), (1.0, 1, 1.5), (4.0, 2, 4.0)])
def test_occupied_boundary_layer_outside_the_outermost_cube(ctx, bound, cascade, aabb_half):
    """The march clamps positions to +-bound and then the cell index to [0, H-1] (render_utils.h:595-611), so with
    bound > 2^(cascade-1) -- or an aabb wider than +-bound -- positions OUTSIDE the outermost cube are looked up in
    its boundary cells.  With that layer occupied the reference emits samples out there; the kernel's box of
    occupied cells, region of interest and visibility walk must not cut them."""
    Hg = 32
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=Hg, bound=bound, cascade=cascade)
    grid = keep[1].reshape(cascade, Hg, Hg, Hg).copy()
    top = grid[cascade - 1]
    top[Hg - 1, 8:24, 8:24] = 1.0   # +x face of the outermost cascade
    top[10:20, 0, 10:20] = 1.0      # -y face
    top[12:18, 12:18, Hg - 1] = 1.0  # +z face
    cfg = dict(cfg)
    cfg["snapshot"] = dict(cfg["snapshot"], aabb=[-aabb_half] * 3 + [aabb_half] * 3, mean_density=float(grid.mean()))
    d2, k2 = nh.desc_from_config(cfg, keep[0], grid.reshape(-1))
    ctx.load_model(d2)
    o = op.Oracle(d2)
    W, H = 96, 64
    cam = syn.default_camera(W, H) * np.float32(0.5)  # wide field of view: the faces are in the picture
    cam[2:] = (W * 0.5, H * 0.5)
    o_plain = op.Oracle(desc)  # the same scene without the occupied faces
    extra = []
    for az, el, radius in ((20, 25, 4.0311), (200, -30, 9.0), (95, 60, 2.0)):
        pose = syn.orbit_pose(az, el, radius=radius)
        ro, rd, nr, fr = _rays(ctx, o, W, H, cam, pose)
        n = W * H
        xyzs = torch.empty((n, 4, 3), device="cuda"); dirs = torch.empty((n, 4, 3), device="cuda")
        deltas = torch.empty((n, 4, 2), device="cuda")
        sync()
        ctx.march(ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr(), n, 4, xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr())
        wx, wd, wdl = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), 4)
        np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
        rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose)
        assert st.n_samples >= wst.n_samples * 0.995 - 8, (st.n_samples, wst.n_samples)  # nothing was culled away
        extra.append(wst.n_samples - o_plain.render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)[2].n_samples)
        assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
        assert np.abs(depth - wdepth).max() <= 2.0 / 255.0
    # the scene of this test: the occupied faces (and, through the index clamp, the space beyond them) add samples
    assert max(extra) > 1000, extra


def test_fast_interp_is_opt_in_and_within_its_stated_tolerance(small):
    """nrf_options::fast_interp (opt-in): the hash-grid interpolation accumulates (half)(w * h + acc) with ONE rounding per
    corner (v_fma_mixlo/hi_f16) instead of the reference's three.  Stated tolerance: every feature within 4 x 2^-11 (absolute;
    the partial sums are of magnitude 0.5) of the bit-exact path, 99.9 % within 2^-11, about half identical; frames within the
    2/255 every parity test allows, PSNR against the oracle
    still >= 45 dB.  The default (fast_interp = 0) stays bit-exact: checked by every other test of this file."""
    desc, keep, o = small
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    rng = np.random.default_rng(11)
    n = 200_000
    pos = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
    exact = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    fast = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_grid(pos.data_ptr(), n, exact.data_ptr())
    opts = nh.default_options()
    assert opts.fast_interp == 0
    opts.fast_interp = 1
    ctx.set_options(opts)
    ctx.encode_grid(pos.data_ptr(), n, fast.data_ptr())
    a = exact.cpu().numpy().view(np.float16).astype(np.float32)
    b = fast.cpu().numpy().view(np.float16).astype(np.float32)
    np.testing.assert_array_equal(exact.cpu().numpy().view(np.uint16), o.encode_grid(pos.cpu().numpy()).view(np.uint16))  # default: bit-exact
    # The eight partial sums of a level are of the table's magnitude (|entries| <= 0.5 here) whatever the final value is:
    # a step rounds differently by at most one ulp AT THAT MAGNITUDE (2^-11 for sums in [0.5, 1), 2^-12 below), so the
    # tolerance is absolute -- 4 x 2^-11 -- and most features must not differ at all.
    err = np.abs(a - b)
    assert err.max() <= 4 * 2.0 ** -11, float(err.max())
    assert np.percentile(err, 99.9) <= 2.0 ** -11, float(np.percentile(err, 99.9))
    assert (a == b).mean() > 0.4, float((a == b).mean())  # (measured: 56 % bit-identical)
    assert (a != b).any()  # the option does something
    # frames: persistent register-resident instance
    W, H = 256, 144
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
    ctx.set_resolution(W, H)
    ctx.render(cam, pose)
    f_fast, d_fast = ctx.read_f32()
    opts.fast_interp = 0
    ctx.set_options(opts)
    ctx.render(cam, pose)
    f_exact, d_exact = ctx.read_f32()
    want, wd, _ = o.render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    assert np.abs(f_fast - f_exact).max() <= 2.0 / 255.0 and np.abs(d_fast - d_exact).max() <= 2.0 / 255.0
    assert np.abs(f_fast - want).max() <= 2.0 / 255.0 and models.psnr(f_fast, want) >= 45.0
    assert not np.array_equal(f_fast, f_exact)
    ctx.close()


@pytest.mark.parametrize("W,H", [(3840, 2160), (7680, 4320)])
def test_large_frames_4k_and_8k(W, H):
    """Maximum sizes: one 4K / 8K view of the config-2 model (130 k / 518 k tiles: 32 k / 130 k strips in the work queues -- the 8K
    one is beyond what a planned launch takes, PLAN_CAP).  Size-independent properties: two renders are the same bits, alpha in
    [0, 1], background exactly the background outside the object's projection, the host-frame path returns the bytes of
    nrf_read_u8, two crops (centre, silhouette) equal the oracle's render of the cropped camera, and the composited samples
    scale with the pixel count of the 1080p frame of the same pose."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    c = nh.NerfHip(0)
    c.load_model(desc)
    c.set_resolution(W, H)
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(30, 30)
    c.render(cam, pose)
    rgba, depth = c.read_f32()
    n1 = int(c.stats().n_composited)
    c.render(cam, pose)
    rgba2, depth2 = c.read_f32()
    assert np.array_equal(rgba, rgba2) and np.array_equal(depth, depth2) and int(c.stats().n_composited) == n1
    assert np.all(np.isfinite(rgba)) and rgba[..., 3].min() >= 0.0 and rgba[..., 3].max() <= 1.0 + 1e-6
    assert np.all(rgba[:8, :8, :3] == 1.0) and np.all(rgba[:8, :8, 3] == 0.0) and np.all(depth[:8, :8] == 0.0)  # a corner: background
    assert rgba[H // 2, W // 2, 3] > 0.5  # the object is in the middle
    rgb8, d8 = c.read_u8()
    host_rgb, host_d = c.render_host_u8([cam], [pose])
    np.testing.assert_array_equal(host_rgb[0], rgb8)
    np.testing.assert_array_equal(host_d[0], d8)
    o = op.Oracle(desc)
    cw, ch = 96, 48
    ys, xs = np.nonzero(rgba[..., 3] > 0.5)
    for x0, y0 in ((W // 2 - cw // 2, H // 2 - ch // 2), (max(int(xs.min()) - cw // 2, 0) & ~7, int(ys[np.argmin(xs)]) & ~7)):
        y0 = min(y0, H - ch)
        ccam = cam.copy(); ccam[2] -= x0; ccam[3] -= y0
        want, wdepth, _ = o.render(ccam, pose, cw, ch, schedule=op.SCHED_PER_RAY)
        crop = rgba[y0:y0 + ch, x0:x0 + cw]
        assert np.abs(crop - want).max() <= 2.0 / 255.0 and models.psnr(crop, want) >= 45.0, (x0, y0)
        assert np.abs(depth[y0:y0 + ch, x0:x0 + cw] - wdepth).max() <= 2.0 / 255.0
    c.set_resolution(1920, 1080)
    c.render(syn.default_camera(1920, 1080), pose)
    n_1080 = int(c.stats().n_composited)
    assert abs(n1 / n_1080 - (W * H) / (1920 * 1080)) <= 0.02 * (W * H) / (1920 * 1080)
    c.close()


@pytest.mark.parametrize("log2T", [12, 19])
def test_quad_gather_copies_change_no_bit(log2T):
    """Round 6: the render kernel reads the levels of an F = 2 x 16 grid from cell-major quad copies (two aligned 16-byte gathers
    per level instead of eight 4-byte ones; nrf_device.h level_gather_quad / _far) as far as nrf_model_desc.gather_copy_budget_mb
    allows.  The copies hold the entries grid_index (T/.../grid.h:100-117) names, so NOTHING may change: features against the
    oracle, and features / frames / sample counts between no copies (128 lane addresses per sample), the copies of levels 0..7
    (95 MB: 80 addresses) and those of levels 0..11 (4.6 GB, the last four beyond a buffer resource's reach: 56 addresses)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=log2T, H=32)
    o = op.Oracle(desc)
    rng = np.random.default_rng(11)
    pos = rng.random((20000, 3), dtype=np.float32)
    edge = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 1, 0], [1 - 2 ** -24, 2 ** -24, 0.5], [1, 1, 0], [0, 1, 1]], np.float32)
    pos = np.concatenate([edge, pos])
    # positions outside [0, 1] (a caller's garbage through the stage entry point): no value is promised, no fault may happen
    wild = np.array([[-0.5, 0.5, 0.5], [1.5, 2.0, -3.0], [np.nan, 0.5, 0.5], [np.inf, -np.inf, 0.5], [1e30, 1e30, 1e30]], np.float32)
    want = o.encode_grid(pos)
    W, H = 96, 64
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(40, 25)
    seen = {}
    for budget, addrs in ((1, 128), (256, 80), (0, 56)):
        d = nh.ModelDesc.from_buffer_copy(desc)  # (the pointers stay `keep`'s)
        d.gather_copy_budget_mb = budget
        h = nh.NerfHip(0)
        try:
            h.load_model(d)
            p_d = dev(np.concatenate([pos, wild]))
            out = torch.empty((len(pos) + len(wild), 32), dtype=torch.int16, device="cuda")
            sync()
            h.encode_grid(p_d.data_ptr(), len(pos) + len(wild), out.data_ptr())
            got = out.cpu().numpy().view(np.uint16)[:len(pos)]
            np.testing.assert_array_equal(got, want, err_msg=f"budget {budget} MB")
            h.set_resolution(W, H)
            h.render(cam, pose)
            rgba, depth = h.read_f32()
            st = h.stats()
            assert st.gather_addresses_per_sample == addrs, (budget, st.gather_addresses_per_sample)
            seen[budget] = (rgba.copy(), depth.copy(), int(st.n_composited), int(st.grid_device_bytes))
        finally:
            h.close()
    for budget in (256, 0):
        np.testing.assert_array_equal(seen[budget][0], seen[1][0])
        np.testing.assert_array_equal(seen[budget][1], seen[1][1])
        assert seen[budget][2] == seen[1][2]
    assert seen[1][3] < 40e6 and 90e6 < seen[256][3] - seen[1][3] < 100e6 and 4.4e9 < seen[0][3] - seen[256][3] < 4.6e9


def test_relu_non_finite_hidden_activations_clamp_to_zero_deviation_d10(ctx):
    """DESIGN.md deviation D-10.  tcnn's ReLU is the product x * (half)(x > 0) (T/include/tiny-cuda-nn/common_device.h:71-76): a
    hidden pre-activation that is NaN, or below -65504 (an fp16 -inf), stays / becomes NaN and poisons every output of the sample
    -- the oracle says so (tests/test_oracle_kat.py::test_relu_is_tcnns_product_not_a_max).  The HIP path's ReLU is v_pk_max_f16
    (and `v > 0 ? v : 0` in the generic instance): both cases clamp to 0, the sample's outputs are those of a network whose
    hidden value is 0.  Pinned here so that the behaviour is a decision, not an accident."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    p = keep[0].copy()
    D0 = np.zeros((64, 32), np.float32)
    D0[np.arange(32), np.arange(32)] = 1.0
    D0[1, 1] = -2.0
    p[:64 * 32] = D0.reshape(-1)
    desc2, keep2 = nh.desc_from_config({**cfg}, p, keep[1])
    ctx.load_model(desc2)
    feat = np.full((5, 32), 0.25, np.float16)
    feat[1, 1] = 60000.0           # hidden row 1: -120000 -> fp16 -inf -> (reference: NaN) HIP: 0 ...
    feat[2, 1] = 1.0               # ... like any other negative pre-activation
    feat[3, 5] = np.float16("nan")  # NaN x 0 = NaN in EVERY row of the layer: (reference: all NaN) HIP: every hidden value 0 ...
    feat[4, :] = 0.0               # ... like an all-zero input
    dirf = np.full((5, 16), 0.5, np.float16)
    out = torch.empty((5, 4), dtype=torch.float16, device="cuda")
    f_d, d_d = dev(feat.view(np.uint16).view(np.int16)), dev(dirf.view(np.uint16).view(np.int16))
    sync()
    ctx.mlp_forward(f_d.data_ptr(), d_d.data_ptr(), 5, out.data_ptr())
    got = out.cpu().numpy().view(np.uint16)
    want = op.Oracle(desc2).mlp_forward(feat.view(np.uint16), dirf.view(np.uint16))
    wf = want.view(np.float16).astype(np.float32)
    assert np.all(np.isnan(wf[1])) and np.all(np.isnan(wf[3])) and np.all(np.isfinite(wf[[0, 2, 4]]))  # the reference's semantics
    assert np.all(np.isfinite(got.view(np.float16).astype(np.float32)))
    np.testing.assert_array_equal(got[1], got[2])
    np.testing.assert_array_equal(got[3], got[4])
    mlp_close(got[[0, 2, 4]].view(np.float16).astype(np.float32), wf[[0, 2, 4]], "finite samples")


@pytest.mark.parametrize("kw", [dict(), dict(bound=4.0, cascade=3), dict(rgb_output_activation="Sigmoid", n_neurons=32)])
def test_perturb_branch_march_bit_exact_and_frames(ctx, kw):
    """nrf_options.perturb > 0 -- the perturb branch of kernel_march_rays (R/include/nerf-cuda/render_utils.h:585-589:
    `pcg32 rng(n, perturb); t += MIN_STEPSIZE() * rng.next_float()`, T/dependencies/pcg32/pcg32.h), dead in the reference
    (m_perturb = false, nerf_render.h:75) and the last statement of a SURVEY 8(a) function that was not restated.
    (1) nrf_march against the oracle, BIT-EXACT for n_step 1 / 3 / 8 (n = the ray's place in the call), another seed = other samples;
    (2) whole frames: the kernel applies the branch with the per-ray loop's n_step == 1 and n = the ray's pixel (what the
    reference's first round uses; its later rounds number the rays by an atomicAdd compaction whose order changes from run to
    run) -- against the oracle's per-ray schedule at the frame tolerance, shards + untile bit-identical to the whole frame
    (the ray number is the GLOBAL pixel), and a batch of views identical to single renders."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, **kw)
    ctx.load_model(desc)
    o = op.Oracle(desc)
    W, H = 48, 40
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(45, 25)
    opts = nh.default_options()
    opts.perturb = 5
    ctx.set_options(opts)
    ro, rd, nr, fr = _rays(ctx, o, W, H, cam, pose)
    n = W * H
    seen = {}
    for n_step in (1, 3, 8):
        xyzs = torch.empty((n, n_step, 3), device="cuda"); dirs = torch.empty((n, n_step, 3), device="cuda")
        deltas = torch.empty((n, n_step, 2), device="cuda")
        sync()
        ctx.march(ro.data_ptr(), rd.data_ptr(), nr.data_ptr(), fr.data_ptr(), n, n_step, xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr())
        wx, wd, wdl = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), n_step, opts)
        np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
        np.testing.assert_array_equal(dirs.cpu().numpy(), wd)
        np.testing.assert_array_equal(deltas.cpu().numpy(), wdl)
        assert (wdl[:, :, 0] > 0).sum() > 0
        seen[n_step] = wx
    plain = o.march(ro.cpu().numpy(), rd.cpu().numpy(), nr.cpu().numpy(), fr.cpu().numpy(), 8)[0]
    assert not np.array_equal(plain, seen[8])
    # (2) frames
    rgba, depth, st, want, wdepth, wst = _render_both(ctx, o, W, H, cam, pose, opts)
    assert np.abs(rgba - want).max() <= 2.0 / 255.0 and np.abs(depth - wdepth).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0
    assert abs(int(st.n_composited) - int(wst.n_samples)) <= 3 + int(wst.n_samples) // 1000   # a T ~ 1e-4 tie may fall either way (v_exp_f32)
    off = nh.default_options()
    plain_rgba, _, _, _, _, _ = _render_both(ctx, o, W, H, cam, pose, off)
    assert not np.array_equal(plain_rgba, rgba)  # the branch does something
    for count in (2, 3):
        tps = nh.tiles_per_shard(W, H, count)
        gathered = torch.zeros((count, tps * 64, 4), device="cuda")
        for idx in range(count):
            so = nh.default_options(); so.perturb = 5; so.shard_index, so.shard_count = idx, count
            ctx.set_options(so)
            f = ctx.render(cam, pose)
            n_px = f.n_tiles * 64
            shard = torch.empty((n_px, 4), device="cuda")
            sync()
            _d2d(shard.data_ptr(), f.rgba, n_px * 16)
            gathered[idx, :n_px] = shard
        out = torch.empty((H, W, 4), device="cuda")
        sync()
        ctx.untile(gathered.data_ptr(), count, tps, 4, out.data_ptr())
        np.testing.assert_array_equal(out.cpu().numpy(), rgba)
    ctx.set_options(opts)
    ctx.set_max_views(3)
    poses = np.stack([pose, syn.orbit_pose(200, 10), pose])
    ctx.render_views(np.stack([cam] * 3), poses)
    np.testing.assert_array_equal(ctx.read_view_f32(0)[0], rgba)
    np.testing.assert_array_equal(ctx.read_view_f32(2)[0], rgba)
    bad = nh.default_options(); bad.perturb = -1
    with pytest.raises(nh.NerfHipError, match="perturb"):
        ctx.set_options(bad)
    ctx.set_max_views(1)
    ctx.set_options(nh.default_options())


def test_quad_gather_copies_full_size_frame_identical():
    """BASELINE config 2 at its own size: the 1920x1080 frame of the bench scene with no gather copies (the reference's table
    alone, 128 lane addresses per sample) and with the default budget (levels 0-11 from their cell-major quad copies, 56) --
    float planes and sample counts bit for bit (size-independent property of test_quad_gather_copies_change_no_bit)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128)
    W, H = 1920, 1080
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(45, 30)
    got = {}
    for budget in (1, 0):
        d = nh.ModelDesc.from_buffer_copy(desc)
        d.gather_copy_budget_mb = budget
        h = nh.NerfHip(0)
        try:
            h.load_model(d)
            h.set_resolution(W, H)
            h.render(cam, pose)
            rgba, depth = h.read_f32()
            st = h.stats()
            got[budget] = (rgba, depth, int(st.n_composited), int(st.gather_addresses_per_sample))
        finally:
            h.close()
    assert got[1][3] == 128 and got[0][3] == 56 and got[0][2] == got[1][2] > 5_000_000
    np.testing.assert_array_equal(got[0][0], got[1][0])
    np.testing.assert_array_equal(got[0][1], got[1][1])
