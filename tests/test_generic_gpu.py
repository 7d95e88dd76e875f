"""The GENERIC instance of the fused kernel (nerf-cuda_amd/csrc/nrf_generic.h) against the oracle: every network
shape of the reference's JSON vocabulary outside the base.json shape -- direction encodings at real widths
(Frequency with tcnn's default 12 frequencies, SphericalHarmonics to degree 8), n_neurons 16/32/128, other
hidden-layer counts, n_features_per_level 1/4/8, fewer levels, Nearest / Smoothstep interpolation.

Tolerances as in test_parity_gpu.py: hash-grid and SH encodings BIT-EXACT, Frequency 4e-3 (v_sin_f32 on
arguments up to 2^11 pi against libm's sinf; the reference itself uses __sinf), MLP outputs a few fp16 ulps
(fp32 summation order), frames max |d| <= 2/255 and PSNR >= 45 dB."""
import ctypes as C

import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_parity_gpu import dev, mlp_close, sync  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    h = nh.NerfHip(0)
    yield h
    h.close()


SHAPES = {
    # a10: direction encodings at real widths (rgb input 16 + 80 / 16 + 64 ... columns)
    "freq12": dict(dir_otype="Frequency", n_frequencies=12),
    "freq10": dict(dir_otype="Frequency", n_frequencies=10),
    "freq4": dict(dir_otype="Frequency", n_frequencies=4),
    "sh5": dict(sh_degree=5),
    "sh6": dict(sh_degree=6),
    "sh7": dict(sh_degree=7),
    "sh8": dict(sh_degree=8),
    # MLP shapes (fully_fused_mlp.cu:700-725, 636-687)
    "w32_h2_h3": dict(n_neurons=32, density_hidden_layers=2, rgb_hidden_layers=3),
    "w128_h1_h1": dict(n_neurons=128, density_hidden_layers=1, rgb_hidden_layers=1),
    "w16_h3_h4": dict(n_neurons=16, density_hidden_layers=3, rgb_hidden_layers=4),
    "w64_h2_h2": dict(density_hidden_layers=2),
    # grid shapes (grid.h:1365-1411)
    "F1_L16": dict(n_features_per_level=1),
    "F4_L8": dict(n_features_per_level=4, n_levels=8),
    "F8_L16_w128": dict(n_features_per_level=8, n_neurons=128),
    "F2_L5": dict(n_levels=5),
    "F2_L11_sh8_w32": dict(n_levels=11, sh_degree=8, n_neurons=32),
    "nearest": dict(interpolation="Nearest"),
    "smoothstep_F4": dict(interpolation="Smoothstep", n_features_per_level=4, n_levels=6),
    # activations + explicit density width
    "sigmoid_softplus": dict(activation="Softplus", rgb_output_activation="Sigmoid", sigma_activation="ReLU", density_n_output=1),
    # hidden activations other than ReLU at 64 neurons (T/include/tiny-cuda-nn/common_device.h:68-114): frames come from the
    # register-resident NET_ACT instance (round 6; activations on the fp32 accumulators), the stages from the generic kernels
    "act_squareplus": dict(activation="Squareplus"),
    "act_softplus_h2_h1": dict(activation="Softplus", density_hidden_layers=2, rgb_hidden_layers=1),
    "act_sigmoid": dict(activation="Sigmoid", rgb_output_activation="Sigmoid"),
    "act_none_h1_h3": dict(activation="None", rgb_hidden_layers=3),
    "act_sine": dict(activation="Sine"),  # (stays in the generic instance)
}
ACT_INSTANCE = {"act_squareplus": 3, "act_softplus_h2_h1": 3, "act_sigmoid": 3, "act_none_h1_h3": 3, "act_sine": 1}


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p01 = (np.float32(0.5) * xyz + np.float32(0.5)).astype(np.float32)
    d01 = (np.float32(0.5) * d + np.float32(0.5)).astype(np.float32)
    return xyz, d, p01, d01


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_generic_shape_stage_by_stage_and_frame(ctx, name):
    kw = SHAPES[name]
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, **kw)
    ctx.load_model(desc)
    if name in ACT_INSTANCE:
        assert _instance(ctx) == ACT_INSTANCE[name], name
    o = op.Oracle(desc)
    n = 3001  # ragged: not a multiple of 32
    xyz, d, p01, d01 = _inputs(n, 5)
    p01[:4] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [1 - 2 ** -24, 2 ** -24, 0.5]]
    # ---- encodings
    feat_w = o.encode_grid(p01)
    out = torch.empty((n, o.feat_width), dtype=torch.int16, device="cuda")
    p_d = dev(p01)
    sync()
    ctx.encode_grid(p_d.data_ptr(), n, out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), feat_w)          # bit-exact, zero padding included
    dir_w = o.encode_dir(d01)
    outd = torch.empty((n, o.dir_width), dtype=torch.int16, device="cuda")
    d_d = dev(d01)
    sync()
    ctx.encode_dir(d_d.data_ptr(), n, outd.data_ptr())
    got = outd.cpu().numpy().view(np.uint16)
    if kw.get("dir_otype") == "Frequency":
        np.testing.assert_allclose(got.view(np.float16).astype(np.float32), dir_w.view(np.float16).astype(np.float32), atol=4e-3)
        np.testing.assert_array_equal(got[:, 6 * kw["n_frequencies"]:], dir_w[:, 6 * kw["n_frequencies"]:])  # trailing ones
    else:
        np.testing.assert_array_equal(got, dir_w)
    # ---- both MLPs on the oracle's encodings
    want = o.mlp_forward(feat_w, dir_w).view(np.float16).astype(np.float32)
    out4 = torch.empty((n, 4), dtype=torch.float16, device="cuda")
    f_d, dd_d = dev(feat_w.view(np.int16)), dev(dir_w.view(np.int16))
    sync()
    ctx.mlp_forward(f_d.data_ptr(), dd_d.data_ptr(), n, out4.data_ptr())
    got4 = out4.cpu().numpy().astype(np.float32)
    mlp_close(got4[:, :3], want[:, :3], f"{name}: rgb (mlp_forward)")
    if kw.get("sigma_activation") == "ReLU":
        mlp_close(got4[:, 3], want[:, 3], f"{name}: sigma (mlp_forward)")
    else:
        mlp_close(np.log(got4[:, 3]), np.log(want[:, 3]), f"{name}: log sigma (mlp_forward)")
    # ---- the whole network from raw march output (the render kernel's own code path; Frequency models with up to 80
    # direction values run the WIDE form of the register-resident instance there, the stages above the generic kernels)
    # Frequency inputs differ by the __sinf tolerance before the MLP amplifies them: compared at low octaves only
    if kw.get("dir_otype") != "Frequency" or kw["n_frequencies"] <= 4:
        sig_w, rgb_w = o.network(xyz, d)
        sig = torch.empty(n, dtype=torch.float32, device="cuda")
        rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        x_d, dv = dev(xyz), dev(d)
        sync()
        ctx.network(x_d.data_ptr(), dv.data_ptr(), n, sig.data_ptr(), rgb.data_ptr())
        mlp_close(rgb.cpu().numpy(), rgb_w, f"{name}: rgb (network)")
        if kw.get("sigma_activation") == "ReLU":
            mlp_close(sig.cpu().numpy(), sig_w, f"{name}: sigma (network)")
        else:
            mlp_close(np.log(sig.cpu().numpy()), np.log(sig_w), f"{name}: log sigma (network)")
    # ---- a frame
    W, H = 72, 48
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(215, 25)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    ctx.render(cam, pose)
    rgba, depth = ctx.read_f32()
    st = ctx.stats()
    wantf, wdepth, wst = o.render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    # evaluated samples: a ray queues up to 8 per round, so a few lie behind its terminating one; how many depends on timing
    # since tail splitting (a tiny frame is ALL tail: idle waves take rays every round).  Since round 5 a launch with fewer
    # tiles than the chip has waves keeps the transmittance-dependent queue (nrf_api.hip render_views_impl), and the guard is
    # the measured margin, per instance (profiles/r05/waste_small.txt, these 72x48 frames, five renders each): 0-6 % for every
    # shape but Nearest, whose blocky densities end most rays on their FIRST sample -- queued with seven more while T was
    # still 1: 39 %.  The composited samples are the oracle's own.
    # (act_sigmoid: the synthetic weights behind Sigmoid hidden layers give densities that end most rays on their first sample too)
    margin = 1.50 if name in ("nearest", "act_sigmoid") else 1.15
    assert st.n_composited <= st.n_samples <= margin * st.n_composited + 64, (name, st.n_samples, st.n_composited)
    assert abs(int(st.n_composited) - int(wst.n_composited)) <= 0.01 * wst.n_composited + 16
    assert np.abs(rgba - wantf).max() <= 2.0 / 255.0 and models.psnr(rgba, wantf) >= 45.0, name
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0


def test_generic_instance_full_size_and_batches(ctx):
    """Frequency-12 directions on the full-size table (T = 2^19) at 1920x1080: the generic instance's LDS map with the
    march tables, batched views bit-identical to single renders, and a crop against the oracle."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=19, H=128, dir_otype="Frequency", n_frequencies=12)
    ctx.load_model(desc)
    W, H = 1920, 1080
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(30, 30), syn.orbit_pose(200, 15)]
    singles = []
    for p in poses:
        ctx.render(cam, p)
        singles.append(ctx.read_f32())
    assert ctx.stats().n_samples > 5_000_000
    ctx.set_max_views(2)
    ctx.render_views(np.stack([cam, cam]), np.stack(poses))
    for v in range(2):
        rgba, depth = ctx.read_view_f32(v)
        np.testing.assert_array_equal(rgba, singles[v][0])
        np.testing.assert_array_equal(depth, singles[v][1])
    ctx.set_max_views(1)
    assert np.all(np.isfinite(singles[0][0])) and singles[0][0][..., 3].min() >= 0 and singles[0][0][..., 3].max() <= 1 + 1e-5
    # a 128x64 crop through the object against the oracle (same rays: the crop is a shifted principal point)
    cw, ch, x0, y0 = 128, 64, 900, 500
    ccam = cam.copy()
    ccam[2] -= x0
    ccam[3] -= y0
    want, wdepth, _ = op.Oracle(desc).render(ccam, poses[0], cw, ch, schedule=op.SCHED_PER_RAY)
    got = singles[0][0][y0:y0 + ch, x0:x0 + cw]
    assert np.abs(got - want).max() <= 2.0 / 255.0 and models.psnr(got, want) >= 45.0


def test_unsupported_shapes_are_refused_loudly(ctx):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    for field, value, code in (("n_neurons", 48, nh.NRF_E_INVALID), ("n_features_per_level", 3, nh.NRF_E_INVALID),
                               ("sh_degree", 9, nh.NRF_E_INVALID), ("interpolation", 7, nh.NRF_E_INVALID),
                               ("density_n_output", 32, nh.NRF_E_UNSUPPORTED)):
        old = getattr(desc, field)
        setattr(desc, field, value)
        with pytest.raises(nh.NerfHipError) as e:
            ctx.load_model(desc)
        assert e.value.code in (code, nh.NRF_E_PARAMS), (field, e.value)
        setattr(desc, field, old)
    ctx.load_model(desc)


@pytest.mark.parametrize("kw", [dict(), dict(cascade=2, bound=2.0), dict(n_neurons=32, n_features_per_level=4, n_levels=8)])
def test_density_grid_from_the_network(ctx, kw):
    """f4: NerfRender::generate_density_grid (nerf_render.cu:388-429) completed behind nrf_generate_density_grid.  A
    model loaded WITHOUT a density grid refuses to render until the grid has been evaluated from the network; the
    generated grid matches the oracle's (sigma carries the MLP tolerance, so: values within a few fp16 ulps,
    occupancy bits equal except for cells within that tolerance of the threshold); and with the GPU's grid handed to
    the oracle, frames agree as for any snapshot."""
    Hg = 32
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=Hg, **kw)
    cascade = int(desc.cascade)
    n_cells = cascade * Hg ** 3
    d0, k0 = nh.desc_from_config({**cfg, "snapshot": {k: v for k, v in cfg["snapshot"].items()}}, keep[0])  # no grid
    assert d0.n_density_grid == 0
    ctx.load_model(d0)
    ctx.set_options(nh.default_options())
    W, H = 64, 48
    ctx.set_resolution(W, H)
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(40, 25)
    with pytest.raises(nh.NerfHipError) as e:
        ctx.render(cam, pose)
    assert e.value.code == nh.NRF_E_STATE and "density grid" in str(e.value)
    mean = ctx.generate_density_grid(16, 0.95)
    grid, mean2 = ctx.read_density_grid(n_cells)
    assert mean == mean2 and np.isfinite(grid).all() and grid.min() >= np.float32(1 / 64) * np.float32(0.95) ** 16 * 0.999
    # the oracle's grid: same procedure on the CPU
    want, wmean = op.Oracle(d0).density_grid(n_cells, 16, 0.95)
    floor = float(want.min())
    rel = np.abs(grid - want) / np.maximum(np.abs(want), 1e-12)
    assert rel.max() <= 4 * 2.0 ** -8, rel.max()           # sigma = exp(g0) of an fp16 g0 with the MLP tolerance
    assert (rel > 0).mean() < 0.3                           # ... and most cells are bit-identical
    assert abs(mean - wmean) <= 2e-3 * abs(wmean)
    thresh = min(0.01, wmean)
    near = np.abs(want - thresh) <= 4 * 2.0 ** -8 * thresh
    np.testing.assert_array_equal((grid > min(0.01, mean))[~near], (want > thresh)[~near])
    assert floor < 0.01  # sixteen passes of decay took untouched cells below the threshold
    # render with the generated grid; the oracle gets the same grid as a snapshot would carry it
    ctx.render(cam, pose)
    rgba, depth = ctx.read_f32()
    assert ctx.stats().n_samples > 0
    cfg2 = dict(cfg)
    cfg2["snapshot"] = dict(cfg["snapshot"], mean_density=mean)
    d2, k2 = nh.desc_from_config(cfg2, keep[0], grid)
    wantf, wdepth, wst = op.Oracle(d2).render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    assert np.abs(rgba - wantf).max() <= 2.0 / 255.0 and models.psnr(rgba, wantf) >= 45.0
    assert np.abs(depth - wdepth).max() <= 2.0 / 255.0
    # regenerating replaces the tables in place; a snapshot WITH a grid is untouched by all this
    assert ctx.generate_density_grid(16, 0.95) == mean
    ctx.load_model(desc)
    g3, m3 = ctx.read_density_grid(n_cells)
    np.testing.assert_array_equal(g3, keep[1])


def test_instant_ngp_geometry_runs_in_the_register_resident_instance(ctx):
    """instant-ngp's level geometry at aabb_scale 32 (per_level_scale from 2048 * 32 / 16: the finest level has
    res = 65536) meets grid_index's uint32 stride overflow (grid.h:106-109): stride wraps to 0 after the y term, the
    level is (x + y * res) & (size - 1) without hash and without z.  Oracle and kernel follow the uint32 arithmetic
    literally; the hot instance covers the level (LV_ADD_POW2) and instant-ngp's logistic colours (rgb output Sigmoid),
    so such a model renders at the base shape's speed.  Encoding bit-exact, frames at the usual tolerance."""
    pls = nh.default_per_level_scale(32.0, 16, 16)
    for log2T, H in ((12, 32), (19, 32)):
        desc, keep, cfg = models.build_model(log2_hashmap_size=log2T, H=H, bound=16.0, cascade=5, per_level_scale=pls,
                                             rgb_output_activation="Sigmoid")
        assert nh.level_table(desc).resolution[15] == 65536
        ctx.load_model(desc)
        o = op.Oracle(desc)
        rng = np.random.default_rng(4)
        pos = np.concatenate([rng.random((6000, 3), dtype=np.float32),
                              np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.999999, 1e-7, 0.5], [0.25, 1, 1]], np.float32)])
        want = o.encode_grid(pos)
        out = torch.empty((len(pos), 32), dtype=torch.int16, device="cuda")
        p_d = dev(pos)
        sync()
        ctx.encode_grid(p_d.data_ptr(), len(pos), out.data_ptr())
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), want)
        W, Hh = 96, 64
        cam, pose = syn.default_camera(W, Hh), syn.orbit_pose(60, 25)
        ctx.set_options(nh.default_options())
        ctx.set_resolution(W, Hh)
        ctx.render(cam, pose)
        rgba, depth = ctx.read_f32()
        wantf, wdepth, wst = o.render(cam, pose, W, Hh, schedule=op.SCHED_PER_RAY)
        assert ctx.stats().n_samples > 0
        assert np.abs(rgba - wantf).max() <= 2.0 / 255.0 and models.psnr(rgba, wantf) >= 45.0
        assert rgba[..., :3].max() <= 1.0 + 1e-3  # logistic colours + white background stay in range


def _instance(ctx):
    ctx.lib.nrf_debug_instance.argtypes = [C.c_void_p]
    return ctx.lib.nrf_debug_instance(ctx.h) & 15  # 0 register-resident, 1 generic, 2 wide


def test_large_tables_keep_the_register_resident_instance(ctx):
    """A hash table of 2^22 entries or more meets grid_index's uint32 stride overflow at its finest levels (grid.h:106-114):
    for res in 1626 .. sqrt(T) the stride res^3 wraps below the table size, `hashmap_size < stride` is false and the level
    is indexed ADDITIVELY, (x + y res + z res^2 mod 2^32) & (T - 1), not hashed -- at res 2048 on 2^22 entries the z term
    vanishes as well.  Such levels used to send the whole model to the generic instance (2.2 instead of 11 Gsamples/s at
    1080p); they are LV_ADD_POW2 levels of the register-resident one.  Encoding bit-exact against the oracle (which
    restates the uint32 loop literally), a frame at the usual tolerance."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=22, H=32)
    lt = nh.level_table(desc)
    assert lt.resolution[15] == 2048 and lt.offset[16] - lt.offset[15] == 1 << 22
    ctx.load_model(desc)
    assert _instance(ctx) == 0
    o = op.Oracle(desc)
    rng = np.random.default_rng(9)
    pos = np.concatenate([rng.random((4000, 3), dtype=np.float32),
                          np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.999999, 1e-7, 0.5], [0.25, 1, 1]], np.float32)])
    want = o.encode_grid(pos)
    out = torch.empty((len(pos), 32), dtype=torch.int16, device="cuda")
    p_d = dev(pos)
    sync()
    ctx.encode_grid(p_d.data_ptr(), len(pos), out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), want)
    # the finest level really is additive there: two positions that differ in z only (same x, y cell) read the same entries
    assert o.grid_index(15, 100, 200, 5) == o.grid_index(15, 100, 200, 1900) == (100 + 200 * 2048) % (1 << 22)
    W, Hh = 96, 64
    cam, pose = syn.default_camera(W, Hh), syn.orbit_pose(200, 35)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, Hh)
    ctx.render(cam, pose)
    rgba, depth = ctx.read_f32()
    wantf, wdepth, wst = o.render(cam, pose, W, Hh, schedule=op.SCHED_PER_RAY)
    assert ctx.stats().n_samples > 0
    assert np.abs(rgba - wantf).max() <= 2.0 / 255.0 and models.psnr(rgba, wantf) >= 45.0


def test_tiled_grid_with_power_of_two_table_runs_register_resident(ctx):
    """A Tiled grid never hashes: its levels beyond the dense ones are (x + y res + z res^2) % T -- LV_ADD_POW2 as well."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, grid_type="Tiled")
    ctx.load_model(desc)
    assert _instance(ctx) == 0
    o = op.Oracle(desc)
    pos = np.random.default_rng(3).random((3000, 3), dtype=np.float32)
    out = torch.empty((len(pos), 32), dtype=torch.int16, device="cuda")
    p_d = dev(pos)
    sync()
    ctx.encode_grid(p_d.data_ptr(), len(pos), out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), o.encode_grid(pos))


@pytest.mark.parametrize("kw", [dict(dir_otype="Frequency", n_frequencies=12), dict(n_neurons=32, n_features_per_level=4, n_levels=8)])
def test_wide_and_generic_instances_shard_batch_and_group_like_the_hot_one(ctx, kw):
    """The other kernel instances behind the same launch machinery: batched views, 3-way strip shards + untile and a
    two-member device group must all reproduce the single renders bit for bit (wide: Frequency-12; generic: a 32-neuron
    F = 4 model), at a resolution with ragged strips."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=14, H=64, **kw)
    W, H, n = 200, 104, 3
    cams = np.stack([syn.default_camera(W, H)] * n)
    poses = np.stack([syn.orbit_pose(40.0 + 100.0 * i, 20.0) for i in range(n)])
    ctx.load_model(desc)
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    singles = []
    for i in range(n):
        ctx.render(cams[i], poses[i])
        singles.append(ctx.read_f32())
    assert ctx.stats().n_samples > 0
    ctx.set_max_views(n)
    ctx.render_views(cams, poses)
    for i in range(n):
        rgba, depth = ctx.read_view_f32(i)
        np.testing.assert_array_equal(rgba, singles[i][0])
        np.testing.assert_array_equal(depth, singles[i][1])
    ctx.set_max_views(1)
    # strip shards of view 0, untiled on the host
    world = 3
    tps = nh.tiles_per_shard(W, H, world)
    gathered = np.zeros((world, tps * 64, 4), np.float32)
    for r in range(world):
        o = nh.default_options()
        o.shard_index, o.shard_count = r, world
        ctx.set_options(o)
        ctx.set_resolution(W, H)
        f = ctx.render(cams[0], poses[0])
        part = np.empty((f.n_tiles * 64, 4), np.float32)  # the last shards may hold one strip less than tiles_per_shard
        dpart = np.empty(f.n_tiles * 64, np.float32)
        nh._check(ctx.lib.nrf_read_shard_f32(ctx.h, part.ctypes.data, dpart.ctypes.data))
        shard = np.zeros((tps * 64, 4), np.float32)
        shard[:f.n_tiles * 64] = part
        gathered[r] = shard
    np.testing.assert_array_equal(nh.untile_numpy(gathered, W, H), singles[0][0])
    ctx.set_options(nh.default_options())
    ctx.set_resolution(W, H)
    # the one-process device group (both members on this box's one device)
    g = nh.NerfGroup([0, 0])
    g.load_model(desc)
    g.set_resolution(W, H)
    g.render_views(cams, poses)
    for i in range(n):
        rgba, depth = g.read_view_f32(i)
        np.testing.assert_array_equal(rgba, singles[i][0])
    g.close()


GRID_SHAPES = {  # base.json's MLPs behind another grid: the GRID instances (NET_GRID2 / 4 / 8, instance 5)
    "g4_8": dict(n_features_per_level=4, n_levels=8), "g8_4": dict(n_features_per_level=8, n_levels=4),
    "g4_6s": dict(n_features_per_level=4, n_levels=6, interpolation="Smoothstep"), "g8_2": dict(n_features_per_level=8, n_levels=2),
    "g4_3s": dict(n_features_per_level=4, n_levels=3, interpolation="Smoothstep"), "g2_5": dict(n_levels=5), "g2_11": dict(n_levels=11),
    "g2_16s": dict(interpolation="Smoothstep"), "g2_8": dict(n_levels=8), "g4_8_sig": dict(n_features_per_level=4, n_levels=8, rgb_output_activation="Sigmoid"),
    # round 5: Nearest interpolation (grid.h:215-232: the entry at floor(pos), one gather per level) in the GRID instances --
    # the base 16 x 2 grid, and F = 4 / 8
    "g2_16n": dict(interpolation="Nearest"), "g4_8n": dict(n_features_per_level=4, n_levels=8, interpolation="Nearest"),
    "g8_3n": dict(n_features_per_level=8, n_levels=3, interpolation="Nearest"), "g2_7n": dict(n_levels=7, interpolation="Nearest"),
    # round 5: ONE feature per level (NET_GRID1: 2-byte gathers, two levels to a dword) -- Linear, Smoothstep, Nearest
    "g1_16": dict(n_features_per_level=1), "g1_9s": dict(n_features_per_level=1, n_levels=9, interpolation="Smoothstep"),
    "g1_13n": dict(n_features_per_level=1, n_levels=13, interpolation="Nearest"), "g1_3": dict(n_features_per_level=1, n_levels=3),
}


@pytest.mark.parametrize("shape", ["w16", "w32", "w128", "sh5", "sh6", "sh7", "sh8", "d2_2", "d1_1", "d3_4", "d1_3", "d2_1"] + sorted(GRID_SHAPES))
def test_other_widths_and_sh_degrees_render_in_a_register_resident_instance(shape):
    """tcnn's FullyFusedMLP takes 16 / 32 / 64 / 128 neurons (T/src/fully_fused_mlp.cu:700-725).  In the base.json shape the
    other three widths have register-resident instances of the persistent kernel too (NET_W16 / NET_W32 / NET_W128: the MFMA
    chain over MlpShape<W> fragments); stage entry points and the per-strip kernel stay generic.  Frames: against the oracle
    at the MLP tolerance, against the generic instance of the same model (NRF_WIDTH_INSTANCES=0) likewise (the two sum in
    different K orders), batches and host frames bit-identical to single renders, sharded too.
    SphericalHarmonics of degree 5..8 (32..64 padded direction values) likewise keep the register-resident MLPs: NET_WIDE_SH,
    the wide form with every coefficient of a ray computed once into an LDS row (instance 4)."""
    import os

    # "d<a>_<b>": 64 neurons with a / b hidden layers in the density / rgb MLP (base.json: 1 / 2) -- the DEPTH instance
    # (mlp_tiles_depth: a runtime number of 64 -> 64 layers, the same in-lane chaining), reported as a width instance too
    # "g<F>_<L>[s|n]": base.json's MLPs behind a grid of F features x L levels (s: Smoothstep, n: Nearest) -- the GRID instances (round 4):
    # F = 2 with fewer than 16 levels, F = 4 / 8 as 8- / 16-byte gathers; additionally the encode entry point (which runs the
    # instance's own gathers for such a model) is compared with the oracle BIT FOR BIT
    if shape[0] == "g":
        kw = GRID_SHAPES[shape]
    elif shape[0] == "d":
        kw = dict(density_hidden_layers=int(shape[1]), rgb_hidden_layers=int(shape[3]))
    else:
        kw = dict(n_neurons=int(shape[1:])) if shape[0] == "w" else dict(sh_degree=int(shape[2:]))
    width = shape
    want_instance = 5 if shape[0] == "g" else (3 if shape[0] in "wd" else 4)
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, **kw)
    o = op.Oracle(desc)
    if shape[0] == "g":
        c = nh.NerfHip(0)
        c.load_model(desc)
        rng = np.random.default_rng(17)
        p01 = np.concatenate([rng.random((40000, 3), dtype=np.float32), np.array([[0, 0, 0], [1, 1, 1], [1, 0, 0.25], [0.5, 1, 1]], np.float32)])
        want_rows = o.encode_grid(p01)
        out_rows = torch.empty((len(p01), o.feat_width), dtype=torch.int16, device="cuda")
        pd = dev(p01)
        sync()
        c.encode_grid(pd.data_ptr(), len(p01), out_rows.data_ptr())
        np.testing.assert_array_equal(out_rows.cpu().numpy().view(np.uint16), want_rows)
        c.close()
    W, H = 120, 88
    cam = syn.default_camera(W, H)
    poses = [syn.orbit_pose(215, 25), syn.orbit_pose(40, -10), syn.orbit_pose(120, 60)]
    frames = {}
    for env in ("1", "0"):
        os.environ["NRF_WIDTH_INSTANCES"] = env
        try:
            c = nh.NerfHip(0)
        finally:
            os.environ.pop("NRF_WIDTH_INSTANCES", None)
        c.load_model(desc)
        assert _instance(c) == (want_instance if env == "1" else 1)
        c.set_resolution(W, H)
        c.set_max_views(3)
        out = []
        for p in poses:
            c.render(cam, p)
            out.append(c.read_f32())
        frames[env] = out
        if env == "1":
            c.render_views([cam] * 3, poses)
            for v in range(3):
                a, b = c.read_view_f32(v)
                np.testing.assert_array_equal(a, out[v][0])
                np.testing.assert_array_equal(b, out[v][1])
            rgb, d8 = c.render_host_u8([cam] * 3, poses)
            for v in range(3):
                c.render(cam, poses[v])
                r8, dd8 = c.read_u8()
                np.testing.assert_array_equal(rgb[v], r8)
                np.testing.assert_array_equal(d8[v], dd8)
            # two shards + untile reproduce the frame
            import torch as _t
            tps = nh.tiles_per_shard(W, H, 2)
            gathered = _t.zeros((2, tps * 64, 4), device="cuda")
            for idx in range(2):
                opts = nh.default_options(); opts.shard_index, opts.shard_count = idx, 2
                c.set_options(opts)
                f = c.render(cam, poses[0])
                sh = np.zeros((tps * 64, 4), np.float32)
                got_rgba = np.empty((f.n_tiles * 64, 4), np.float32); got_depth = np.empty((f.n_tiles * 64,), np.float32)
                nh._check(c.lib.nrf_read_shard_f32(c.h, got_rgba.ctypes.data, got_depth.ctypes.data))
                sh[:f.n_tiles * 64] = got_rgba
                gathered[idx] = _t.from_numpy(sh).cuda()
            c.set_options(nh.default_options())
            np.testing.assert_array_equal(nh.untile_numpy(gathered.cpu().numpy(), W, H), out[0][0])
        c.close()
    for v, p in enumerate(poses):
        want, wdepth, _ = o.render(cam, p, W, H, schedule=op.SCHED_PER_RAY)
        for env in ("1", "0"):
            rgba, depth = frames[env][v]
            assert np.abs(rgba - want).max() <= 2.0 / 255.0 and models.psnr(rgba, want) >= 45.0, (width, env, v)
            assert np.abs(depth - wdepth).max() <= 2.0 / 255.0
        assert np.abs(frames["1"][v][0] - frames["0"][v][0]).max() <= 2.0 / 255.0
    assert frames["1"][0][0][..., 3].max() > 0.5  # the object is in view


def test_wide_instance_with_the_generic_march():
    """NET_WIDE (Frequency directions of 32-80 values) on a grid whose bound is no power of two: the generic march form, which
    runs 8-wave workgroups since round 5 (at 12 it spilled registers; nrf_launch.h WIDE_GENERIC_MARCH_WAVES).  The persistent
    form against the per-strip kernel bit for bit, and against the oracle at the frame tolerance."""
    import os
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32, bound=3.0, cascade=2, dir_otype="Frequency", n_frequencies=4)
    o = op.Oracle(desc)
    W, H = 200, 136
    cam, poses = syn.default_camera(W, H), [syn.orbit_pose(215, 25), syn.orbit_pose(70, -15)]
    frames = {}
    for env in ("1", "0"):
        os.environ["NRF_PERSISTENT"] = env
        try:
            c = nh.NerfHip(0)
        finally:
            os.environ.pop("NRF_PERSISTENT", None)
        c.load_model(desc)
        c.lib.nrf_debug_instance.argtypes = [C.c_void_p]
        assert c.lib.nrf_debug_instance(c.h) == 2 + (16 if env == "1" else 0)  # the wide instance, persistent or per strip
        c.set_resolution(W, H)
        c.set_max_views(2)
        c.render_views([cam] * 2, poses)
        frames[env] = [c.read_view_f32(v) for v in range(2)]
        c.close()
    for v, p in enumerate(poses):
        np.testing.assert_array_equal(frames["1"][v][0], frames["0"][v][0])
        np.testing.assert_array_equal(frames["1"][v][1], frames["0"][v][1])
        want, wdepth, _ = o.render(cam, p, W, H, schedule=op.SCHED_PER_RAY)
        assert np.abs(frames["1"][v][0] - want).max() <= 2.0 / 255.0 and models.psnr(frames["1"][v][0], want) >= 45.0
