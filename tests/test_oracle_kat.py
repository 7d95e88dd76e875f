"""Known-answer tests that pin the CPU oracle.

The reference has no golden vectors for this path ("parity unpinned"); these
answers are derived from its source text, not from running it."""
import math

import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn

import ctypes as _C
C_FLOAT_P = _C.POINTER(_C.c_float)


@pytest.fixture(scope="module")
def base():
    desc, keep, cfg = models.build_model(log2_hashmap_size=19)
    return desc, keep, op.Oracle(desc)


def test_f16_conversion_matches_ieee():
    # every half -> float -> half round-trips, and float -> half equals numpy's RNE conversion
    L = op.lib()
    allh = np.arange(65536, dtype=np.uint16)
    asf = allh.view(np.float16).astype(np.float32)
    for h in (0, 1, 0x03ff, 0x0400, 0x3c00, 0x7bff, 0x7c00, 0x8000, 0xfbff, 0xfc00):
        assert L.nrfo_f16_to_f32(h) == asf[h] or (math.isnan(asf[h]))
        assert L.nrfo_f32_to_f16(float(asf[h])) == h
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-5, 2000), rng.normal(0, 3e4, 2000),
                         [65504.0, 65519.9, 65520.0, 1e9, 5.9604645e-08, 2.98e-08, 2.9802322e-08, 6.1e-05]]).astype(np.float32)
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.nrfo_f32_to_f16(float(x)) for x in xs], np.uint16)
    np.testing.assert_array_equal(got, want)
    got_back = np.array([L.nrfo_f16_to_f32(int(h)) for h in want], np.float32)
    np.testing.assert_array_equal(got_back, want.view(np.float16).astype(np.float32))


def test_param_count_and_level_table(base):
    desc, _, _ = base
    # nerf_network.h:425 hard check: 3072 + 7168 + 12196240
    assert desc.n_params == 12206480 == nh.expected_n_params(desc)
    lt = nh.level_table(desc)
    assert list(lt.resolution) == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    sizes = [lt.offset[i + 1] - lt.offset[i] for i in range(16)]
    assert sizes == [4096, 12168, 29792, 79512, 205384] + [524288] * 11
    assert lt.offset[16] == 6098120
    assert abs(desc.per_level_scale - 1.3819129) < 1e-6


@pytest.mark.parametrize("bound,entries", [(2.0, 6299960), (16.0, 6811592)])
def test_grid_entries_other_bounds(bound, entries):
    cfg = syn.base_config()
    cfg["snapshot"] = {"aabb": [-bound] * 3 + [bound] * 3, "bound": bound, "params": [0.0], "density_grid": [0.0]}
    d, _ = nh.desc_from_config(cfg)
    assert nh.level_table(d).offset[16] == entries


def test_fast_hash_and_dense_index(base):
    _, _, o = base
    L = op.lib()
    assert L.nrfo_fast_hash3(1, 1, 1) == 2922720805
    assert L.nrfo_fast_hash3(3, 5, 7) == 1191511397
    assert L.nrfo_fast_hash3(100, 200, 300) == 3655970992
    assert L.nrfo_fast_hash3(2047, 2047, 2047) == 4281096667
    assert o.grid_index(15, 1, 1, 1) == 339493
    assert o.grid_index(15, 3, 5, 7) == 329061
    assert o.grid_index(15, 100, 200, 300) == 110768
    assert o.grid_index(15, 2047, 2047, 2047) == 285147
    assert o.grid_index(0, 3, 5, 7) == 3 + 5 * 16 + 7 * 256  # level 0 is dense (res 16)
    assert o.grid_index(0, 16, 15, 15) == (16 + 15 * 16 + 15 * 256) % 4096  # +1 corner at x=1 wraps by modulo


def test_sh4_known_answers(base):
    _, _, o = base
    def sh(d):
        d01 = (np.asarray(d, np.float32) * 0.5 + 0.5)[None]
        return o.encode_dir(d01)[0].view(np.float16).astype(np.float32)
    want1 = [0.282095, 0, 0.488603, 0, 0, 0, 0.630783, 0, 0, 0, 0, 0, 0.746353, 0, 0, 0]
    want2 = [0.282095, 0, 0.390882, -0.293162, 0, 0, 0.290160, -0.524423, 0.196659, 0, 0, 0, 0.059708, -0.603300,
             0.416248, -0.127449]
    np.testing.assert_allclose(sh([0, 0, 1]), want1, atol=6e-4)
    np.testing.assert_allclose(sh([0.6, 0, 0.8]), want2, atol=6e-4)


def test_nerf_matrix_to_ngp_main_pose():
    out = np.zeros(16, np.float32)
    pose = np.ascontiguousarray(syn.REFERENCE_MAIN_POSE.reshape(-1))
    fp = lambda a: a.ctypes.data_as(op.C.POINTER(op.C.c_float))
    op.lib().nrfo_nerf_matrix_to_ngp(fp(pose), 0.33, fp(out))
    want = [[0.83003, 0.09497, -0.54957, 0.88025], [0.01385, -0.98860, -0.14991, 0.15165],
            [-0.55754, 0.11682, -0.82189, 1.30924], [0, 0, 0, 1]]
    np.testing.assert_allclose(out.reshape(4, 4), want, atol=1e-5)


def test_step_sizes_and_constant_sigma_ray(base):
    desc, _, o = base
    # dt_min = 2*sqrt(3)/1024, dt_max = 2*bound/H
    rays_o = np.array([[0.0, -0.16, -1.5]], np.float32)   # through the chassis slab along +z
    rays_d = np.array([[0.0, 0.0, 1.0]], np.float32)
    xyzs, dirs, deltas = o.march(rays_o, rays_d, np.array([0.2], np.float32), np.array([2.5], np.float32), 8)
    dts = deltas[0, :, 0]
    assert np.all(dts > 0)
    assert np.all(dts >= np.float32(0.00338291) - 1e-9) and np.all(dts <= 0.015625 + 1e-9)
    np.testing.assert_allclose(dts[0], np.clip(np.float32(1.25) / 128, 0.00338291, 0.015625), rtol=2e-2)
    # samples start in the first occupied cell of the slab (cells are 1/64 wide, marked conservatively)
    assert -0.27 <= xyzs[0, 0, 2] <= -0.23
    # telescoping: constant sigma => weight_sum = 1 - exp(-sigma * sum dt)
    sig = np.full((1, 8), 3.0, np.float32)
    rgb = np.full((1, 8, 3), 0.5, np.float32)
    t, st = o.composite(sig, rgb, deltas, np.array([0.2], np.float32), np.zeros((1, 5), np.float32))
    assert abs(st[0, 0] - (1 - math.exp(-3.0 * float(dts.sum())))) < 1e-5
    assert t[0] > 0  # still alive: all 8 samples consumed, T large


def test_composite_termination_rules():
    deltas = np.zeros((3, 4, 2), np.float32)
    deltas[:, :, 0] = 0.01
    deltas[:, :, 1] = 0.01
    deltas[1, 2:, :] = 0  # ray 1 ran out of samples after 2
    sig = np.full((3, 4), 1.0, np.float32)
    sig[2, :] = 5000.0  # ray 2 saturates: T < 1e-4 after the first two samples
    rgb = np.ones((3, 4, 3), np.float32)
    t, st = op.composite(sig, rgb, deltas, np.full(3, 0.5, np.float32), np.zeros((3, 5), np.float32))
    assert t[0] == pytest.approx(0.54) and t[1] == -1 and t[2] == -1
    assert st[2, 0] == pytest.approx(1.0, abs=1e-6)


def test_sigma_exp_saturates_to_fp16_inf():
    # fp16(exp(x)) is +inf for x > 11.0899
    assert op.f16(math.exp(11.08)) < 0x7c00
    assert op.f16(math.exp(11.10)) == 0x7c00


def test_relu_is_tcnns_product_not_a_max():
    """tcnn's ReLU is `x * (T)(x > 0)` in the network's precision (T/include/tiny-cuda-nn/common_device.h:71-76, and
    R/include/nerf-cuda/nerf_network.h:36-37 for the sigma activation): negative -> -0, NaN -> NaN, an fp16 accumulator of -inf
    (any pre-activation below -65504) -> -inf * 0 = NaN, +inf -> +inf.  max(x, 0) -- the HIP path's v_pk_max_f16, DESIGN.md
    deviation D-10 -- gives +0 for the first three."""
    import struct
    relu = lambda v: op.lib().nrfo_activation(nh.ACT["relu"], v)  # noqa: E731
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]  # noqa: E731
    assert bits(relu(-1.5)) == 0x80000000 and bits(relu(-6e-8)) == 0x80000000   # -0 (also for what rounds to the smallest subnormal)
    assert bits(relu(0.0)) == 0 and bits(relu(-0.0)) == 0x80000000                # 0 * 0, -0 * 0
    assert math.isnan(relu(float("nan"))) and math.isnan(relu(float("-inf"))) and math.isnan(relu(-1.0e6))
    assert relu(float("inf")) == float("inf") and relu(1.0e6) == float("inf")    # fp16 overflow of the accumulator
    assert relu(3.0003) == 3.0 and relu(65504.0) == 65504.0                       # the value is an fp16 value
    # a whole network: one hidden pre-activation below -65504 makes every output of the sample NaN (NaN x 0 in the next layers)
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    p = keep[0].copy()
    D0 = np.zeros((64, 32), np.float32)
    D0[np.arange(32), np.arange(32)] = 1.0
    D0[1, 1] = -2.0
    p[:64 * 32] = D0.reshape(-1)
    desc2, keep2 = nh.desc_from_config({**cfg}, p, keep[1])
    feat = np.full((3, 32), 0.25, np.float16)
    feat[1, 1] = 60000.0          # -120000: an fp16 -inf
    feat[2, 1] = np.float16("nan")
    dirf = np.full((3, 16), 0.5, np.float16)
    out = op.Oracle(desc2).mlp_forward(feat.view(np.uint16), dirf.view(np.uint16)).view(np.float16).astype(np.float32)
    assert np.all(np.isfinite(out[0])) and np.all(np.isnan(out[1])) and np.all(np.isnan(out[2]))


def test_perturb_branch_pcg32_and_the_shifted_march(base):
    """The perturb branch of kernel_march_rays (R/include/nerf-cuda/render_utils.h:585-589): `pcg32 rng(n, perturb); t +=
    MIN_STEPSIZE() * rng.next_float()` ahead of a call's loop, `last_t = t` after it.  (1) The generator
    (T/dependencies/pcg32/pcg32.h) against ITS OWN published vector -- pcg32-demo: seed(42, 54) yields 0xa15c02b7, 0x7b47f409,
    ... -- an answer that does not come from this repository.  (2) The shift moves every sample by the same amount when the
    whole stretch is occupied, does not enter deltas[1], and differs per ray number and per seed."""
    import struct
    first = op.lib().nrfo_pcg32_first_float(42, 54)
    want = struct.unpack("<f", struct.pack("<I", (0xa15c02b7 >> 9) | 0x3f800000))[0] - 1.0
    assert first == want and 0.0 <= first < 1.0
    assert op.lib().nrfo_pcg32_first_float(0, 1) != op.lib().nrfo_pcg32_first_float(1, 1) != op.lib().nrfo_pcg32_first_float(1, 2)
    desc, keep, cfg = base
    full = nh.ModelDesc.from_buffer_copy(desc)
    grid = np.full(int(desc.n_density_grid), 1.0, np.float32)  # every cell occupied: the march emits a sample per step
    full.density_grid = grid.ctypes.data_as(C_FLOAT_P)
    o = op.Oracle(full)
    ro = np.tile(np.array([[0.1, 0.05, -0.9]], np.float32), (3, 1))
    rd = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (3, 1))
    t0, far = np.full(3, 0.3, np.float32), np.full(3, 1.5, np.float32)
    plain = o.march(ro, rd, t0, far, 4)
    opts = nh.default_options()
    opts.perturb = 7
    xyz, _, dl = o.march(ro, rd, t0, far, 4, opts)
    dt_min = np.float32(2 * 1.7320508075688772 / 1024)
    for n in range(3):
        shift = np.float32(dt_min * np.float32(op.lib().nrfo_pcg32_first_float(n, 7)))
        t_first = np.float32(t0[n] + shift)
        assert xyz[n, 0, 2] == np.float32(np.float32(-0.9) + t_first)          # z = oz + t * dz with the shifted t
        np.testing.assert_array_equal(dl[n, :, 0], plain[2][n, :, 0])           # dt = clamp(t / 128, dt_min, dt_max) = dt_min here
        assert dl[n, 0, 1] == np.float32(np.float32(t_first + dl[n, 0, 0]) - t_first)  # t - last_t, last_t = the SHIFTED start
    assert len({float(xyz[n, 0, 2]) for n in range(3)}) == 3                     # a different number per ray
    opts.perturb = 8
    assert not np.array_equal(o.march(ro, rd, t0, far, 4, opts)[0], xyz)         # ... and per seed


def test_param_and_grid_size_errors(base):
    desc, keep, _ = base
    import copy, ctypes
    bad = nh.ModelDesc.from_buffer_copy(desc)
    bad.n_params = desc.n_params - 1
    with pytest.raises(op.OracleError) as e:
        op.Oracle(bad)
    assert e.value.code == nh.NRF_E_PARAMS
    bad = nh.ModelDesc.from_buffer_copy(desc)
    bad.cascade = 2
    with pytest.raises(op.OracleError) as e:
        op.Oracle(bad)
    assert e.value.code == nh.NRF_E_PARAMS


def test_reference_and_tile_schedules_agree():
    """The image does not depend on how rays are batched (per-ray sequential compositing)."""
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    o = op.Oracle(desc)
    W, H = 40, 24  # not a multiple of 8 on purpose: ragged tiles
    cam, pose = syn.default_camera(W, H), syn.orbit_pose(70, 20)
    a, da, sa = o.render(cam, pose, W, H, schedule=op.SCHED_REFERENCE)
    b, db, sb = o.render(cam, pose, W, H, schedule=op.SCHED_TILE64)
    c, dc, sc = o.render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    np.testing.assert_allclose(a, b, atol=2e-6)
    np.testing.assert_allclose(da, db, atol=2e-6)
    np.testing.assert_allclose(a, c, atol=2e-6)
    np.testing.assert_allclose(da, dc, atol=2e-6)
    assert sa.n_samples > 0 and sb.n_samples > 0
    assert sc.n_samples <= sa.n_samples  # one sample at a time never evaluates past a ray's end
    # rays that miss the aabb: background, alpha 0, depth 0 (deviation D-3)
    o2, d2, nr, fr = o.generate_rays(cam, pose, W, H)
    miss = (nr >= fr).reshape(H, W)
    assert np.all(a[miss][:, 3] == 0) and np.all(a[miss][:, :3] == 1.0) and np.all(da[miss] == 0)


def test_quantize_u8_saturates():
    rgba = np.array([[-0.5, 0.0, 0.5, 1.0], [1.0, 1.5, np.nan, 0.0], [0.999, 254.9 / 255, 1e9, 0]], np.float32)
    depth = np.array([0.25, 2.0, -1.0], np.float32)
    rgb8, d8 = op.quantize_u8(rgba, depth)
    assert rgb8.tolist() == [[0, 0, 127], [255, 255, 0], [254, 254, 255]]
    assert d8.tolist() == [63, 255, 0]


def test_fp16_instructions_equal_the_software_definition():
    """oracle/Makefile builds with -mf16c where the host has it (one instruction per conversion: the timed CPU baseline is
    then not a software-float emulator).  The instruction forms against the bit-level software definition: all 2^16 halves
    and every third of the 2^32 floats (NaN payloads included)."""
    assert op.fp16_backend() in ("f16c", "software")
    assert op.lib().nrfo_fp16_selfcheck(3) == 0
    for x in (0.0, -0.0, 1.0, 65504.0, 65520.0, 2.0 ** -24, 2.0 ** -25, 3.0e-8, float("inf"), -1.5):
        assert op.lib().nrfo_f32_to_f16(x) == op.lib().nrfo_f32_to_f16_soft(x)


def test_base_config_is_the_references_base_json():
    """The four network blocks the path reads (R/src/nerf_render.cu:113-117) as R/configs/nerf/base.json states them
    (encoding :23-29, network :30-36, dir_encoding :37-51, rgb_network :52-58): the model every benchmark and golden frame of
    this repository is built on is the reference's own configuration, value for value (data restated from that file)."""
    want = {
        "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1},
        "dir_encoding": {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4},
                                                         {"otype": "Identity", "n_bins": 4, "degree": 4}]},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }
    cfg = syn.base_config()
    for block, values in want.items():
        assert cfg[block] == values, block


def test_sh_table_is_the_real_spherical_harmonics_basis():
    """An independent pin of kernel_sh's 64-entry polynomial table (T/.../spherical_harmonics.h:66-152) that does not pass
    through this repository's reading of it: scipy's complex spherical harmonics, folded into the real basis
    (sqrt(2) Re / Im of Y_l^|m|, Condon-Shortley phase included), must equal every oracle coefficient up to degree 8 at
    500 random directions, to fp16 rounding of values of magnitude <= 1.6."""
    special = pytest.importorskip("scipy.special")
    sph = getattr(special, "sph_harm_y", None)
    desc, keep, _ = models.build_model(log2_hashmap_size=12, H=32, sh_degree=8)
    o = op.Oracle(desc)
    rng = np.random.default_rng(0)
    d = rng.normal(size=(500, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d01 = (0.5 * d + 0.5).astype(np.float32)
    got = o.encode_dir(d01).view(np.float16).astype(np.float64)
    v = d01.astype(np.float64) * 2 - 1
    theta, phi = np.arccos(np.clip(v[:, 2] / np.linalg.norm(v, axis=1), -1, 1)), np.arctan2(v[:, 1], v[:, 0])
    for l in range(8):
        for m in range(-l, l + 1):
            Y = sph(l, abs(m), theta, phi) if sph is not None else special.sph_harm(abs(m), l, phi, theta)
            want = Y.real if m == 0 else np.sqrt(2) * (Y.real if m > 0 else Y.imag)
            assert np.abs(got[:, l * l + l + m] - want).max() <= 1e-3, (l, m)


def test_film_curves_are_the_published_operators():
    """The oracle's tonemap curves (restating R/src/render_buffer.cu:261-318, which folds each operator into a ratio of two
    quadratics) against the operators as published: Narkowicz's ACES fit x (2.51 x + 0.03) / (x (2.43 x + 0.59) + 0.14) on
    0.6 x, Hable's filmic curve ((x (A x + C B) + D E) / (x (A x + B) + D F)) - E / F on 2 x, divided by its value at the
    white point 11.2, and Reinhard's x / (1 + Y) with Rec. 709 luminance."""
    x = np.linspace(0.0, 8.0, 257, dtype=np.float32)
    acc = np.zeros((len(x), 4), np.float32)
    acc[:, 0], acc[:, 1], acc[:, 2], acc[:, 3] = x, 0.5 * x, 2.0 * x, 1.0
    bg = [0, 0, 0, 0]
    X = acc[:, :3].astype(np.float64)
    aces = op.rb_tonemap(acc, 0.0, bg, nh.CS_LINEAR, nh.CS_LINEAR, nh.TM_ACES)[:, :3]
    a = 0.6 * X
    np.testing.assert_allclose(aces, a * (2.51 * a + 0.03) / (a * (2.43 * a + 0.59) + 0.14), rtol=2e-6, atol=1e-7)
    A, B, Cc, D, E, F = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
    hable_f = lambda v: (v * (A * v + Cc * B) + D * E) / (v * (A * v + B) + D * F) - E / F  # noqa: E731
    hable = op.rb_tonemap(acc, 0.0, bg, nh.CS_LINEAR, nh.CS_LINEAR, nh.TM_HABLE)[:, :3]
    np.testing.assert_allclose(hable, hable_f(2.0 * X) / hable_f(11.2), rtol=3e-5, atol=2e-6)
    rein = op.rb_tonemap(acc, 0.0, bg, nh.CS_LINEAR, nh.CS_LINEAR, nh.TM_REINHARD)[:, :3]
    Y = 0.2126 * X[:, 0] + 0.7152 * X[:, 1] + 0.0722 * X[:, 2]
    np.testing.assert_allclose(rein, X / (1.0 + Y)[:, None], rtol=2e-6, atol=1e-7)
