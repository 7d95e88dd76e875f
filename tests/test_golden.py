"""Committed golden fixture (tests/golden/tiny_scene.npz, made by make_golden.py with the oracle):
the oracle must keep reproducing it bit for bit (CPU), the HIP path must match it (GPU)."""
from pathlib import Path

import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn

G = np.load(Path(__file__).parent / "golden" / "tiny_scene.npz")
LOG2T, H, W, HH, SEED = [int(v) for v in G["meta"]]


def _model():
    desc, keep, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    # the fixture stores its own inputs: rebuild the description from them, not from the generator
    desc, keep = nh.desc_from_config(cfg, G["params"], G["density_grid"].astype(np.float32))
    return desc, keep


def test_generator_is_reproducible():
    desc, keep, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    np.testing.assert_array_equal(keep[0], G["params"])
    np.testing.assert_array_equal(keep[1], G["density_grid"].astype(np.float32))


def test_oracle_reproduces_golden():
    desc, keep = _model()
    o = op.Oracle(desc)
    d01 = (np.float32(0.5) * G["dir"] + np.float32(0.5)).astype(np.float32)
    np.testing.assert_array_equal(o.encode_grid(G["pos01"]), G["feat"])
    np.testing.assert_array_equal(o.encode_dir(d01), G["dirf"])
    np.testing.assert_array_equal(o.mlp_forward(G["feat"], G["dirf"]), G["out4"])
    xyz = ((G["pos01"] - np.float32(0.5)) * np.float32(2.0)).astype(np.float32)
    s, c = o.network(xyz, G["dir"])
    np.testing.assert_array_equal(s, G["sigma"])
    np.testing.assert_array_equal(c, G["rgb"])
    ro, rd, nr, fr = o.generate_rays(G["cam"], G["pose"], W, HH)
    np.testing.assert_array_equal(rd, G["rays_d"])
    np.testing.assert_array_equal(nr, G["nears"])
    xyzs, dirs, deltas = o.march(ro, rd, nr, fr, 4)
    np.testing.assert_array_equal(xyzs, G["xyzs"])
    np.testing.assert_array_equal(deltas, G["deltas"])
    rgba, depth, st = o.render(G["cam"], G["pose"], W, HH, schedule=op.SCHED_PER_RAY)
    np.testing.assert_array_equal(rgba, G["rgba"])
    np.testing.assert_array_equal(depth, G["depth"])
    assert st.n_samples == int(G["n_samples"]) > 0


@pytest.mark.gpu
def test_hip_matches_golden():
    torch = pytest.importorskip("torch")
    desc, keep = _model()
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    n = len(G["pos01"])
    pos = torch.from_numpy(G["pos01"]).cuda()
    out = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_grid(pos.data_ptr(), n, out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), G["feat"])  # bit-exact
    ctx.set_resolution(W, HH)
    ctx.render(G["cam"], G["pose"])
    rgba, depth = ctx.read_f32()
    assert np.abs(rgba - G["rgba"]).max() <= 2.0 / 255.0 and models.psnr(rgba, G["rgba"]) >= 45.0
    assert np.abs(depth - G["depth"]).max() <= 2.0 / 255.0
    assert ctx.stats().n_samples >= int(G["n_samples"])
    ctx.close()


@pytest.mark.gpu
def test_frame_fingerprints_unchanged():
    """SHA-1 of the float RGBA / depth planes of 26 frames (config 2: three resolutions x six cameras; config 4
    shape: two resolutions x four cameras, outside, inside and far from the volume), recorded when the
    kernel last changed numerically (tests/golden/frame_hashes.txt, written by scripts/frame_hash.py).  Every
    optimisation that claims to be exact -- culling, lookup skipping, instruction selection -- must leave them
    bit-identical; a deliberate numerical change regenerates the file and says so in its commit."""
    import hashlib

    import models
    import nerfhip as nh
    import synthetic as syn
    torch = pytest.importorskip("torch")

    want = [ln.split() for ln in (Path(__file__).parent / "golden" / "frame_hashes.txt").read_text().splitlines() if ln.strip()]
    desc, keep, _ = models.build_model(log2_hashmap_size=19, H=128)
    desc4, keep4, _ = models.build_model(log2_hashmap_size=19, H=128, cascade=5, bound=16.0)  # BASELINE config 4 shape
    c = nh.NerfHip(0)
    loaded = None
    for row in want:
        config4 = row[0] == "c4"
        # columns: ... hash of rgba, hash of depth, evaluated samples when recorded (history), composited samples (round 4 on)
        n_composited = None
        if config4:
            _, W, H, az, el, radius, h_rgba, h_depth, n_samples, *rest = row
        else:
            (W, H, az, el, h_rgba, h_depth, n_samples, *rest), radius = row, 4.0311
        if rest:
            n_composited = rest[0]
        if loaded != config4:
            c.load_model(desc4 if config4 else desc)
            o = nh.default_options()
            o.max_steps = 1024 if config4 else o.max_steps
            c.set_options(o)
            loaded = config4
        W, H = int(W), int(H)
        c.set_resolution(W, H)
        # into poisoned caller-owned planes: a pixel the kernel leaves unwritten cannot hide behind the previous frame
        t_rgba = torch.full((H, W, 4), 7.0, device="cuda")
        t_depth = torch.full((H, W), 7.0, device="cuda")
        torch.cuda.synchronize()
        c.bind_output(t_rgba.data_ptr(), t_depth.data_ptr())
        c.render(syn.default_camera(W, H), syn.orbit_pose(float(az), float(el), radius=float(radius)))
        c.bind_output(0, 0)
        rgba, depth = t_rgba.cpu().numpy(), t_depth.cpu().numpy()
        # the count of evaluated samples depends on the batching (speculation past a ray's end), the picture does not
        # (recorded before tail splitting existed: a frame rendered alone now hands rays of its last tiles to idle waves, which
        #  queue more samples per ray and round)
        st = c.stats()
        if n_composited is not None:  # the samples that reach a ray's compositing sum: deterministic, recorded, equal
            assert int(st.n_composited) == int(n_composited), (row, st.n_composited)
        # the evaluated ones depend on the batching of rays into rounds: at most a measured margin above the composited ones
        # (measured, profiles/r05/waste_small.txt: <= 5.4 % at 800x800 and above -- a view alone queues its full eight samples per
        #  ray and round there --, <= 2 % at 333x211 and 640x360 and <= 9 % for the config-4 shape at 201x133, which are launches
        #  of fewer tiles than the chip has waves and keep the transmittance-dependent queue)
        assert st.n_composited <= st.n_samples <= (1.10 if W * H >= 800 * 800 else 1.15) * st.n_composited + 64, (row, st.n_samples, st.n_composited)
        assert hashlib.sha1(rgba.tobytes()).hexdigest()[:16] == h_rgba, row
        assert hashlib.sha1(depth.tobytes()).hexdigest()[:16] == h_depth, row
    c.close()


# ---- second fixture: a shape outside base.json (tests/golden/generic_scene.npz, make_golden.py main_generic) ----
GG = np.load(Path(__file__).parent / "golden" / "generic_scene.npz")
GENERIC_KW = dict(dir_otype="Frequency", n_frequencies=4, interpolation="Smoothstep", n_features_per_level=4, n_levels=8,
                  n_neurons=32, density_hidden_layers=2, rgb_hidden_layers=1)


def _generic_model():
    log2t, h, w, hh, seed = [int(v) for v in GG["meta"]]
    desc, keep, cfg = models.build_model(log2_hashmap_size=log2t, H=h, seed=seed, **GENERIC_KW)
    np.testing.assert_array_equal(keep[0], GG["params"])  # the generator is reproducible
    desc, keep = nh.desc_from_config(cfg, GG["params"], GG["density_grid"].astype(np.float32))
    return desc, keep, w, hh


def test_oracle_reproduces_generic_golden():
    desc, keep, w, hh = _generic_model()
    o = op.Oracle(desc)
    assert (o.feat_width, o.dir_width) == (32, 32)
    d01 = (np.float32(0.5) * GG["dir"] + np.float32(0.5)).astype(np.float32)
    np.testing.assert_array_equal(o.encode_grid(GG["pos01"]), GG["feat"])
    np.testing.assert_array_equal(o.encode_dir(d01), GG["dirf"])
    np.testing.assert_array_equal(o.mlp_forward(GG["feat"], GG["dirf"]), GG["out4"])
    rgba, depth, st = o.render(GG["cam"], GG["pose"], w, hh, schedule=op.SCHED_PER_RAY)
    np.testing.assert_array_equal(rgba, GG["rgba"])
    np.testing.assert_array_equal(depth, GG["depth"])
    assert st.n_samples == int(GG["n_samples"]) > 0


@pytest.mark.gpu
def test_hip_generic_instance_matches_generic_golden():
    torch = pytest.importorskip("torch")
    desc, keep, w, hh = _generic_model()
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    n = len(GG["pos01"])
    pos = torch.from_numpy(GG["pos01"]).cuda()
    out = torch.empty((n, 32), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_grid(pos.data_ptr(), n, out.data_ptr())
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), GG["feat"])  # bit-exact (Smoothstep, F = 4)
    ctx.set_resolution(w, hh)
    ctx.render(GG["cam"], GG["pose"])
    rgba, depth = ctx.read_f32()
    assert np.abs(rgba - GG["rgba"]).max() <= 2.0 / 255.0 and models.psnr(rgba, GG["rgba"]) >= 45.0
    assert np.abs(depth - GG["depth"]).max() <= 2.0 / 255.0
    ctx.close()
