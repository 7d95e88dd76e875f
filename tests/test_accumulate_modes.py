"""How far is the fp32-accumulate arithmetic of this repository (oracle default, HIP path) from the REFERENCE's MLP arithmetic?

The reference accumulates every layer's products in fp16 WMMA fragments (`wmma::fragment<accumulator,16,16,16,__half>`,
T/src/fully_fused_mlp.cu:69 hidden, :334 input, :437 last; one `mma_sync` per 16-wide K block).  The oracle emulates that
accumulator (`nrfo_set_mlp_accumulate`: the running sum is an fp16 value after every block of n products; n = 16 is the
reference's granularity, n = 1 the pessimistic bound).  These tests pin the emulation (fixture made by
tests/golden/make_accumulate_golden.py) and state the tolerance of the HIP path against each mode:

    HIP frame vs fp32-accumulate oracle (the shared contract):   max |d| <= 2/255,  PSNR >= 45 dB   (measured ~106 dB)
    HIP frame vs FP16_K16 (the reference's accumulator):         max |d| <= 1/255,  PSNR >= 72 dB   (measured ~78 dB)
    HIP frame vs FP16_STEP (rounded after every product):        max |d| <= 2/255,  PSNR >= 65 dB   (measured ~71 dB)

north_star's "matches reference PSNR within 0.1 dB": two renderings 72 dB apart change a PSNR against ground truth of
<= 40 dB by < 0.01 dB.  The emulation is still an emulation (tensor-core rounding inside a K block is unspecified, and the
reference cannot be run here): parity stays unpinned."""
from pathlib import Path

import numpy as np
import pytest

import models
import nerfhip as nh
import oracle_py as op
import synthetic as syn

A = np.load(Path(__file__).parent / "golden" / "accumulate_modes.npz")
G = np.load(Path(__file__).parent / "golden" / "tiny_scene.npz")
LOG2T, H, W, HH, SEED = [int(v) for v in G["meta"]]
MODES = [int(m) for m in A["modes"]]
# stated tolerances of a frame against each accumulate mode: (max |d|, PSNR dB)
FRAME_TOL = {op.ACC_FP32: (2.0 / 255.0, 45.0), op.ACC_FP16_K16: (1.0 / 255.0, 72.0), op.ACC_FP16_STEP: (2.0 / 255.0, 65.0)}


def _tiny():
    _, _, cfg = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    return nh.desc_from_config(cfg, G["params"], G["density_grid"].astype(np.float32))


def _f(h):
    return h.view(np.float16).astype(np.float32)


def test_accumulate_mode_is_validated():
    desc, keep = _tiny()
    o = op.Oracle(desc)
    for bad in (-1, 2, 3, 32):
        with pytest.raises(op.OracleError):
            o.set_mlp_accumulate(bad)
    o.set_mlp_accumulate(op.ACC_FP16_K16)
    o.set_mlp_accumulate(op.ACC_FP32)
    np.testing.assert_array_equal(o.mlp_forward(G["feat"], G["dirf"]), G["out4"])  # back to the default arithmetic


def test_oracle_reproduces_accumulate_fixture():
    desc, keep = _tiny()
    for mode in MODES:
        o = op.Oracle(desc, accumulate=mode)
        np.testing.assert_array_equal(o.mlp_forward(G["feat"], G["dirf"]), A[f"tiny_out4_{mode}"])
        rgba, depth, st = o.render(G["cam"], G["pose"], W, HH, schedule=op.SCHED_PER_RAY)
        np.testing.assert_array_equal(rgba, A[f"tiny_rgba_{mode}"])
        np.testing.assert_array_equal(depth, A[f"tiny_depth_{mode}"])
        assert st.n_samples == int(A[f"tiny_n_{mode}"])
    desc2, keep2, _ = models.build_model(log2_hashmap_size=19, H=128)
    for mode in (op.ACC_FP16_K16, op.ACC_FP16_STEP):
        rgba, depth, st = op.Oracle(desc2, accumulate=mode).render(A["c2_cam"], A["c2_pose"], 64, 64, schedule=op.SCHED_PER_RAY)
        np.testing.assert_array_equal(rgba, A[f"c2_rgba_{mode}"])
        np.testing.assert_array_equal(depth, A[f"c2_depth_{mode}"])


def test_modes_are_ordered_by_granularity():
    """Finer rounding granularity -> no smaller a distance from the fp32 sums; every mode stays within a handful of fp16
    ulps of the pre-activations (rgb outputs are sigmoids in [0, 1])."""
    desc, keep = _tiny()
    o = op.Oracle(desc)
    rng = np.random.default_rng(5)
    feat = rng.uniform(-1, 1, (512, o.feat_width)).astype(np.float16)
    dirf = rng.uniform(-1, 1, (512, o.dir_width)).astype(np.float16)
    base = _f(o.mlp_forward(feat.view(np.uint16), dirf.view(np.uint16)))
    prev = 0.0
    for mode in (op.ACC_FP16_K16, op.ACC_FP16_K8, op.ACC_FP16_K4, op.ACC_FP16_STEP):
        got = _f(op.Oracle(desc, accumulate=mode).mlp_forward(feat.view(np.uint16), dirf.view(np.uint16)))
        d = np.abs(got[:, :3] - base[:, :3])
        assert d.max() <= 2e-2, (mode, d.max())  # random +-1 inputs: larger pre-activations than a scene's
        assert d.mean() >= prev * 0.9, (mode, d.mean(), prev)
        prev = float(d.mean())
    assert prev > 0


def test_numpy_restatement_of_one_layer_matches_every_mode():
    """The accumulate modes against an independent numpy restatement of ONE dot product: density MLP 32 -> 64 -> 16 whose
    output layer copies hidden neurons 0..15 and whose sigma activation is None, so that out4[3] is hidden neuron 0 after
    ReLU -- a 32-term dot product in the mode's arithmetic (two mma_sync K blocks in the reference)."""
    _, _, cfg0 = models.build_model(log2_hashmap_size=LOG2T, H=H, seed=SEED)
    params = np.array(G["params"], np.float32).copy()
    Wn, fin = 64, 32
    rng = np.random.default_rng(11)
    w0 = rng.uniform(-2, 2, (Wn, fin)).astype(np.float16).astype(np.float32)
    w1 = np.zeros((16, Wn), np.float32)
    w1[np.arange(16), np.arange(16)] = 1.0
    params[:Wn * fin] = w0.ravel()
    params[Wn * fin:Wn * fin + 16 * Wn] = w1.ravel()
    desc, keep = nh.desc_from_config(cfg0, params, G["density_grid"].astype(np.float32))
    desc.sigma_activation = 0  # NRF_ACT_NONE: wrap_a_activation passes the fp16 value through (nerf_network.h:32-47)
    feat = rng.uniform(-1, 1, (256, fin)).astype(np.float16)
    dirf = np.zeros((256, 16), np.float16)
    for mode in (op.ACC_FP32, op.ACC_FP16_K16, op.ACC_FP16_K8, op.ACC_FP16_K4, op.ACC_FP16_STEP):
        o = op.Oracle(desc, accumulate=mode)
        got = _f(o.mlp_forward(feat.view(np.uint16), dirf.view(np.uint16)))[:, 3]
        x = feat.astype(np.float32)
        want = np.empty(256, np.float32)
        for i in range(256):
            acc = np.float32(0)
            if mode == op.ACC_FP32:
                for k in range(fin):
                    acc = np.float32(acc + np.float32(w0[0, k] * x[i, k]))
            else:
                for kb in range(0, fin, mode):
                    part = acc
                    for k in range(kb, kb + mode):
                        part = np.float32(part + np.float32(w0[0, k] * x[i, k]))
                    acc = np.float32(np.float16(part))
            # hidden activation: ReLU on the accumulator, fp16 store; the output layer copies it (one product h * 1)
            want[i] = np.float32(np.float16(max(acc, np.float32(0))))
        np.testing.assert_array_equal(got, want)


def test_gap_between_fp32_and_fp16_accumulate_is_what_design_states():
    """The CPU-side figure quoted in DESIGN.md (c): config-2 model, 64x64 frame, oracle against oracle."""
    desc2, keep2, _ = models.build_model(log2_hashmap_size=19, H=128)
    base, bdepth, _ = op.Oracle(desc2).render(A["c2_cam"], A["c2_pose"], 64, 64, schedule=op.SCHED_PER_RAY)
    for mode, lo, hi in ((op.ACC_FP16_K16, 76.0, 82.0), (op.ACC_FP16_STEP, 69.0, 75.0)):
        psnr = models.psnr(A[f"c2_rgba_{mode}"], base)
        err = float(np.abs(A[f"c2_rgba_{mode}"] - base).max())
        assert lo <= psnr <= hi and err <= FRAME_TOL[mode][0], (mode, psnr, err)


@pytest.mark.gpu
def test_hip_frames_against_every_accumulate_mode():
    """HIP path (fp32 MFMA accumulation) against the oracle in each accumulate mode: the committed 64x64 frame of the
    config-2 model and the 128x64 crop of the 1920x1080 view of tests/test_parity_gpu.py, at the stated tolerances."""
    desc2, keep2, _ = models.build_model(log2_hashmap_size=19, H=128)
    ctx = nh.NerfHip(0)
    ctx.load_model(desc2)
    ctx.set_resolution(64, 64)
    ctx.render(A["c2_cam"], A["c2_pose"])
    got, gdepth = ctx.read_f32()
    report = {}
    for mode in (op.ACC_FP32, op.ACC_FP16_K16, op.ACC_FP16_STEP):
        if mode == op.ACC_FP32:
            want, wdepth, _ = op.Oracle(desc2).render(A["c2_cam"], A["c2_pose"], 64, 64, schedule=op.SCHED_PER_RAY)
        else:
            want, wdepth = A[f"c2_rgba_{mode}"], A[f"c2_depth_{mode}"]
        err, psnr = float(np.abs(got - want).max()), models.psnr(got, want)
        report[mode] = (err, psnr)
        assert err <= FRAME_TOL[mode][0] and psnr >= FRAME_TOL[mode][1], (mode, err, psnr)
        assert np.abs(gdepth - wdepth).max() <= 2.0 / 255.0
    # the HIP frame is (much) nearer to the fp32-accumulate oracle than to either emulation: it implements THAT contract
    assert report[op.ACC_FP32][1] > report[op.ACC_FP16_K16][1] > report[op.ACC_FP16_STEP][1]
    # BASELINE config 2 at full size: crop of the 1920x1080 view
    Wf, Hf = 1920, 1080
    cam, pose = syn.default_camera(Wf, Hf), syn.orbit_pose(30, 30)
    ctx.set_resolution(Wf, Hf)
    ctx.render(cam, pose)
    full, _ = ctx.read_f32()
    x0, y0, cw, ch = 896, 508, 128, 64
    ccam = cam.copy(); ccam[2] -= x0; ccam[3] -= y0
    crop = full[y0:y0 + ch, x0:x0 + cw]
    for mode in (op.ACC_FP32, op.ACC_FP16_K16, op.ACC_FP16_STEP):
        want, _, _ = op.Oracle(desc2, accumulate=mode).render(ccam, pose, cw, ch, schedule=op.SCHED_PER_RAY)
        err, psnr = float(np.abs(crop - want).max()), models.psnr(crop, want)
        assert err <= FRAME_TOL[mode][0] and psnr >= FRAME_TOL[mode][1], ("crop", mode, err, psnr)
    ctx.close()


@pytest.mark.gpu
def test_hip_network_outputs_against_fp16_accumulate():
    """Stage level: `nrf_mlp_forward` on the tiny scene's inputs against the FP16_K16 fixture.  Stated tolerance: the MLP
    tolerance of tests/test_parity_gpu.py widened by the accumulator's rounding -- |d| <= 8 * 2^-11 * |x| + 4e-3 for the
    sigmoid rgb outputs, and the pre-activation of sigma (log sigma) within 8 * 2^-11 * |x| + 2e-2."""
    torch = pytest.importorskip("torch")
    desc, keep = _tiny()
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    n = len(G["feat"])
    f_d = torch.from_numpy(G["feat"].view(np.int16)).cuda()
    d_d = torch.from_numpy(G["dirf"].view(np.int16)).cuda()
    out = torch.empty((n, 4), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    ctx.mlp_forward(f_d.data_ptr(), d_d.data_ptr(), n, out.data_ptr())
    got = out.cpu().numpy().astype(np.float64)
    want = _f(A[f"tiny_out4_{op.ACC_FP16_K16}"]).astype(np.float64)
    assert np.all(np.abs(got[:, :3] - want[:, :3]) <= 8 * 2.0 ** -11 * np.abs(want[:, :3]) + 4e-3)
    lg, lw = np.log(got[:, 3]), np.log(want[:, 3])
    assert np.all(np.abs(lg - lw) <= 8 * 2.0 ** -11 * np.abs(lw) + 2e-2)
    ctx.close()
