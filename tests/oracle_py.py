"""ctypes binding of oracle/libnerf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product (nerf-cuda_amd/) never does."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

import nerfhip as nh

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB = ORACLE_DIR / "libnerf_oracle.so"

SCHED_REFERENCE, SCHED_TILE64, SCHED_PER_RAY = 0, 1, 2
# nrfo_set_mlp_accumulate: fp32 sums (the contract shared with the HIP path) / fp16 accumulator rounded every n products
ACC_FP32, ACC_FP16_STEP, ACC_FP16_K4, ACC_FP16_K8, ACC_FP16_K16 = 0, 1, 4, 8, 16
_lib = None


def build():
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        build()
    L = C.CDLL(str(LIB))
    vp, u32, fp = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
    L.nrfo_last_error.restype = C.c_char_p
    L.nrfo_create.argtypes = [C.POINTER(nh.ModelDesc), C.POINTER(vp)]
    L.nrfo_destroy.argtypes = [vp]
    L.nrfo_destroy.restype = None
    L.nrfo_widths.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.nrfo_widths.restype = None
    L.nrfo_set_mlp_accumulate.argtypes = [vp, C.c_int]
    L.nrfo_set_contract.argtypes = [vp, C.c_int]
    L.nrfo_f32_to_f16_soft.argtypes = [C.c_float]
    L.nrfo_f32_to_f16_soft.restype = C.c_uint16
    L.nrfo_f16_to_f32_soft.argtypes = [C.c_uint16]
    L.nrfo_f16_to_f32_soft.restype = C.c_float
    L.nrfo_pcg32_first_float.argtypes = [C.c_uint64, C.c_uint64]
    L.nrfo_pcg32_first_float.restype = C.c_float
    L.nrfo_activation.argtypes = [C.c_uint32, C.c_float]
    L.nrfo_activation.restype = C.c_float
    L.nrfo_fp16_backend.restype = C.c_char_p
    L.nrfo_fp16_selfcheck.argtypes = [C.c_uint32]
    L.nrfo_fp16_selfcheck.restype = C.c_uint64
    L.nrfo_render_rays.argtypes = [vp, fp, fp, C.c_int, C.c_int, C.POINTER(nh.Options), C.c_int, C.c_int, vp, vp,
                                   C.POINTER(nh.Stats), vp, vp]
    L.nrfo_render_per_ray_rounds.argtypes = [vp, fp, fp, C.c_int, C.c_int, C.POINTER(nh.Options), vp, vp, C.POINTER(nh.Stats)]
    L.nrfo_f32_to_f16.argtypes = [C.c_float]
    L.nrfo_f32_to_f16.restype = C.c_uint16
    L.nrfo_f16_to_f32.argtypes = [C.c_uint16]
    L.nrfo_f16_to_f32.restype = C.c_float
    L.nrfo_nerf_matrix_to_ngp.argtypes = [fp, C.c_float, fp]
    L.nrfo_nerf_matrix_to_ngp.restype = None
    L.nrfo_fast_hash3.argtypes = [u32, u32, u32]
    L.nrfo_fast_hash3.restype = u32
    L.nrfo_grid_index.argtypes = [vp, u32, u32, u32, u32]
    L.nrfo_grid_index.restype = u32
    L.nrfo_encode_grid.argtypes = [vp, vp, u32, vp]
    L.nrfo_encode_dir.argtypes = [vp, vp, u32, vp]
    L.nrfo_mlp_forward.argtypes = [vp, vp, vp, u32, vp]
    L.nrfo_network.argtypes = [vp, vp, vp, u32, vp, vp]
    L.nrfo_generate_rays.argtypes = [vp, fp, fp, C.c_int, C.c_int, C.POINTER(nh.Options), vp, vp, vp, vp]
    L.nrfo_march.argtypes = [vp, C.POINTER(nh.Options), vp, vp, vp, vp, u32, u32, vp, vp, vp]
    L.nrfo_composite.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    L.nrfo_render.argtypes = [vp, fp, fp, C.c_int, C.c_int, C.POINTER(nh.Options), C.c_int, C.c_int, vp, vp,
                              C.POINTER(nh.Stats)]
    L.nrfo_density_grid.argtypes = [vp, C.c_int, C.c_float, vp, C.POINTER(C.c_float)]
    L.nrfo_quantize_u8.argtypes = [vp, vp, C.c_int, vp, vp]
    L.nrfo_quantize_u8.restype = None
    L.nrfo_max_threads.restype = C.c_int
    L.nrfo_rb_accumulate.argtypes = [vp, vp, C.c_int, C.c_float, C.c_int]
    L.nrfo_rb_accumulate.restype = None
    L.nrfo_rb_tonemap.argtypes = [vp, vp, C.c_int, C.c_float, fp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.nrfo_rb_tonemap.restype = None
    _lib = L
    return L


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def _ck(rc):
    if rc != 0:
        raise OracleError(rc, lib().nrfo_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    def __init__(self, desc: nh.ModelDesc, accumulate: int = ACC_FP32, contract: int = 0):
        self.L = lib()
        h = C.c_void_p()
        _ck(self.L.nrfo_create(C.byref(desc), C.byref(h)))
        self.h = h
        if accumulate != ACC_FP32:
            self.set_mlp_accumulate(accumulate)
        if contract:
            self.set_contract(contract)
        fw, dw = C.c_uint32(), C.c_uint32()
        self.L.nrfo_widths(h, C.byref(fw), C.byref(dw))
        self.feat_width, self.dir_width = int(fw.value), int(dw.value)  # padded encoding widths (MLP input widths)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nrfo_destroy(self.h)
            self.h = None

    def set_mlp_accumulate(self, mode):
        """MLP accumulator arithmetic: ACC_FP32 (default, the HIP path's) or the fp16-accumulator emulations of the
        reference's `__half` WMMA fragments (T/src/fully_fused_mlp.cu:69,334,437)."""
        _ck(self.L.nrfo_set_mlp_accumulate(self.h, int(mode)))

    def grid_index(self, level, x, y, z):
        return int(self.L.nrfo_grid_index(self.h, level, x, y, z))

    def encode_grid(self, pos01):
        pos01 = _f32(pos01).reshape(-1, 3)
        out = np.empty((len(pos01), self.feat_width), np.uint16)
        _ck(self.L.nrfo_encode_grid(self.h, pos01.ctypes.data, len(pos01), out.ctypes.data))
        return out

    def encode_dir(self, dir01):
        dir01 = _f32(dir01).reshape(-1, 3)
        out = np.empty((len(dir01), self.dir_width), np.uint16)
        _ck(self.L.nrfo_encode_dir(self.h, dir01.ctypes.data, len(dir01), out.ctypes.data))
        return out

    def mlp_forward(self, feat, dirfeat):
        feat = np.ascontiguousarray(feat, np.uint16)
        dirfeat = np.ascontiguousarray(dirfeat, np.uint16)
        out = np.empty((len(feat), 4), np.uint16)
        _ck(self.L.nrfo_mlp_forward(self.h, feat.ctypes.data, dirfeat.ctypes.data, len(feat), out.ctypes.data))
        return out

    def network(self, xyz, dirs):
        xyz, dirs = _f32(xyz).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        sigma = np.empty(len(xyz), np.float32)
        rgb = np.empty((len(xyz), 3), np.float32)
        _ck(self.L.nrfo_network(self.h, xyz.ctypes.data, dirs.ctypes.data, len(xyz), sigma.ctypes.data, rgb.ctypes.data))
        return sigma, rgb

    def generate_rays(self, cam, pose, W, H, opts=None):
        opts = opts or nh.default_options()
        cam, pose = _f32(cam).reshape(4), _f32(pose).reshape(16)
        o = np.empty((H * W, 3), np.float32)
        d = np.empty((H * W, 3), np.float32)
        nr = np.empty(H * W, np.float32)
        fr = np.empty(H * W, np.float32)
        _ck(self.L.nrfo_generate_rays(self.h, _fp(cam), _fp(pose), W, H, C.byref(opts), o.ctypes.data, d.ctypes.data,
                                      nr.ctypes.data, fr.ctypes.data))
        return o, d, nr, fr

    def march(self, rays_o, rays_d, rays_t, fars, n_step, opts=None):
        opts = opts or nh.default_options()
        rays_o, rays_d, rays_t, fars = _f32(rays_o), _f32(rays_d), _f32(rays_t), _f32(fars)
        n = len(rays_t)
        xyzs = np.empty((n, n_step, 3), np.float32)
        dirs = np.empty((n, n_step, 3), np.float32)
        deltas = np.empty((n, n_step, 2), np.float32)
        _ck(self.L.nrfo_march(self.h, C.byref(opts), rays_o.ctypes.data, rays_d.ctypes.data, rays_t.ctypes.data,
                              fars.ctypes.data, n, n_step, xyzs.ctypes.data, dirs.ctypes.data, deltas.ctypes.data))
        return xyzs, dirs, deltas

    def composite(self, sigmas, rgbs, deltas, rays_t, state):
        return composite(sigmas, rgbs, deltas, rays_t, state)

    def density_grid(self, n_cells, n_iterations=16, decay=0.95):
        grid = np.empty(n_cells, np.float32)
        mean = C.c_float()
        _ck(self.L.nrfo_density_grid(self.h, int(n_iterations), C.c_float(decay), grid.ctypes.data, C.byref(mean)))
        return grid, float(mean.value)

    def set_contract(self, on: int):
        """`a * b + c` as one fused multiply-add wherever the reference's device source has it in one expression (nvcc's
        default contraction); 0 / False (default): every operation rounded, the contract shared with the HIP path; 1 / True: of two
        products in a sum the left one is fused, products with other uses are fused as well; 2: the other choice at each of those
        sites (the sensitivity run: an emulation's uncertainty)."""
        _ck(self.L.nrfo_set_contract(self.h, int(on)))

    def render_rays(self, cam, pose, W, H, opts=None, schedule=SCHED_PER_RAY, n_threads=0):
        """render() + per ray the number of samples its march emitted and a hash of their (dt, t - last_t) bits."""
        opts = opts or nh.default_options()
        cam, pose = _f32(cam).reshape(4), _f32(pose).reshape(16)
        rgba = np.empty((H, W, 4), np.float32)
        depth = np.empty((H, W), np.float32)
        counts = np.empty((H, W), np.uint32)
        hashes = np.empty((H, W), np.uint64)
        st = nh.Stats()
        _ck(self.L.nrfo_render_rays(self.h, _fp(cam), _fp(pose), W, H, C.byref(opts), schedule, n_threads, rgba.ctypes.data,
                                    depth.ctypes.data, C.byref(st), counts.ctypes.data, hashes.ctypes.data))
        return rgba, depth, st, counts, hashes

    def render_per_ray_rounds(self, cam, pose, W, H, opts=None):
        """SCHED_PER_RAY through the reference's global round loop (n_step fixed to 1): the cross-check of the independent-ray form."""
        opts = opts or nh.default_options()
        cam, pose = _f32(cam).reshape(4), _f32(pose).reshape(16)
        rgba = np.empty((H, W, 4), np.float32)
        depth = np.empty((H, W), np.float32)
        st = nh.Stats()
        _ck(self.L.nrfo_render_per_ray_rounds(self.h, _fp(cam), _fp(pose), W, H, C.byref(opts), rgba.ctypes.data, depth.ctypes.data,
                                              C.byref(st)))
        return rgba, depth, st

    def render(self, cam, pose, W, H, opts=None, schedule=SCHED_REFERENCE, n_threads=0):
        opts = opts or nh.default_options()
        cam, pose = _f32(cam).reshape(4), _f32(pose).reshape(16)
        rgba = np.empty((H, W, 4), np.float32)
        depth = np.empty((H, W), np.float32)
        st = nh.Stats()
        _ck(self.L.nrfo_render(self.h, _fp(cam), _fp(pose), W, H, C.byref(opts), schedule, n_threads,
                               rgba.ctypes.data, depth.ctypes.data, C.byref(st)))
        return rgba, depth, st


def composite(sigmas, rgbs, deltas, rays_t, state):
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    rays_t, state = _f32(rays_t).copy(), _f32(state).copy()
    n, n_step = sigmas.shape
    _ck(lib().nrfo_composite(sigmas.ctypes.data, rgbs.ctypes.data, deltas.ctypes.data, n, n_step,
                             rays_t.ctypes.data, state.ctypes.data))
    return rays_t, state


def rb_accumulate(frame, accum, sample_count, color_space):
    frame, accum = _f32(frame), _f32(accum).copy()
    lib().nrfo_rb_accumulate(frame.ctypes.data, accum.ctypes.data, frame.size // 4, float(sample_count), color_space)
    return accum


def rb_tonemap(accum, exposure, bg, color_space, output_color_space, curve, clamp=False):
    accum, bg = _f32(accum), _f32(bg).reshape(4)
    out = np.empty_like(accum)
    lib().nrfo_rb_tonemap(accum.ctypes.data, out.ctypes.data, accum.size // 4, float(exposure), _fp(bg), color_space,
                          output_color_space, curve, int(clamp))
    return out


def rb_overlay_depth(surface, alpha, depth, depth_scale, fov_axis, zoom, center):
    surface, depth = _f32(surface).copy(), _f32(depth)
    H, W = surface.shape[:2]
    ih, iw = depth.shape
    L = lib()
    L.nrfo_rb_overlay_depth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_float]
    L.nrfo_rb_overlay_depth.restype = None
    L.nrfo_rb_overlay_depth(surface.ctypes.data, W, H, float(alpha), depth.ctypes.data, float(depth_scale), iw, ih, int(fov_axis),
                            float(zoom), float(center[0]), float(center[1]))
    return surface


def quantize_u8(rgba, depth):
    rgba, depth = _f32(rgba), _f32(depth)
    n = depth.size
    rgb8 = np.empty(rgba.shape[:-1] + (3,), np.uint8)
    d8 = np.empty(depth.shape, np.uint8)
    lib().nrfo_quantize_u8(rgba.ctypes.data, depth.ctypes.data, n, rgb8.ctypes.data, d8.ctypes.data)
    return rgb8, d8


def fp16_backend() -> str:
    return lib().nrfo_fp16_backend().decode()


def f16(x):
    return int(lib().nrfo_f32_to_f16(float(x)))


def f32_from_f16(h):
    return float(lib().nrfo_f16_to_f32(int(h)))
