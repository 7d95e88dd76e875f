"""ctypes binding of include/nerfhip.h (the C ABI of the HIP render path).

Host-side mirror, in Python, of what ngp::NerfRender does above its kernels:
  * `desc_from_config`  -- NerfRender::load_snapshot + reset_network
    (reference src/nerf_render.cu:431-473, 111-184) and the NerfNetwork ctor's
    defaulting (include/nerf-cuda/nerf_network.h:95-135): snapshot dict ->
    nrf_model_desc.
  * `NerfHip`           -- thin object over nrf_create / nrf_load_model /
    nrf_set_resolution / nrf_render ... used by tests, bench.py and smoke().

There is no CPU fallback here: if libnerfhip.so is missing or no gfx950 device
is present, construction raises.  The oracle (oracle/) is never imported from
this module.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libnerfhip.so"

NRF_ABI_VERSION = 6
GATHER_PEER_COPY, GATHER_RCCL = 0, 1  # nrf_group_set_gather
NRF_MAX_VIEWS = 128
NRF_OK, NRF_E_INVALID, NRF_E_UNSUPPORTED, NRF_E_NODEVICE, NRF_E_HIP, NRF_E_STATE, NRF_E_PARAMS = range(7)

ACT = {"none": 0, "relu": 1, "exponential": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5, "sine": 6}
DIR_SH, DIR_FREQUENCY, DIR_IDENTITY = 0, 1, 2
GRID_HASH, GRID_DENSE, GRID_TILED = 0, 1, 2
INTERP = {"linear": 0, "nearest": 1, "smoothstep": 2}  # T/.../grid.h:1383, common.h string_to_interpolation_type


class ModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("grid_type", C.c_uint32),
        ("n_levels", C.c_uint32),
        ("n_features_per_level", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32),
        ("per_level_scale", C.c_float),
        ("interpolation", C.c_uint32),
        ("n_neurons", C.c_uint32),
        ("density_hidden_layers", C.c_uint32),
        ("density_activation", C.c_uint32),
        ("density_output_activation", C.c_uint32),
        ("density_n_output", C.c_uint32),
        ("sigma_activation", C.c_uint32),
        ("rgb_hidden_layers", C.c_uint32),
        ("rgb_activation", C.c_uint32),
        ("rgb_output_activation", C.c_uint32),
        ("dir_encoding", C.c_uint32),
        ("sh_degree", C.c_uint32),
        ("n_frequencies", C.c_uint32),
        ("aabb", C.c_float * 6),
        ("bound", C.c_float),
        ("scale", C.c_float),
        ("cascade", C.c_uint32),
        ("density_grid_size", C.c_uint32),
        ("mean_density", C.c_float),
        ("params", C.POINTER(C.c_float)),
        ("n_params", C.c_uint64),
        ("density_grid", C.POINTER(C.c_float)),
        ("n_density_grid", C.c_uint64),
        ("gather_copy_budget_mb", C.c_uint32),
    ]


class LevelTable(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32),
        ("offset", C.c_uint32 * 17),
        ("resolution", C.c_uint32 * 16),
        ("scale", C.c_float * 16),
    ]


class Options(C.Structure):
    _fields_ = [
        ("bg_color", C.c_float),
        ("min_near", C.c_float),
        ("dt_gamma", C.c_float),
        ("max_steps", C.c_int32),
        ("density_scale", C.c_float),
        ("perturb", C.c_int32),
        ("shard_index", C.c_int32),
        ("shard_count", C.c_int32),
        ("fast_interp", C.c_int32),
        ("tile_major", C.c_int32),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("n_tiles", C.c_int32),
        ("rgba", C.c_void_p),
        ("depth", C.c_void_p),
        ("tile_major", C.c_int32),
        ("n_views", C.c_int32),
        ("view_stride_px", C.c_int64),
    ]


class HostFrame(C.Structure):
    """nrf_host_frame: 8-bit planes in pinned host memory owned by the context (nrf_submit_host_u8)."""
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("n_views", C.c_int32),
        ("rgb", C.c_void_p),
        ("depth", C.c_void_p),
        ("view_stride_px", C.c_int64),
        ("render_ms", C.c_float),
        ("copied_bytes", C.c_uint64),
    ]


NRF_HOST_RGB_ONLY = 1


class Stats(C.Structure):
    _fields_ = [
        ("n_rays", C.c_uint64),
        ("n_samples", C.c_uint64),
        ("n_rounds", C.c_uint64),
        ("n_network_evals", C.c_uint64),
        ("render_ms", C.c_float),
        ("n_composited", C.c_uint64),
        ("shader_clock_mhz", C.c_float),
        ("gather_addresses_per_sample", C.c_uint32),
        ("grid_device_bytes", C.c_uint64),
    ]


def default_options() -> Options:
    """Private member defaults of NerfRender (include/nerf-cuda/nerf_render.h:55-78)."""
    return Options(1.0, 0.2, 1.0 / 128.0, 1024, 1.0, 0, 0, 1, 0)


class NerfHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"nerfhip error {code}: {msg}")
        self.code = code


# --------------------------------------------------------------------------
# library loading
# --------------------------------------------------------------------------
_lib = None

_SIGS = {
    "nrf_last_error": (C.c_char_p, []),
    "nrf_abi_version": (C.c_int, []),
    "nrf_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nrf_destroy": (C.c_int, [C.c_void_p]),
    "nrf_default_options": (None, [C.POINTER(Options)]),
    "nrf_level_table_compute": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(LevelTable)]),
    "nrf_expected_n_params": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_uint64)]),
    "nrf_default_per_level_scale": (C.c_int, [C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "nrf_load_model": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc)]),
    "nrf_set_resolution": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "nrf_generate_density_grid": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float)]),
    "nrf_read_density_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_float)]),
    "nrf_set_options": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "nrf_render": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.POINTER(Frame)]),
    "nrf_set_max_views": (C.c_int, [C.c_void_p, C.c_int]),
    "nrf_render_views": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p,
                                   C.POINTER(Frame)]),
    "nrf_render_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p,
                                   C.POINTER(Frame)]),
    "nrf_read_view_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_read_view_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_bind_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_bind_output_rgbd8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nrf_bind_output_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_submit_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "nrf_wait_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(HostFrame)]),
    "nrf_render_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(HostFrame)]),
    "nrf_quantize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_untile_views_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_group_submit_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                           C.POINTER(C.c_int)]),
    "nrf_group_wait_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(HostFrame)]),
    "nrf_group_render_host_u8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                           C.POINTER(HostFrame)]),
    "nrf_render_async": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Frame)]),
    "nrf_sync": (C.c_int, [C.c_void_p]),
    "nrf_generate_rays_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "nrf_read_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_read_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_read_shard_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_untile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_quantize_rgbd8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "nrf_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "nrf_group_destroy": (C.c_int, [C.c_void_p]),
    "nrf_group_size": (C.c_int, [C.c_void_p]),
    "nrf_group_member": (C.c_void_p, [C.c_void_p, C.c_int]),
    "nrf_group_load_model": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc)]),
    "nrf_group_set_options": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "nrf_group_set_resolution": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "nrf_group_render_views": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(Frame)]),
    "nrf_group_read_view_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_group_read_view_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_group_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "nrf_group_set_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "nrf_group_get_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nrf_untile_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_tiles_per_shard": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "nrf_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "nrf_rb_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nrf_rb_destroy": (C.c_int, [C.c_void_p]),
    "nrf_rb_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "nrf_rb_reset_accumulation": (C.c_int, [C.c_void_p]),
    "nrf_rb_spp": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "nrf_rb_set_color_space": (C.c_int, [C.c_void_p, C.c_int]),
    "nrf_rb_set_tonemap_curve": (C.c_int, [C.c_void_p, C.c_int]),
    "nrf_rb_buffers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p)]),
    "nrf_rb_clear_frame": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nrf_rb_accumulate": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "nrf_rb_tonemap": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "nrf_rb_present": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p]),
    "nrf_rb_overlay_depth": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.POINTER(C.c_float), C.c_void_p]),
    "nrf_rb_host_to_accumulate_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "nrf_rb_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_encode_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "nrf_encode_dir": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "nrf_mlp_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "nrf_mlp_forward_repeat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "nrf_network": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_generate_rays": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_march": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrf_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
}


def exported_symbols():
    """Names include/nerfhip.h declares (checked against the .so by the CPU tests)."""
    return sorted(_SIGS)


def load_library(path: os.PathLike | None = None):
    """dlopen libnerfhip.so and attach signatures.  Fails loudly when missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(p))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing from the .so
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int):
    if rc != NRF_OK:
        raise NerfHipError(rc, load_library().nrf_last_error().decode("utf-8", "replace"))


# --------------------------------------------------------------------------
# snapshot dict -> nrf_model_desc  (load_snapshot + reset_network + NerfNetwork ctor)
# --------------------------------------------------------------------------
def _act(name, default):
    s = (name if name is not None else default).lower()
    if s not in ACT:
        raise ValueError(f"Invalid activation name: {name}")  # T/src/network.cu:41-60
    return ACT[s]


def default_per_level_scale(bound: float, base_resolution: int, n_levels: int) -> float:
    """fp32 exp(log(2048*bound/base)/(L-1)), reference src/nerf_render.cu:158-165.
    Evaluated by the C library so that libm's expf/logf are used."""
    out = C.c_float()
    _check(load_library().nrf_default_per_level_scale(C.c_float(bound), base_resolution, n_levels, C.byref(out)))
    return float(out.value)


# --------------------------------------------------------------------------
# instant-ngp snapshots (the reference itself only reads its own array form, nerf_render.cu:441-453)
# --------------------------------------------------------------------------
# Layout, from instant-ngp's public Testbed::save_snapshot / load_snapshot and NerfNetwork (the reference's
# nerf_network.h is derived from the latter, so the parameter order -- density MLP | rgb MLP | hash grid -- and the
# row-major [out][in] matrices are the same):
#   snapshot.params_binary      raw little-endian blob of snapshot.params_type ("__half" | "float"), snapshot.n_params values
#   snapshot.density_grid_size  128
#   snapshot.density_grid_binary  (K+1) * 128^3 values ("__half", or "float" in the first format revision -- told apart
#                               by the blob's size), K = log2(aabb_scale); value of cell (x, y, z) of cascade c at
#                               c * 128^3 + morton3D(x, y, z); cascade c is the cube of side 2^c around (0.5, 0.5, 0.5)
#   snapshot.nerf.aabb_scale    (also snapshot.nerf.dataset.aabb_scale); snapshot.nerf.dataset.scale / .offset = 0.33 / 0.5
#   snapshot.aabb               {"min": [3], "max": [3]} in instant-ngp's unit-cube coordinates
# Mapping onto the reference's conventions (torch-ngp's, which it restates): coordinates x_ref = x_ngp - 0.5,
# bound = aabb_scale / 2 (so x_ref / (2 bound) + 0.5 is instant-ngp's position inside its aabb); the reference's cascade k
# is the cube +-min(2^k, bound), i.e. instant-ngp's cascade k + 1 (cascade 0 for aabb_scale 1), each cell taking the
# maximum of its own value and of its eight children in the next finer instant-ngp cascade (instant-ngp's own bitfield
# is max-pooled that way); mean_density = mean(max(v, 0)) over instant-ngp's cascade 0, as instant-ngp computes it;
# per_level_scale, when the file leaves it out, is instant-ngp's exp(log(2048 aabb_scale / N_min) / (L - 1)); and
# the colour activation instant-ngp applies outside its network (logistic) becomes rgb_network.output_activation.
def morton3d(x, y, z):
    """x | y << 1 | z << 2 bit-interleaved (instant-ngp's morton3D; the reference carries the same helper, render_utils.h:157-170)."""
    def expand(v):
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    x, y, z = (np.asarray(a, np.uint64) for a in (x, y, z))
    return (expand(x) | (expand(y) << np.uint64(1)) | (expand(z) << np.uint64(2))).astype(np.uint32)


def blob_bytes(value, what: str) -> bytes:
    """A binary value of a snapshot as bytes.  tcnn writes its parameters with `gpu_memory_to_json_binary` (a nlohmann
    binary_t: msgpack `bin`) and READS either that or, from a text JSON, the object `{"bytes": [..], "subtype": ..}`
    (T/include/tiny-cuda-nn/gpu_memory_json.h:37-72): both are accepted here."""
    if isinstance(value, (bytes, bytearray, memoryview)):
        return bytes(value)
    if isinstance(value, dict) and "bytes" in value:
        return bytes(bytearray(int(v) & 0xff for v in value["bytes"]))
    raise RuntimeError(f"{what}: Invalid json type: must be either binary or object")  # gpu_memory_json.h:70


def is_ngp_snapshot(config: dict) -> bool:
    snap = config.get("snapshot", {})
    return isinstance(snap, dict) and ("nerf" in snap or isinstance(snap.get("aabb"), dict))


def ngp_snapshot_to_reference(config: dict) -> dict:
    """An instant-ngp snapshot dict -> the same model as a snapshot dict in the reference's own format."""
    snap = config["snapshot"]
    nerf = snap.get("nerf", {})
    dataset = nerf.get("dataset", {})
    aabb_scale = int(nerf.get("aabb_scale", dataset.get("aabb_scale", 1)))
    if aabb_scale < 1 or aabb_scale & (aabb_scale - 1):
        raise RuntimeError("instant-ngp snapshot: aabb_scale must be a power of two")
    offset = dataset.get("offset", [0.5, 0.5, 0.5])
    if any(abs(float(v) - 0.5) > 1e-6 for v in offset):
        raise NotImplementedError("instant-ngp snapshot: dataset.offset other than 0.5 has no counterpart in the reference")
    H = int(snap.get("density_grid_size", 128))
    if H & (H - 1):
        raise RuntimeError("instant-ngp snapshot: density_grid_size must be a power of two (Morton order)")
    n_ngp = aabb_scale.bit_length()  # K + 1 cascades
    blob = blob_bytes(snap["density_grid_binary"], "snapshot.density_grid_binary")
    cells = n_ngp * H ** 3
    kind = snap.get("density_grid_type")
    if kind is None:
        kind = "float" if len(blob) == 4 * cells else "__half"
    grid = np.frombuffer(blob, np.float32 if kind == "float" else np.float16).astype(np.float32)
    if grid.size < cells:
        raise RuntimeError("Incompatible number of grid cascades.")
    grid = grid[:cells].reshape(n_ngp, H ** 3)
    ax = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    m = morton3d(X, Y, Z).reshape(-1)
    xmajor = grid[:, m].reshape(n_ngp, H, H, H)  # [cascade][x][y][z]
    bound = aabb_scale / 2.0
    C = 1 if aabb_scale == 1 else n_ngp - 1
    out = np.empty((C, H, H, H), np.float32)
    for k in range(C):
        c = 0 if aabb_scale == 1 else k + 1
        own = xmajor[c].copy()
        if c > 0:  # the inner half of cascade c is covered twice as finely by cascade c - 1
            fine = xmajor[c - 1].reshape(H // 2, 2, H // 2, 2, H // 2, 2).max(axis=(1, 3, 5))
            q = H // 4
            inner = own[q:q + H // 2, q:q + H // 2, q:q + H // 2]
            np.maximum(inner, fine, out=inner)
        out[k] = own
    lo, hi = snap.get("aabb", {}).get("min"), snap.get("aabb", {}).get("max")
    if lo is None or hi is None:
        lo, hi = [0.5 - bound] * 3, [0.5 + bound] * 3
    ref = {k: (dict(v) if isinstance(v, dict) else v) for k, v in config.items() if k != "snapshot"}
    enc = ref.setdefault("encoding", {})
    if not float(enc.get("per_level_scale", 0.0)) > 0.0 and int(enc.get("n_levels", 16)) > 1:
        base = int(enc.get("base_resolution", 0)) or (1 << (int(enc.get("log2_hashmap_size", 15)) // 3))
        enc["per_level_scale"] = default_per_level_scale(float(aabb_scale), base, int(enc.get("n_levels", 16)))
    rgb = ref.setdefault("rgb_network", {})
    if str(rgb.get("output_activation", "None")).lower() == "none":
        rgb["output_activation"] = "Sigmoid" if str(nerf.get("rgb_activation", "Logistic")).lower() in ("logistic", "sigmoid") else "None"
    new = {"aabb": [float(v) - 0.5 for v in lo] + [float(v) - 0.5 for v in hi], "bound": float(bound),
           "scale": float(dataset.get("scale", 0.33)), "cascade": int(C), "density_grid_size": H,
           "mean_density": float(np.maximum(xmajor[0], 0.0).mean(dtype=np.float64)),
           "density_grid_binary": out.reshape(-1).tobytes(), "density_grid_type": "float",
           "params_binary": blob_bytes(snap["params_binary"], "snapshot.params_binary"), "params_type": snap.get("params_type", "__half")}
    # tcnn's Trainer::serialize writes `n_params` beside the blob (T/include/tiny-cuda-nn/trainer.h:267-279: sizeof(PARAMS_T) *
    # n_params bytes of the inference parameters, PARAMS_T = __half for a FullyFusedMLP model); a file whose two disagree is corrupt
    if "n_params" in snap:
        width = 4 if new["params_type"] == "float" else 2
        if int(snap["n_params"]) * width != len(new["params_binary"]):
            raise RuntimeError(f"snapshot.params_binary holds {len(new['params_binary']) // width} values, snapshot.n_params says {int(snap['n_params'])}")
    ref["snapshot"] = new
    return ref


def desc_from_config(config: dict, params: np.ndarray | None = None, density_grid: np.ndarray | None = None):
    """Build a nrf_model_desc from a snapshot dict in the reference's format
     -- or in instant-ngp's, which is converted first.  Returns (desc, keepalive)."""
    if "snapshot" not in config:
        raise RuntimeError("File does not contain a snapshot.")  # nerf_render.cu:434-436
    if is_ngp_snapshot(config):
        config = ngp_snapshot_to_reference(config)
    snap = config["snapshot"]
    enc = dict(config.get("encoding", {}))
    net = dict(config.get("network", {}))
    rgb = dict(config.get("rgb_network", {}))
    dire = dict(config.get("dir_encoding", {}))

    d = ModelDesc()
    d.abi_version = NRF_ABI_VERSION
    # snapshot block, nerf_render.cu:441-453 with member defaults nerf_render.h:55-67
    aabb = [float(v) for v in snap["aabb"]]
    d.aabb = (C.c_float * 6)(*aabb)
    d.bound = float(snap.get("bound", 1))
    d.scale = float(snap.get("scale", 0.33))
    d.cascade = int(snap.get("cascade", 1))
    d.density_grid_size = int(snap.get("density_grid_size", 128))
    d.mean_density = float(snap.get("mean_density", 1e-4))

    # encoding block, nerf_render.cu:125-165 and grid.h:1365-1386
    otype = enc.get("otype", "OneBlob").lower()
    if "grid" not in otype:
        raise NotImplementedError(f"position encoding '{otype}' is outside the hot path (hash grid only)")
    default_type = "tiled" if otype == "tiledgrid" else ("dense" if otype == "densegrid" else "hash")
    d.grid_type = {"hash": GRID_HASH, "dense": GRID_DENSE, "tiled": GRID_TILED}[enc.get("type", default_type).lower()]
    F = int(enc.get("n_features_per_level", 2))
    d.n_features_per_level = F
    if enc.get("n_features", 0) and enc["n_features"] > 0:
        if "n_levels" in enc:
            raise RuntimeError("GridEncoding: may not specify n_features and n_levels simultaneously")
        d.n_levels = int(enc["n_features"]) // F
    else:
        d.n_levels = int(enc.get("n_levels", 16))
    d.log2_hashmap_size = int(enc.get("log2_hashmap_size", 15))  # the reference's default, nerf_render.cu:144-145
    base = int(enc.get("base_resolution", 0))
    if not base:
        base = 1 << (d.log2_hashmap_size // 3)
    d.base_resolution = base
    pls = float(enc.get("per_level_scale", 0.0))
    if pls <= 0.0 and d.n_levels > 1:
        pls = default_per_level_scale(d.bound, base, d.n_levels)
    if pls <= 0.0:
        pls = 2.0
    d.per_level_scale = pls
    interp = enc.get("interpolation", "Linear").lower()
    if interp not in INTERP:
        raise RuntimeError(f"Invalid interpolation type: {interp}")
    d.interpolation = INTERP[interp]
    if F not in (1, 2, 4, 8):
        raise RuntimeError("GridEncoding: n_features_per_level must be 1, 2, 4, or 8.")  # grid.h:1403-1411

    # network blocks, T/src/network.cu:127-143 + nerf_network.h:117-135
    def mlp(cfg):
        ot = cfg.get("otype", "FullyFusedMLP").lower()
        if ot not in ("fullyfusedmlp", "megakernelmlp", "mlp", "cutlassmlp"):
            raise RuntimeError(f"Invalid network type: {ot}")
        return int(cfg.get("n_neurons", 128)), int(cfg.get("n_hidden_layers", 5)), \
            _act(cfg.get("activation"), "ReLU"), _act(cfg.get("output_activation"), "None")

    n1, h1, a1, o1 = mlp(net)
    n2, h2, a2, o2 = mlp(rgb)
    if n1 != n2:
        raise NotImplementedError("density and rgb networks must share n_neurons")
    d.n_neurons = n1
    d.density_hidden_layers, d.density_activation, d.density_output_activation = h1, a1, o1
    d.density_n_output = int(net.get("n_output_dims", 16))
    d.sigma_activation = _act(net.get("sigma_activation"), "Exponential")
    d.rgb_hidden_layers, d.rgb_activation, d.rgb_output_activation = h2, a2, o2

    # dir_encoding: Composite{nested[0] on 3 dims, Identity on the remaining 0 dims (dropped)}
    # T/include/tiny-cuda-nn/encodings/composite.h:139-216
    node = dire
    if node.get("otype", "Composite").lower() == "composite":
        # a nested encoding without n_dims_to_encode takes the remaining dims; with all 3
        # consumed by the first, the rest encode 0 dims and are dropped (composite.h:182-185)
        covering = [n for n in node.get("nested", []) if int(n.get("n_dims_to_encode", 0)) == 3]
        if not covering and len(node.get("nested", [])) == 1 and "n_dims_to_encode" not in node["nested"][0]:
            covering = [node["nested"][0]]
        if not covering:
            raise NotImplementedError("dir_encoding must encode all 3 direction dims with one nested encoding")
        node = covering[0]
    ot = node.get("otype", "").lower()
    if ot == "sphericalharmonics":
        d.dir_encoding, d.sh_degree = DIR_SH, int(node.get("degree", 4))
    elif ot == "frequency":
        d.dir_encoding, d.n_frequencies = DIR_FREQUENCY, int(node.get("n_frequencies", 12))
    elif ot == "identity":
        d.dir_encoding = DIR_IDENTITY
    else:
        raise NotImplementedError(f"dir encoding '{ot}' is outside the hot path")

    def numbers(key):
        """`key` as an array of numbers (the reference's format) or `key_binary` + `key_type` (instant-ngp's)."""
        if key in snap:
            return np.asarray(snap[key], dtype=np.float32)
        blob = blob_bytes(snap[key + "_binary"], f"snapshot.{key}_binary")
        kind = snap.get(key + "_type", "__half")
        if kind not in ("__half", "half", "float"):
            raise RuntimeError(f"snapshot.{key}_type: unknown element type '{kind}'")
        return np.frombuffer(blob, np.float32 if kind == "float" else np.float16).astype(np.float32)

    p = np.ascontiguousarray(params if params is not None else numbers("params"), dtype=np.float32)
    d.params = p.ctypes.data_as(C.POINTER(C.c_float))
    d.n_params = p.size
    if density_grid is None and "density_grid" not in snap and "density_grid_binary" not in snap:
        # no grid in the snapshot: NerfRender::generate_density_grid (nrf_generate_density_grid) makes one from the network
        d.density_grid = None
        d.n_density_grid = 0
        return d, (p, None)
    g = np.ascontiguousarray(density_grid if density_grid is not None else numbers("density_grid"), dtype=np.float32)
    d.density_grid = g.ctypes.data_as(C.POINTER(C.c_float))
    d.n_density_grid = g.size
    return d, (p, g)


def level_table(desc: ModelDesc) -> LevelTable:
    t = LevelTable()
    _check(load_library().nrf_level_table_compute(C.byref(desc), C.byref(t)))
    return t


def expected_n_params(desc: ModelDesc) -> int:
    n = C.c_uint64()
    _check(load_library().nrf_expected_n_params(C.byref(desc), C.byref(n)))
    return int(n.value)


def tiles_per_shard(width: int, height: int, shard_count: int) -> int:
    n = C.c_int()
    _check(load_library().nrf_tiles_per_shard(width, height, shard_count, C.byref(n)))
    return int(n.value)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _host_frame_arrays(f: HostFrame, copy: bool):
    """numpy views (or copies) of a HostFrame's pinned planes: rgb u8 [n][H][W][3], depth u8 [n][H][W] (or None)."""
    n, h, w, stride = f.n_views, f.height, f.width, int(f.view_stride_px)
    rgb = np.ctypeslib.as_array(C.cast(f.rgb, C.POINTER(C.c_uint8)), shape=(n, stride * 3))[:, :h * w * 3].reshape(n, h, w, 3)
    depth = None
    if f.depth:
        depth = np.ctypeslib.as_array(C.cast(f.depth, C.POINTER(C.c_uint8)), shape=(n, stride))[:, :h * w].reshape(n, h, w)
    if copy:
        rgb = rgb.copy()
        depth = None if depth is None else depth.copy()
    return rgb, depth


def _views(cams, poses):
    cams = np.ascontiguousarray(cams, dtype=np.float32).reshape(-1, 4)
    poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
    if len(cams) != len(poses):
        raise ValueError("cams and poses must have the same length")
    return cams, poses


class NerfHip:
    """One context on one device (nrf_context)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.nrf_create(device, C.byref(h)))
        self.h = h
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.nrf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_model(self, desc: ModelDesc):
        _check(self.lib.nrf_load_model(self.h, C.byref(desc)))

    def set_resolution(self, width: int, height: int):
        _check(self.lib.nrf_set_resolution(self.h, width, height))
        self.width, self.height = width, height

    def generate_density_grid(self, n_iterations: int = 16, decay: float = 0.95) -> float:
        """nrf_generate_density_grid; returns the new mean_density."""
        m = C.c_float()
        _check(self.lib.nrf_generate_density_grid(self.h, int(n_iterations), C.c_float(decay), C.byref(m)))
        return float(m.value)

    def read_density_grid(self, n_cells: int):
        grid = np.empty(n_cells, np.float32)
        m = C.c_float()
        _check(self.lib.nrf_read_density_grid(self.h, grid.ctypes.data, n_cells, C.byref(m)))
        return grid, float(m.value)

    def set_options(self, opts: Options):
        _check(self.lib.nrf_set_options(self.h, C.byref(opts)))

    def render(self, cam, pose, stream=None) -> Frame:
        cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(4)
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        f = Frame()
        _check(self.lib.nrf_render(self.h, _fptr(cam), _fptr(pose), C.c_void_p(stream or 0), C.byref(f)))
        return f

    def set_max_views(self, n: int):
        _check(self.lib.nrf_set_max_views(self.h, int(n)))

    def render_views(self, cams, poses, stream=None) -> Frame:
        """nrf_render_views: cams [n][4], poses [n][4][4]; one launch per NRF_MAX_VIEWS views."""
        cams = np.ascontiguousarray(cams, dtype=np.float32).reshape(-1, 4)
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
        if len(cams) != len(poses):
            raise ValueError("cams and poses must have the same length")
        f = Frame()
        _check(self.lib.nrf_render_views(self.h, len(cams), _fptr(cams), _fptr(poses), C.c_void_p(stream or 0), C.byref(f)))
        return f

    def read_view_f32(self, view: int):
        rgba = np.empty((self.height, self.width, 4), np.float32)
        depth = np.empty((self.height, self.width), np.float32)
        _check(self.lib.nrf_read_view_f32(self.h, int(view), rgba.ctypes.data, depth.ctypes.data))
        return rgba, depth

    def read_view_u8(self, view: int):
        rgb = np.empty((self.height, self.width, 3), np.uint8)
        depth = np.empty((self.height, self.width), np.uint8)
        _check(self.lib.nrf_read_view_u8(self.h, int(view), rgb.ctypes.data, depth.ctypes.data))
        return rgb, depth

    def bind_output(self, rgba_ptr, depth_ptr):
        _check(self.lib.nrf_bind_output(self.h, C.c_void_p(rgba_ptr or 0), C.c_void_p(depth_ptr or 0)))

    def bind_output_rgbd8(self, rgbd8_ptr):
        """Packed 8-bit target (r | g << 8 | b << 16 | depth << 24 per pixel) of subsequent renders; 0: back to the float planes."""
        _check(self.lib.nrf_bind_output_rgbd8(self.h, C.c_void_p(rgbd8_ptr or 0)))

    def bind_output_u8(self, rgb8_ptr, depth8_ptr):
        """8-bit planar targets (the reference's Image layout) of subsequent renders; 0, 0: back to the float planes."""
        _check(self.lib.nrf_bind_output_u8(self.h, C.c_void_p(rgb8_ptr or 0), C.c_void_p(depth8_ptr or 0)))

    # ---- host frames: render_frame's result in (pinned) host memory
    def submit_host_u8(self, cams, poses, flags: int = 0) -> int:
        cams, poses = _views(cams, poses)
        t = C.c_int(-1)
        _check(self.lib.nrf_submit_host_u8(self.h, len(cams), _fptr(cams), _fptr(poses), int(flags), C.byref(t)))
        return int(t.value)

    def wait_host_u8(self, ticket: int, copy: bool = True):
        """(rgb u8 [n][H][W][3], depth u8 [n][H][W] or None); copy=False: views of the context's pinned memory, valid
        until the second next submit."""
        f = HostFrame()
        _check(self.lib.nrf_wait_host_u8(self.h, int(ticket), C.byref(f)))
        return _host_frame_arrays(f, copy)

    def render_host_u8(self, cams, poses, flags: int = 0, copy: bool = True):
        cams, poses = _views(cams, poses)
        f = HostFrame()
        _check(self.lib.nrf_render_host_u8(self.h, len(cams), _fptr(cams), _fptr(poses), int(flags), C.byref(f)))
        return _host_frame_arrays(f, copy)

    def render_host_u8_raw(self, cams, poses, flags: int = 0) -> HostFrame:
        """nrf_render_host_u8 with prepared float32 arrays (cams [n][4], poses [n][16]); returns the HostFrame itself
        (pointers into the context's pinned memory, render_ms, copied_bytes): the timing form bench.py uses."""
        f = HostFrame()
        _check(self.lib.nrf_render_host_u8(self.h, len(cams), _fptr(cams), _fptr(poses), int(flags), C.byref(f)))
        return f

    def read_f32(self):
        rgba = np.empty((self.height, self.width, 4), np.float32)
        depth = np.empty((self.height, self.width), np.float32)
        _check(self.lib.nrf_read_f32(self.h, rgba.ctypes.data, depth.ctypes.data))
        return rgba, depth

    def read_u8(self):
        rgb = np.empty((self.height, self.width, 3), np.uint8)
        depth = np.empty((self.height, self.width), np.uint8)
        _check(self.lib.nrf_read_u8(self.h, rgb.ctypes.data, depth.ctypes.data))
        return rgb, depth

    def stats(self) -> Stats:
        s = Stats()
        _check(self.lib.nrf_get_stats(self.h, C.byref(s)))
        return s

    def untile(self, gathered_ptr, shard_count, tiles, channels, out_ptr, stream=None):
        _check(self.lib.nrf_untile(self.h, C.c_void_p(gathered_ptr), shard_count, tiles, channels,
                                   C.c_void_p(out_ptr), C.c_void_p(stream or 0)))

    def quantize_rgbd8(self, rgba_ptr, depth_ptr, n_px, out_ptr, stream=None):
        _check(self.lib.nrf_quantize_rgbd8(self.h, C.c_void_p(rgba_ptr), C.c_void_p(depth_ptr), int(n_px), C.c_void_p(out_ptr),
                                           C.c_void_p(stream or 0)))

    def quantize_u8(self, rgba_ptr, depth_ptr, n_px, rgb8_ptr, depth8_ptr, stream=None):
        _check(self.lib.nrf_quantize_u8(self.h, C.c_void_p(rgba_ptr), C.c_void_p(depth_ptr), int(n_px), C.c_void_p(rgb8_ptr),
                                        C.c_void_p(depth8_ptr), C.c_void_p(stream or 0)))

    def untile_views_u8(self, gathered_ptr, shard_count, tiles, n_views, rgb8_ptr, depth8_ptr, stream=None):
        _check(self.lib.nrf_untile_views_u8(self.h, C.c_void_p(gathered_ptr), shard_count, tiles, n_views, C.c_void_p(rgb8_ptr),
                                            C.c_void_p(depth8_ptr), C.c_void_p(stream or 0)))

    def untile_views(self, gathered_ptr, shard_count, tiles, channels, n_views, out_ptr, stream=None):
        _check(self.lib.nrf_untile_views(self.h, C.c_void_p(gathered_ptr), shard_count, tiles, channels, n_views,
                                         C.c_void_p(out_ptr), C.c_void_p(stream or 0)))

    # ---- stage entry points; arguments are device pointers (ints) ----
    def encode_grid(self, pos01, n, out, stream=None):
        _check(self.lib.nrf_encode_grid(self.h, C.c_void_p(pos01), n, C.c_void_p(out), C.c_void_p(stream or 0)))

    def encode_dir(self, dir01, n, out, stream=None):
        _check(self.lib.nrf_encode_dir(self.h, C.c_void_p(dir01), n, C.c_void_p(out), C.c_void_p(stream or 0)))

    def mlp_forward(self, feat, dirfeat, n, out, stream=None):
        _check(self.lib.nrf_mlp_forward(self.h, C.c_void_p(feat), C.c_void_p(dirfeat), n, C.c_void_p(out),
                                        C.c_void_p(stream or 0)))

    def mlp_forward_repeat(self, feat, dirfeat, n, out, repeat, stream=None):
        _check(self.lib.nrf_mlp_forward_repeat(self.h, C.c_void_p(feat), C.c_void_p(dirfeat), n, C.c_void_p(out), int(repeat),
                                               C.c_void_p(stream or 0)))

    def network(self, xyz, dirs, n, sigma, rgb, stream=None):
        _check(self.lib.nrf_network(self.h, C.c_void_p(xyz), C.c_void_p(dirs), n, C.c_void_p(sigma), C.c_void_p(rgb),
                                    C.c_void_p(stream or 0)))

    def generate_rays(self, cam, pose, rays_o, rays_d, nears, fars, stream=None):
        cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(4)
        pose = np.ascontiguousarray(pose, dtype=np.float32).reshape(16)
        _check(self.lib.nrf_generate_rays(self.h, _fptr(cam), _fptr(pose), C.c_void_p(rays_o), C.c_void_p(rays_d),
                                          C.c_void_p(nears), C.c_void_p(fars), C.c_void_p(stream or 0)))

    def march(self, rays_o, rays_d, rays_t, fars, n, n_step, xyzs, dirs, deltas, stream=None):
        _check(self.lib.nrf_march(self.h, C.c_void_p(rays_o), C.c_void_p(rays_d), C.c_void_p(rays_t), C.c_void_p(fars),
                                  n, n_step, C.c_void_p(xyzs), C.c_void_p(dirs), C.c_void_p(deltas),
                                  C.c_void_p(stream or 0)))

    def composite(self, sigmas, rgbs, deltas, n, n_step, rays_t, state, stream=None):
        _check(self.lib.nrf_composite(self.h, C.c_void_p(sigmas), C.c_void_p(rgbs), C.c_void_p(deltas), n, n_step,
                                      C.c_void_p(rays_t), C.c_void_p(state), C.c_void_p(stream or 0)))


CS_LINEAR, CS_SRGB, CS_VISPOSNEG = 0, 1, 2
TM_IDENTITY, TM_ACES, TM_HABLE, TM_REINHARD = 0, 1, 2, 3


class NerfGroup:
    """nrf_group: several devices driven by this process (the reference's NGPU, common.h:91)."""

    def __init__(self, devices):
        self.lib = load_library()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(self.lib.nrf_group_create(len(devices), devs, C.byref(h)))
        self.h = h
        self.width = self.height = 0

    def close(self):
        if self.h:
            self.lib.nrf_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def load_model(self, desc):
        _check(self.lib.nrf_group_load_model(self.h, C.byref(desc)))

    def set_gather(self, mode: int):
        """The transport of the exchange step: GATHER_PEER_COPY (hipMemcpyPeerAsync) or GATHER_RCCL (ncclSend / ncclRecv to
        devices[0]; a one-member group then runs the whole exchange too)."""
        _check(self.lib.nrf_group_set_gather(self.h, int(mode)))

    def gather(self):
        """(mode, RCCL version code or 0)"""
        m, v = C.c_int(), C.c_int()
        _check(self.lib.nrf_group_get_gather(self.h, C.byref(m), C.byref(v)))
        return int(m.value), int(v.value)

    def set_options(self, opts):
        _check(self.lib.nrf_group_set_options(self.h, C.byref(opts)))

    def generate_density_grid(self, n_iterations: int = 16, decay: float = 0.95) -> float:
        """Every member evaluates the density grid from its replica of the network (nrf_generate_density_grid)."""
        m = C.c_float()
        for i in range(self.lib.nrf_group_size(self.h)):
            ctx = C.c_void_p(self.lib.nrf_group_member(self.h, i))
            _check(self.lib.nrf_generate_density_grid(ctx, int(n_iterations), C.c_float(decay), C.byref(m)))
        return float(m.value)

    def set_resolution(self, width, height):
        _check(self.lib.nrf_group_set_resolution(self.h, width, height))
        self.width, self.height = width, height

    def render_views(self, cams, poses) -> Frame:
        cams = np.ascontiguousarray(cams, dtype=np.float32).reshape(-1, 4)
        poses = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 16)
        f = Frame()
        _check(self.lib.nrf_group_render_views(self.h, len(cams), _fptr(cams), _fptr(poses), C.byref(f)))
        return f

    def read_view_f32(self, view):
        rgba = np.empty((self.height, self.width, 4), np.float32)
        depth = np.empty((self.height, self.width), np.float32)
        _check(self.lib.nrf_group_read_view_f32(self.h, int(view), rgba.ctypes.data, depth.ctypes.data))
        return rgba, depth

    def read_view_u8(self, view):
        rgb = np.empty((self.height, self.width, 3), np.uint8)
        depth = np.empty((self.height, self.width), np.uint8)
        _check(self.lib.nrf_group_read_view_u8(self.h, int(view), rgb.ctypes.data, depth.ctypes.data))
        return rgb, depth

    def submit_host_u8(self, cams, poses, flags: int = 0) -> int:
        cams, poses = _views(cams, poses)
        t = C.c_int(-1)
        _check(self.lib.nrf_group_submit_host_u8(self.h, len(cams), _fptr(cams), _fptr(poses), int(flags), C.byref(t)))
        return int(t.value)

    def wait_host_u8(self, ticket: int, copy: bool = True):
        f = HostFrame()
        _check(self.lib.nrf_group_wait_host_u8(self.h, int(ticket), C.byref(f)))
        return _host_frame_arrays(f, copy)

    def render_host_u8(self, cams, poses, flags: int = 0, copy: bool = True):
        cams, poses = _views(cams, poses)
        f = HostFrame()
        _check(self.lib.nrf_group_render_host_u8(self.h, len(cams), _fptr(cams), _fptr(poses), int(flags), C.byref(f)))
        return _host_frame_arrays(f, copy)

    def stats(self) -> Stats:
        s = Stats()
        _check(self.lib.nrf_group_get_stats(self.h, C.byref(s)))
        return s


class NerfRender:
    """Python mirror of ngp::NerfRender (reference include/nerf-cuda/nerf_render.h:29-50): the same method names
    and call order as the reference's testbed (src/main.cu:140-170), on an nrf_group of `n_gpus` devices.

        render = NerfRender()
        render.reload_network_from_file("snapshot.msgpack")
        render.set_resolution((1920, 1080))
        rgb, depth = render.render_frame(cam, pose)          # u8 [H][W][3], u8 [H][W]  (ngp::Image)
    """

    def __init__(self, n_gpus: int = 1, devices=None):
        self.group = NerfGroup(list(devices) if devices is not None else list(range(n_gpus)))
        self.network_config = None
        self.desc = None
        self._keep = None
        self.resolution = None

    def load_network_config(self, path):  # nerf_render.cu:66-91: .json or .msgpack
        path = str(path)
        if not os.path.exists(path):
            raise RuntimeError(f"Network config path {path} does not exist.")
        if path.lower().endswith(".json"):
            import json

            with open(path) as f:
                return json.load(f)
        import msgpack

        with open(path, "rb") as f:
            return msgpack.unpackb(f.read(), raw=False)

    def load_snapshot(self, path):  # nerf_render.cu:431-473
        config = self.load_network_config(path)
        if "snapshot" not in config:
            raise RuntimeError(f"File {path} does not contain a snapshot.")
        self.network_config = config

    def reset_network(self):  # nerf_render.cu:111-184
        self.desc, self._keep = desc_from_config(self.network_config)
        self.desc.gather_copy_budget_mb = int(getattr(self, "gather_copy_budget_mb", 0))  # 0: the library's default; 1: no gather copies

    def reload_network_from_file(self, path):  # nerf_render.cu:93-109
        self.load_snapshot(path)
        self.reset_network()
        self.group.load_model(self.desc)
        if self.desc.n_density_grid == 0:  # the snapshot carries no density grid
            self.generate_density_grid()

    def generate_density_grid(self):  # nerf_render.cu:388-429 (dead in the reference; see nrf_generate_density_grid)
        if self.desc is None:
            raise RuntimeError("generate_density_grid: no network loaded")
        self.desc.mean_density = self.group.generate_density_grid(16, 0.95)

    def set_resolution(self, resolution):  # nerf_render.cu:186-236
        self.resolution = (int(resolution[0]), int(resolution[1]))
        self.group.set_resolution(*self.resolution)

    def render_frame(self, cam, pos):  # nerf_render.cu:238-367
        return self.render_frames([cam], [pos])[0]

    def render_frames(self, cams, poses):
        """Batched form: one launch per NRF_MAX_VIEWS cameras and device; [(rgb u8, depth u8), ...]."""
        if self.desc is None or self.resolution is None:
            raise RuntimeError("reload_network_from_file and set_resolution must be called first")
        # the kernel writes the Image's bytes itself; they arrive in pinned host memory by one asynchronous copy
        rgb, depth = self.group.render_host_u8(cams, poses)
        return [(rgb[v], depth[v]) for v in range(len(rgb))]

    def close(self):
        self.group.close()


class RenderBuffer:
    """nrf_render_buffer: the presentation chain of the reference's CudaRenderBuffer (same method names)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.nrf_rb_create(device, C.byref(h)))
        self.h = h
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.nrf_rb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, width, height):
        _check(self.lib.nrf_rb_resize(self.h, width, height))
        self.width, self.height = width, height

    def reset_accumulation(self):
        _check(self.lib.nrf_rb_reset_accumulation(self.h))

    def spp(self):
        v = C.c_uint32()
        _check(self.lib.nrf_rb_spp(self.h, C.byref(v)))
        return int(v.value)

    def set_color_space(self, cs):
        _check(self.lib.nrf_rb_set_color_space(self.h, cs))

    def set_tonemap_curve(self, curve):
        _check(self.lib.nrf_rb_set_tonemap_curve(self.h, curve))

    def buffers(self):
        f, d, a, s = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self.lib.nrf_rb_buffers(self.h, C.byref(f), C.byref(d), C.byref(a), C.byref(s)))
        return f.value, d.value, a.value, s.value

    def clear_frame(self, stream=None):
        _check(self.lib.nrf_rb_clear_frame(self.h, C.c_void_p(stream or 0)))

    def accumulate(self, exposure=0.0, stream=None):
        _check(self.lib.nrf_rb_accumulate(self.h, C.c_float(exposure), C.c_void_p(stream or 0)))

    def tonemap(self, exposure, background_color, output_color_space, stream=None):
        bg = np.ascontiguousarray(background_color, np.float32).reshape(4)
        _check(self.lib.nrf_rb_tonemap(self.h, C.c_float(exposure), _fptr(bg), output_color_space, C.c_void_p(stream or 0)))

    def present(self, exposure, background_color, output_color_space, rgba8_ptr=None, stream=None):
        """accumulate() + tonemap() in one pass over the planes (nrf_rb_present); rgba8_ptr: optional device uint32 [h][w]."""
        bg = np.ascontiguousarray(background_color, np.float32).reshape(4)
        _check(self.lib.nrf_rb_present(self.h, C.c_float(exposure), _fptr(bg), output_color_space, C.c_void_p(rgba8_ptr or 0),
                                       C.c_void_p(stream or 0)))

    def overlay_depth(self, alpha, depth_ptr, depth_scale, image_width, image_height, fov_axis=0, zoom=1.0,
                      screen_center=(0.5, 0.5), stream=None):
        ctr = np.ascontiguousarray(screen_center, np.float32).reshape(2)
        _check(self.lib.nrf_rb_overlay_depth(self.h, C.c_float(alpha), C.c_void_p(depth_ptr), C.c_float(depth_scale), image_width,
                                             image_height, fov_axis, C.c_float(zoom), _fptr(ctr), C.c_void_p(stream or 0)))

    def host_to_accumulate_buffer(self, rgb_u8):
        rgb = np.ascontiguousarray(rgb_u8, np.uint8)
        _check(self.lib.nrf_rb_host_to_accumulate_buffer(self.h, rgb.ctypes.data, rgb.size // 3))

    def read(self):
        acc = np.empty((self.height, self.width, 4), np.float32)
        sur = np.empty((self.height, self.width, 4), np.float32)
        _check(self.lib.nrf_rb_read(self.h, acc.ctypes.data, sur.ctypes.data))
        return acc, sur


# --------------------------------------------------------------------------
# tile partition of one frame over ranks (host-side mirror of render_kernel's mapping and of
# untile_kernel): a strip = 4 horizontally adjacent 8x8 tiles; strip id = ty*strips_x + tx//4 belongs to
# rank strip % world as local strip strip // world; local tile = 4*local strip + tx % 4
# --------------------------------------------------------------------------
def shard_tile_ids(width: int, height: int, rank: int, world: int):
    """(tx, ty) of this rank's local tiles, in local order; padding tiles of a ragged strip included."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    strips_x = (tiles_x + 3) // 4
    out = []
    for strip in range(rank, strips_x * tiles_y, world):
        for j in range(4):
            out.append(((strip % strips_x) * 4 + j, strip // strips_x))
    return out


def untile_numpy(gathered: np.ndarray, width: int, height: int):
    """gathered [world][tiles_per_shard*64][C] (every rank's tile-major shard) -> [H][W][C]."""
    world = gathered.shape[0]
    strips_x = ((width + 7) // 8 + 3) // 4
    ys, xs = np.mgrid[0:height, 0:width]
    tx, ty = xs // 8, ys // 8
    strip = ty * strips_x + tx // 4
    lane = (ys % 8) * 8 + xs % 8
    return gathered[strip % world, ((strip // world) * 4 + tx % 4) * 64 + lane]
