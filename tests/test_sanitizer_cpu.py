"""CPU sanitizer build (SURVEY 5 "Race detection / sanitizers"; the reference has none): the host code that parses
untrusted bytes -- msgpack_lite.h + NerfRender::load_snapshot (both snapshot layouts), json_lite.h + load_camera_path,
png_lite.h -- compiled with -fsanitize=address,undefined (`make -C nerf-cuda_amd/host asan`, host only: no device
context) and run over (a) well-formed files of every layout, (b) the hostile inputs of tests/test_host_cpp.py, (c) a
seeded structural fuzz of 10 000 cases.  A sanitizer report aborts the harness: any non-zero exit status fails."""
import os
import subprocess
from pathlib import Path

import msgpack
import numpy as np
import pytest

import models
import nerfhip as nh
import synthetic as syn

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "nerf-cuda_amd" / "host"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def harness():
    r = subprocess.run(["make", "-C", str(HOST), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return HOST / "parser_fuzz_asan"


def _run(harness, *args):
    r = subprocess.run([str(harness), *map(str, args)], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0, (args, r.returncode, (r.stdout + r.stderr)[-2000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    return r.stdout


def test_well_formed_files_of_every_layout(harness, tmp_path):
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    files = []
    for name, kw in (("arrays", {}), ("half", {"binary": "__half"}), ("float", {"binary": "float"})):
        f = tmp_path / f"{name}.msgpack"
        syn.write_snapshot(f, cfg, keep[0], keep[1], **kw)
        files.append(f)
    f = tmp_path / "nogrid.msgpack"
    syn.write_snapshot(f, cfg, keep[0], None)
    files.append(f)
    for aabb_scale in (2, 8):  # instant-ngp's own layout: bound = aabb_scale / 2, Morton-ordered cascades
        cascade = aabb_scale.bit_length() - 1
        pls = nh.default_per_level_scale(float(aabb_scale), 16, 16)
        d2, k2, c2 = models.build_model(log2_hashmap_size=12, H=32, bound=aabb_scale / 2.0, cascade=cascade, per_level_scale=pls)
        f = tmp_path / f"ngp_{aabb_scale}.msgpack"
        syn.write_ngp_snapshot(f, c2, k2[0], k2[1], aabb_scale)
        files.append(f)
    t = tmp_path / "transforms.json"
    syn.write_transforms_json(t, [syn.orbit_pose(10.0 * i, 20.0) for i in range(5)], 800, 800)
    files.append(t)
    for f in files:
        assert "accepted" in _run(harness, "file", f), f


def test_hostile_inputs_are_rejected_without_a_report(harness, tmp_path):
    cases = {
        "huge_array.msgpack": b"\x81\xa8snapshot\xdd\xff\xff\xff\xff\x01\x02",
        "huge_map.msgpack": b"\x81\xa8snapshot\xdf\xff\xff\xff\xff",
        "deep.msgpack": b"\x81\xa8snapshot" + b"\x91" * 100000 + b"\x00",
        "zero_F.msgpack": msgpack.packb({"encoding": {"otype": "HashGrid", "n_features_per_level": 0, "n_features": 32},
                                         "snapshot": {"aabb": [-1, -1, -1, 1, 1, 1], "density_grid_size": 1, "density_grid": [0.0],
                                                      "params": [0.0]}}, use_single_float=True),
        "odd_blob.msgpack": msgpack.packb({"encoding": {"otype": "HashGrid"},
                                           "snapshot": {"aabb": [-1, -1, -1, 1, 1, 1], "params_binary": b"\x00\x01\x02",
                                                        "density_grid_binary": b"\x00" * 7}}, use_bin_type=True),
        "deep.json": b"[" * 100000,
        "nan.json": b'{"camera_angle_x": nan, "frames": []}',
        "inf.json": b'{"camera_angle_x": 1e999, "w": 8, "h": 8, "frames": []}',
        "hex.json": b'{"camera_angle_x": 0x1p3, "w": 8, "h": 8, "frames": []}',
        "long_number.json": b'{"camera_angle_x": 0.' + b"1" * 5000 + b', "w": 8, "h": 8, "frames": []}',
        "bad_escape.json": b'{"a": "\\u12G4"}',
        "truncated.json": b'{"camera_angle_x": 0.69, "frames": [{"transform_matrix": [[1, 0, 0',
    }
    for name, blob in cases.items():
        f = tmp_path / name
        f.write_bytes(blob)
        assert "rejected" in _run(harness, "file", f), name


def test_seeded_structural_fuzz(harness, tmp_path):
    total = 0
    for seed, n in ((1, 5000), (20240607, 5000)):
        out = _run(harness, "fuzz", seed, n, tmp_path)
        line = [ln for ln in out.splitlines() if ln.startswith("fuzz:")][-1]
        total += int(line.split()[1])
    assert total == 10000
