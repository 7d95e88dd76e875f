#!/usr/bin/env python3
"""In-situ marginal cost of a vector instruction in the render kernel's two big phases (does the kernel
wait for the VALU issue port?).  Diagnostic builds (make -C nerf-cuda_amd diag; never shipped) add a KNOWN number of
instructions whose results nobody reads -- or drop known ones -- and the launch time of bench.py's step is read for each,
alternating with the shipped build on the same box:
    interp1 / interp2   +1 / +2 v_cvt_pk_f16_f32 (a 4.1-cycle instruction) per corner of the trilinear interpolation
    nocvt               the v_cvt_pk_f16_f32 + v_pk_add_f16 of every corner replaced by two v_add_f32 (wrong values)
    march8 / march16    +8 / +16 v_mul_f32 (2.25 cycles) per cell trip of the march
The number of wave-instructions added follows from the kernel's own counters: 32 corner slots per evaluated 16-sample
MFMA tile and lane (4 levels x 8 corners), and the wave-level trip iterations of the march (diagnostic counter of the
`prof` build).  Output: ms per 10^9 added wave-instructions, and the same as cycles per instruction and SIMD
(1024 SIMDs, clock from the kernel's own time)."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "nerf-cuda_amd"
CLOCK_GHZ = 2.25  # what the chip holds under this load (profiles/r02/issue_rate.txt)


def run(lib, reps=10):
    out = subprocess.run([sys.executable, str(ROOT / "scripts" / "launch_ms.py"), str(PKG / lib), str(reps)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [float(x) for x in out.stdout.strip().splitlines()[-1].split()]


prof = run("libnerfhip_prof.so", 3)
wave_trips = prof[4]
variants = ["libnerfhip_diag_interp1.so", "libnerfhip_diag_interp2.so", "libnerfhip_diag_nocvt.so", "libnerfhip_diag_march8.so",
            "libnerfhip_diag_march16.so"]
base_ms, results = [], {}
for v in variants:  # base, variant, base, variant ...: drift of the box cancels
    b = run("libnerfhip.so")
    r = run(v)
    base_ms.append(b[0])
    results[v] = (r[0], b[0], b[3])
base = sum(base_ms) / len(base_ms)
tiles = results[variants[0]][2] / 16.0
added = {
    "libnerfhip_diag_interp1.so": (32 * 1 * tiles, "v_cvt_pk_f16_f32 added"),
    "libnerfhip_diag_interp2.so": (32 * 2 * tiles, "v_cvt_pk_f16_f32 added"),
    "libnerfhip_diag_nocvt.so": (-32 * 2 * tiles, "v_cvt_pk_f16_f32 / v_pk_add_f16 replaced by v_add_f32 (each: 4.1 -> 2.25 nominal cycles)"),
    "libnerfhip_diag_march8.so": (8 * wave_trips, "v_mul_f32 added"),
    "libnerfhip_diag_march16.so": (16 * wave_trips, "v_mul_f32 added"),
}
out = {"launch": "16 views of 1920x1080 (bench.py's step), float planes", "base_ms": round(base, 4), "base_ms_runs": [round(x, 4) for x in base_ms],
       "mfma_tiles_per_launch": int(tiles), "march_wave_trip_iterations_per_launch": int(wave_trips), "variants": {}}
for v in variants:
    ms, b, _ = results[v]
    n, what = added[v]
    d = ms - b
    out["variants"][v.replace("libnerfhip_diag_", "").replace(".so", "")] = {
        "what": what, "wave_instructions": int(n), "ms": round(ms, 4), "base_ms_same_pair": round(b, 4), "delta_ms": round(d, 4),
        "ms_per_1e9_wave_instructions": round(d / (n / 1e9), 4),
        "cycles_per_instruction_and_simd": round(d * 1e-3 * CLOCK_GHZ * 1e9 * 1024 / n, 3),
    }
print(json.dumps(out, indent=1))
