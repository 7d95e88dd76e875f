#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
Usage: make -C nerf-cuda_amd asm 2>&1 | python3 scripts/kernel_resources.py"""
import re
import subprocess
import sys

cur, rows = None, {}
for ln in sys.stdin:
    m = re.search(r"remark: (?:[^:\s]+:\d+:\d+: )?(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:48s} VGPR {v.get('VGPRs'):>4s} AGPR {v.get('AGPRs'):>3s} scratch {v.get('ScratchSize [bytes/lane]'):>4s} "
          f"waves/SIMD {v.get('Occupancy [waves/SIMD]')} spills s{v.get('SGPRs Spill')} v{v.get('VGPRs Spill')}")
