#!/usr/bin/env python3
"""For every packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of a kernel in an ISA listing:
which instruction last wrote each of its source VGPRs, how many instructions earlier, and of what kind (VALU, DPP,
transcendental, MFMA, LDS / VMEM return, v_readlane ...).  Used to look for the producer -> packed-consumer pair behind
the lanes-48..63 corruption of DESIGN.md "Packed-fp32 hazard" in the SLP-vectorised build (which is never shipped).
Usage: scripts/hazard/pk_f32_producers.py nerf-cuda_amd/build/nrf_kernels_slp.s 'render_kernel<false, true, 1>'"""
import collections
import re
import subprocess
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
kern, body = None, []
for ln in lines:
    m = re.match(r"^(_ZN3nrf\w+):", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        kern = name
        continue
    if kern and want in kern and ln.startswith("\t") and not ln.strip().startswith((".", ";")):
        body.append(ln.strip().split(";")[0].strip())


def regs(tok):
    """VGPR numbers named by one operand token (v5, v[4:5], ...)."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def kind(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith(("ds_", "buffer_", "global_", "flat_", "scratch_")): return "MEM"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op): return "TRANS"
    if "dpp" in op or op.startswith(("v_readlane", "v_readfirstlane", "v_permlane", "v_writelane")): return "LANE"
    if op.startswith("v_pk_") and op.endswith("_f32"): return "PKF32"
    if op.startswith("v_"): return "VALU"
    return "OTHER"


last = {}  # vgpr -> (index, text)
pairs = collections.Counter()
examples = {}
for i, ins in enumerate(body):
    parts = ins.replace(",", " ").split()
    op, ops = parts[0], parts[1:]
    if re.match(r"v_pk_(mul|add|fma)_f32", op):
        for tok in ops[1:]:
            for r in regs(tok):
                if r in last:
                    j, txt = last[r]
                    dist = i - j
                    between = [kind(b.split()[0]) for b in body[j + 1:i]]
                    key = (kind(txt.split()[0]) + (":dpp" if "dpp" in txt else ""), min(dist, 9))
                    pairs[key] += 1
                    if key not in examples or dist < examples[key][0]:
                        examples[key] = (dist, txt, ins, sum(1 for b in between if b != "OTHER"))
    # destination registers of this instruction (first operand; MEM loads and MFMA write their first operand as well)
    if op.startswith(("v_", "ds_read", "buffer_load", "global_load", "flat_load")) and ops:
        for r in regs(ops[0]):
            last[r] = (i, ins)
print(f"{want}: {sum(1 for b in body if re.match(r'v_pk_(mul|add|fma)_f32', b))} packed-fp32 instructions in {len(body)}")
print("producer kind, distance in instructions (9 = nine or more): count   [closest example: producer -> consumer, vector instructions in between]")
for (k, d), n in sorted(pairs.items()):
    dist, txt, ins, nb = examples[(k, d)]
    print(f"  {k:10s} d={d}: {n:4d}   [{txt}  ->  {ins}   ({nb} in between)]")
