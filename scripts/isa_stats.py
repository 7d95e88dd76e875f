#!/usr/bin/env python3
"""Static instruction statistics of one kernel in the `make asm` listing.

usage: scripts/isa_stats.py nerf-cuda_amd/build/nrf_kernels.s 'render_kernelILb0ELb1ELb1E' [--blocks]

Prints the instruction-class histogram of the function whose mangled name contains the pattern and,
with --blocks, one line per basic block (label, #VALU, #SALU, #VMEM, #LDS, #MFMA, total) so the
hot loops can be costed against the PMC counts (SQ_INSTS_VALU etc.).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "WAIT"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "OTHER"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") or ": ;" in l:
            name = l.split(":")[0]
            if pat in name and name.startswith("_Z") and start is None:
                start = i + 1
                print("function", name)
        if start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    hist = collections.Counter()
    ops = collections.Counter()
    cur, cur_cnt, out = "entry", collections.Counter(), []
    for l in lines[start:end]:
        s = l.strip()
        if not s or s.startswith((";", "//")):
            continue
        if re.match(r"^\.?[A-Za-z_0-9$.]+:", s):
            out.append((cur, cur_cnt))
            cur, cur_cnt = s.split(":")[0], collections.Counter()
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        hist[c] += 1
        ops[op] += 1
        cur_cnt[c] += 1
    out.append((cur, cur_cnt))
    print("total", sum(hist.values()), dict(hist))
    for k, v in ops.most_common(45):
        print(f"  {k:28s} {v}")
    if blocks:
        print("block                 VALU  SALU  VMEM   LDS  MFMA  total")
        for name, c in out:
            t = sum(c.values())
            if t >= 4:
                print(f"{name:20s} {c['VALU']:5d} {c['SALU']:5d} {c['VMEM']:5d} {c['LDS']:5d} {c['MFMA']:5d} {t:6d}")


if __name__ == "__main__":
    main()
