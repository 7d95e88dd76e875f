#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one row per dispatch and counter) per kernel.
Usage: scripts/pmc_summary.py gpurun_out/prof_<tag>"""
import collections
import csv
import glob
import statistics
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "nrf::" not in k:
            continue
        k = f"{k} grid={r['Grid_Size']}"  # launches of different sizes (8-view steps, single-view replays) apart
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        meta[k] = dict(grid=r["Grid_Size"], wg=r["Workgroup_Size"], lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"],
                       agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"])
for k, v in agg.items():
    print(k, meta[k])
    for c, vals in sorted(v.items()):
        if c.startswith("_"):
            continue
        print(f"   {c:32s} n={len(vals):3d} mean={statistics.mean(vals):.6g}  kernel_ms={statistics.mean(v['_dur_' + c]) / 1e6:.3f}")
