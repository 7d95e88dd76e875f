#!/usr/bin/env python3
"""Per-launch durations of render_kernel from a rocprofv3 --kernel-trace CSV, split the way bench.py
runs them: warm-up launches, the timed region (overlapping launches: steps in flight), then the
untimed replays (8 single-view launches for the sample counts, up to 4 whole steps one at a time).
Usage: scripts/trace_summary.py <kernel_trace.csv> <warmup> <steps> [views_per_step]"""
import csv
import statistics
import sys

path, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
views = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = [r for r in csv.DictReader(open(path)) if "render_kernel" in r["Kernel_Name"] or "render_persistent_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
timed, single, rest = dur[warmup:warmup + steps], dur[warmup + steps:warmup + steps + 8], dur[warmup + steps + 8:]
n_iso = 0  # bench.py replays each distinct step composition once, alone (usually one); what follows are the extras' own launches
while n_iso < min(len(rest), steps) and timed and rest[n_iso] > 0.8 * statistics.mean(timed) and (n_iso == 0 or len(rest) <= steps):
    n_iso += 1
iso, extras = rest[:n_iso], rest[n_iso:]
span = (int(rows[warmup + steps - 1]["End_Timestamp"]) - int(rows[warmup]["Start_Timestamp"])) / 1e6
print(f"render_kernel launches: {len(dur)} (warm-up {warmup}, timed {len(timed)} of {views} views each, "
      f"single-view replays {len(single)}, isolated step replays {len(iso)}, launches of the extras (api / fast_interp legs) {len(extras)})")
print(f"  all launches      avg {statistics.mean(dur):.4f} ms   (what --stats reports)")
print(f"  timed region      avg {statistics.mean(timed):.4f} ms per launch, {span / len(timed) / views:.4f} ms per frame, "
      f"{statistics.mean(timed) * len(timed) / span:.2f} launches in flight")
if single:
    print(f"  one view alone    avg {statistics.mean(single):.4f} ms per launch")
if iso:
    print(f"  one step alone    avg {statistics.mean(iso):.4f} ms per launch ({statistics.mean(iso) / views:.4f} ms per frame)")
