#!/usr/bin/env python3
"""Per-launch durations of render_kernel from a rocprofv3 --kernel-trace CSV, split the way bench.py
runs them: warm-up launches, the timed region (overlapping launches, frames in flight) and the
isolated replays at the end.  Usage: scripts/trace_summary.py <kernel_trace.csv> <warmup> <steps>"""
import csv
import statistics
import sys

path, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if "render_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
timed, iso = dur[warmup:warmup + steps], dur[warmup + steps:]
span = (int(rows[warmup + steps - 1]["End_Timestamp"]) - int(rows[warmup]["Start_Timestamp"])) / 1e6
print(f"render_kernel launches: {len(dur)} (warm-up {warmup}, timed {len(timed)}, isolated replays {len(iso)})")
print(f"  all launches      avg {statistics.mean(dur):.4f} ms   (what --stats reports)")
print(f"  timed region      avg {statistics.mean(timed):.4f} ms per launch, {span / len(timed):.4f} ms per frame, "
      f"{statistics.mean(timed) * len(timed) / span:.2f} launches in flight")
if iso:
    print(f"  isolated replays  avg {statistics.mean(iso):.4f} ms per launch")
