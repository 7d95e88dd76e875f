#!/bin/bash
# usage: scripts/single_view_trace.sh <tag> <lib.so>...   -> gpurun_out/<tag>/single_view_trace.txt
set -eo pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
: > "$OUT/single_view_trace.txt"
for lib in "$@"; do
  name=$(basename $lib .so)
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/svt_$name" -- python3 $REPO/scripts/single_view_trace.py $REPO/$lib > "$OUT/svt_$name.log" 2>&1
  cd $REPO
  python3 - "$OUT"/svt_$name/*/*_kernel_trace.csv "$name" >> "$OUT/single_view_trace.txt" <<'PY'
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "nrf::" in r["Kernel_Name"]]
# the last 24 render launches are the single views; walk back from each to its planning kernels
idx = [i for i, r in enumerate(rows) if "render_persistent_kernel" in r["Kernel_Name"]][-24:]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
render, plan, span = [], {}, []
for i in idx:
    render.append(d(rows[i]))
    j = i - 1
    first = i
    while j >= 0 and "plan" in rows[j]["Kernel_Name"]:
        plan.setdefault(rows[j]["Kernel_Name"].split("(")[0], []).append(d(rows[j]))
        first = j
        j -= 1
    span.append((int(rows[i]["End_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e3)
print(f"{sys.argv[2]}: render kernel {statistics.mean(render):.1f} us (min {min(render):.1f}); first planning kernel's start to the render's end {statistics.mean(span):.1f} us; "
      + "; ".join(f"{k} {statistics.mean(v):.1f} us" for k, v in plan.items()))
PY
done
cat "$OUT/single_view_trace.txt"
