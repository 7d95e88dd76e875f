#!/bin/bash
# Same-box A/B of two builds of the library: device time of the bench launch (16 views of 1920x1080), alternated 3 times.
# usage: scripts/lib_ab.sh <out> <libA.so> <libB.so>
set -eo pipefail
OUT=$1; shift
mkdir -p gpurun_out
: > "$OUT"
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib rep=$rep mean_ms min_ms n_samples n_evals: " >> "$OUT"
    python3 scripts/launch_ms.py "$lib" 10 >> "$OUT"
  done
done
cat "$OUT"
