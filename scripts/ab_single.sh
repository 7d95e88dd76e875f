#!/bin/bash
# A/B of two builds for ONE frame per launch, one launch in flight (latency), alternated on one box
set -eo pipefail
: > gpurun_out/ab_single.txt
for rep in 1 2 3; do
  for lib in "$@"; do
    python3 bench.py --no-cpu-baseline --steps 48 --warmup 4 --views-per-step 1 --frames-in-flight 1 --lib "$lib" 2>> gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'single frame ms', d['ms_per_frame'])" >> gpurun_out/ab_single.txt
  done
done
cat gpurun_out/ab_single.txt
