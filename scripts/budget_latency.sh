#!/bin/bash
set -eo pipefail
: > gpurun_out/budget_latency.txt
for b in 16 64 256 16 256; do
  NRF_MARCH_BUDGET=$b python3 bench.py --no-cpu-baseline --steps 48 --warmup 4 --views-per-step 1 --frames-in-flight 1 2>> gpurun_out/budget_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('single frame, budget', $b, 'ms_per_frame', d['ms_per_frame'])" >> gpurun_out/budget_latency.txt
done
cat gpurun_out/budget_latency.txt
