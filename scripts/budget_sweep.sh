#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
: > gpurun_out/budget_sweep.txt
for b in 48 64 96 128 256 1024 48; do
  NRF_MARCH_BUDGET=$b python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>> gpurun_out/budget_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('budget', $b, 'ms_per_frame', d['ms_per_frame'], 'Msamples/s', d['value'], 'samples/frame', d['config']['samples_per_frame'])" >> gpurun_out/budget_sweep.txt
done
cat gpurun_out/budget_sweep.txt
