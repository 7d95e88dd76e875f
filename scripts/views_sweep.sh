#!/bin/bash
# ms per frame for (views per step, steps in flight) combinations on one box -> gpurun_out/views_sweep.txt
set -eo pipefail
mkdir -p gpurun_out
: > gpurun_out/views_sweep.txt
for cfg in "16 1" "32 1" "16 1" "32 1" "8 1" "1 1" "1 3"; do
  set -- $cfg
  python3 bench.py --no-cpu-baseline --views-per-step $1 --frames-in-flight $2 --steps $((192 / $1)) --warmup 2 2>> gpurun_out/views_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('views', $1, 'in_flight', $2, 'ms_per_frame', d['ms_per_frame'], 'Msamples/s', d['value'], 'launch_ms', r['kernel_ms'], 'frac', r['frac'])" >> gpurun_out/views_sweep.txt
done
cat gpurun_out/views_sweep.txt
