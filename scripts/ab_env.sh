#!/bin/bash
# A/B of one library under two environments on ONE box: alternates bench.py runs with VAR=a and VAR=b.
# usage: scripts/ab_env.sh VAR a b [extra bench flags]   -> gpurun_out/ab_env.txt
set -eo pipefail
VAR=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
: > gpurun_out/ab_env.txt
for rep in 1 2 3; do
  for val in "$A" "$B"; do
    env "$VAR=$val" python3 bench.py --no-cpu-baseline "$@" 2>> gpurun_out/ab_env.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$val', 'Msamples_s', d['value'], 'ms_per_step', d['ms_per_step'], 'single_view_ms', d['single_view_ms'], 'isolated_ms', r['isolated_kernel_ms'])" >> gpurun_out/ab_env.txt
  done
done
cat gpurun_out/ab_env.txt
