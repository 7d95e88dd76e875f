#!/bin/bash
# bench.py under several settings of one environment variable, alternated on ONE box.
# usage: scripts/env_sweep.sh VAR "v1 v2 v3" [reps] [extra bench flags]  -> gpurun_out/env_sweep.txt
VAR=$1; VALS=$2; REPS=${3:-2}; shift 3 || shift $#
mkdir -p gpurun_out
: > gpurun_out/env_sweep.txt
for rep in $(seq $REPS); do
  for val in $VALS; do
    env "$VAR=$val" python3 bench.py --no-cpu-baseline --steps 20 "$@" 2>> gpurun_out/env_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$val', 'Msamples_s', d['value'], 'ms_per_step', d['ms_per_step'], 'single_view_ms', d['single_view_ms'])" >> gpurun_out/env_sweep.txt
  done
done
cat gpurun_out/env_sweep.txt
