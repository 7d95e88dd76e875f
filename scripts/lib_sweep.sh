#!/bin/bash
# bench.py with several library builds, alternated on ONE box.  usage: scripts/lib_sweep.sh "a.so b.so" [reps]
LIBS=$1; REPS=${2:-2}
mkdir -p gpurun_out
: > gpurun_out/lib_sweep.txt
for rep in $(seq $REPS); do
  for lib in $LIBS; do
    python3 bench.py --no-cpu-baseline --steps 20 --lib $lib 2>> gpurun_out/lib_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'Msamples_s', d['value'], 'ms_per_step', d['ms_per_step'], 'single_view_ms', d['single_view_ms'], 'samples_per_frame', d['config']['samples_per_frame'])" >> gpurun_out/lib_sweep.txt
  done
done
cat gpurun_out/lib_sweep.txt
