#!/bin/bash
# A/B on ONE box (boxes differ by up to ~30 % in clocks): alternates bench.py runs of two library builds.
# usage: scripts/ab_bench.sh <libA.so> <libB.so> [extra bench flags]   -> gpurun_out/ab.txt
set -eo pipefail
A=$1; B=$2; shift 2
mkdir -p gpurun_out
: > gpurun_out/ab.txt
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    python3 bench.py --no-cpu-baseline --lib "$lib" "$@" 2>> gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'Msamples_s', d['value'], 'ms_per_step', d['ms_per_step'], 'isolated_ms', r['isolated_kernel_ms'], 'single_view_ms', d['single_view_ms'], 'mlp_tflops', d.get('mlp_kernel', {}).get('achieved'))" >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt
