#!/bin/bash
# usage: scripts/bench_lib.sh <lib.so> [bench args]  -- run bench.py against an alternative build (A/B tuning)
set -eo pipefail
LIB=$1; shift
cp nerf-cuda_amd/libnerfhip.so /tmp/libnerfhip_orig.so
cp -f "$LIB" /tmp/lib_ab.so; cp -f /tmp/lib_ab.so nerf-cuda_amd/libnerfhip.so
python bench.py --no-cpu-baseline "$@" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$LIB', d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"
cp -f /tmp/libnerfhip_orig.so nerf-cuda_amd/libnerfhip.so
