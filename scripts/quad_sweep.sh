#!/bin/bash
# Same-box sweep of the cell-major quad copies: device time of the bench launch (16 views of 1920x1080) per setting of one
# environment variable (NRF_QUAD_BUDGET_MB: MB of gather copies; NRF_QUAD_LEVELS: leading levels that may get one), alternated 3 times.
# usage: scripts/quad_sweep.sh <out> <VAR> <values...>
set -eo pipefail
OUT=$1; VAR=$2; shift 2
mkdir -p gpurun_out
: > "$OUT"
for rep in 1 2 3; do
  for q in "$@"; do
    echo -n "$VAR=$q rep=$rep mean_ms min_ms n_samples n_evals: " >> "$OUT"
    env "$VAR=$q" python3 scripts/launch_ms.py nerf-cuda_amd/libnerfhip.so 10 >> "$OUT"
  done
done
cat "$OUT"
