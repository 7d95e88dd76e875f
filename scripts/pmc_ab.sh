#!/bin/bash
# SQ_INSTS_VALU / SALU / busy cycles of render_kernel for two library builds (single-view launches).
set -eo pipefail
OUT=$PWD/gpurun_out/pmc_ab
rm -rf "$OUT"; mkdir -p "$OUT"; for lib in "$@"; do mkdir -p "$OUT/$(basename $lib .so)"; done
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$tag/pmc_a" -- python3 $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --views-per-step 1 --frames-in-flight 1 --lib $REPO/$lib > "$OUT/$tag.log" 2>&1 || true
  echo "== $tag"
  python3 $REPO/scripts/pmc_summary.py "$OUT/$tag" | grep -A6 render_kernel || true
done
