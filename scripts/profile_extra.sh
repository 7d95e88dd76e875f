#!/bin/bash
# Only the TA / TCP passes of profile_gpu.sh (when the rest is already collected).
set -eo pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline"
cd /tmp
pass() { n=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$n" -- $BENCH > "$OUT/pmc_$n.log" 2>&1; echo "pass $n rc=$?"; }
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || true
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum || true
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum || true
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || true
pass tcp3 TCP_TA_TCP_STATE_READ_sum || true
echo extra done
