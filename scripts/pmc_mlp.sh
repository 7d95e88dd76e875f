#!/bin/bash
# PMC passes for mlp_forward_kernel only (bench.py's microbenchmark), short run.
set -eo pipefail
OUT=$PWD/gpurun_out/pmc_mlp
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --views-per-step 1"
cd /tmp
pass() { n=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$n" -- $BENCH > "$OUT/pmc_$n.log" 2>&1 || true; }
pass a SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
pass c FETCH_SIZE
pass d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $REPO
python3 scripts/pmc_summary.py "$OUT" | grep -A22 mlp_forward
