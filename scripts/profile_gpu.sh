#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and PMC passes of bench.py.
# Usage: scripts/profile_gpu.sh <tag>   -> writes gpurun_out/prof_<tag>/*
set -eo pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 8 --warmup 2 --no-extras"
OLDPWD=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
python3 "$OLDPWD/scripts/trace_summary.py" "$OUT"/stats/*/*_kernel_trace.csv 2 8 16 > "$OUT/trace_summary.txt" 2>&1 || true
# PMC passes: counters in their own runs (no trace flags), a few per pass
pass() { n=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$n" -- $BENCH > "$OUT/pmc_$n.log" 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq3 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
# VALU instruction classes: fp32 add / mul / fma and plain int32 issue in 2 cycles per wave64 instruction on gfx950,
# conversions, packed fp16 and v_fma_mix in 4, transcendentals in 8 (scripts/issue_rate) -- the cycle-weighted VALU load
pass sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 || true
pass sq5 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_TRANS_F16 || true
pass sq6 SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_IOPS || true
# TA / TCP blocks take two counters per pass (more: "exceeds the capabilities of the hardware", and rocprofv3 hangs)
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || true
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum || true
pass tcp1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum || true
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || true
pass tcp3 TCP_TA_TCP_STATE_READ_sum || true
# counter-side MFMA utilisation: the render kernel (bench.py) and the fused-MLP stage kernel alone (scripts/mlp_steady.py)
pass mfma SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE || true
MLP="python3 $OLDPWD/scripts/mlp_steady.py"
mlp_pass() { n=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_mlp_$n" -- $MLP > "$OUT/pmc_mlp_$n.log" 2>&1; }
mlp_pass a SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || true
mlp_pass b SQ_INSTS_VALU_MFMA_MOPS_F16 || true
echo profile done
