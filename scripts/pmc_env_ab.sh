#!/bin/bash
# L1 / L2 / texture-addresser counters of bench.py's 16-view launches under two settings of one environment variable.
# usage: scripts/pmc_env_ab.sh VAR a b   -> gpurun_out/pmc_env/summary.txt
set -eo pipefail
VAR=$1; A=$2; B=$3
OUT=$PWD/gpurun_out/pmc_env
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline"
ROOT=$PWD
cd /tmp
for val in "$A" "$B"; do
  export "$VAR=$val"
  pass() { n=$1; shift; timeout -k 10 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$val/pmc_$n" -- $BENCH > "$OUT/${val}_$n.log" 2>&1; }
  pass l1 GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pass l2 TCC_HIT_sum TCC_MISS_sum || true
  pass fetch FETCH_SIZE || true
  pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum || true
  pass sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || true
done
cd "$ROOT"
python3 - "$OUT" "$A" "$B" > "$OUT/summary.txt" <<'PY'
import csv, glob, statistics, sys, collections
root, vals = sys.argv[1], sys.argv[2:]
for val in vals:
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{val}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "render" not in r["Kernel_Name"]:
                continue
            dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
            if dur < 5.0:  # the 16-view launches only
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_ms_" + r["Counter_Name"]].append(dur)
    print("==", val)
    for k in sorted(agg):
        if not k.startswith("_"):
            print(f"   {k:36s} n={len(agg[k]):3d} mean={statistics.mean(agg[k]):.6g}  kernel_ms={statistics.mean(agg['_ms_' + k]):.3f}")
PY
cat "$OUT/summary.txt"
