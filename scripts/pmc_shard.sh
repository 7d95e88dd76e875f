#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_shard
mkdir -p $OUT
cd /tmp
for cfg in "1 0 16" "8 3 128"; do
  set -- $cfg
  tag=N$1
  timeout -k 10 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$tag/pmc_fetch -- python3 $ROOT/scripts/shard_bench.py $1 $2 $3 > $OUT/${tag}_fetch.log 2>&1
  timeout -k 10 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$tag/pmc_l2 -- python3 $ROOT/scripts/shard_bench.py $1 $2 $3 > $OUT/${tag}_l2.log 2>&1
  timeout -k 10 180 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/$tag/pmc_l1 -- python3 $ROOT/scripts/shard_bench.py $1 $2 $3 > $OUT/${tag}_l1.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, statistics
for tag in ("N1", "N8"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/pmc_shard/{tag}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "render" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(tag, {k: (round(v / n[k], 1), n[k]) for k, v in agg.items()})
PY
