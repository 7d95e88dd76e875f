#!/bin/bash
# A/B of two library builds on ONE box: alternates scripts/launch_ms.py (device time of bench.py's 16-view launch).
# usage: scripts/ab_launch.sh <libA.so> <libB.so> [pairs=3]
A=$1; B=$2; N=${3:-3}
for rep in $(seq 1 $N); do
  for lib in "$A" "$B"; do
    echo "$lib $(python3 scripts/launch_ms.py $lib 10 2>/dev/null | tail -1)"
  done
done
