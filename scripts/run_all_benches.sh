#!/bin/bash
# Run on the GPU box (through gpurun): one bench.py line per workload into gpurun_out/<tag>_<WL>.json
# usage: scripts/run_all_benches.sh <tag> [workloads...]
TAG=$1; shift
WLS=${@:-"M C1 C3 C2 JADE SANSDE SEP CSO CCPSO C4 C5"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for WL in $WLS; do
  STEPS=100; [ "$WL" = "C4" ] && STEPS=20; [ "$WL" = "C5" ] && STEPS=2
  timeout -k 10 400 python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 10 \
      > $ROOT/gpurun_out/${TAG}_$WL.json 2> $ROOT/gpurun_out/${TAG}_$WL.err
  echo "$WL rc=$? $(head -c 180 $ROOT/gpurun_out/${TAG}_$WL.json)"
done
