#!/bin/bash
# Run on the GPU box (through gpurun): the two PMC passes (FETCH_SIZE, WRITE_SIZE) of one bench
# workload; lands in gpurun_out/prof_<tag>_<WL>/{fetch,write}
# usage: scripts/profile_traffic_one.sh <tag> <workload> <populations>
TAG=$1; WL=$2; P=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --workload $WL --populations $P --steps 10 --warmup 3 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "$WL fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "$WL write rc=$?"
find $OUT -name "*kernel_trace.csv" -size +4M -delete
