#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of ONE bench workload;
# lands in gpurun_out/prof_<tag>_<WL>/trace   usage: scripts/profile_one.sh <tag> <workload> <populations> [steps]
TAG=$1; WL=$2; P=$3; STEPS=${4:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --workload $WL --populations $P --steps $STEPS --warmup 3 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "$WL trace rc=$?"
find $OUT -name "*kernel_trace.csv" -size +4M -delete
