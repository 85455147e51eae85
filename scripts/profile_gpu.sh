#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and the two PMC passes for one
# bench workload; summaries land in gpurun_out/prof_<tag>/ and are copied into profiles/
# by hand afterwards.   usage: scripts/profile_gpu.sh <workload> <populations> <tag>
set -e
WL=$1; P=$2; TAG=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --workload $WL --populations $P --steps 20 --warmup 3 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/bench_fetch.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/bench_write.json
find $OUT -name "*.csv" | head -20
