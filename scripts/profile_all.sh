#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats (+ the two PMC passes for the
# headline workload) of the bench workloads; everything lands in gpurun_out/prof_<tag>/.
# usage: scripts/profile_all.sh <tag> [workload ...]   (default: all nine)
TAG=$1
shift
ONLY=" $* "
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "M 256 pmc" "C3 256 pmc" "C2 256 pmc" "C1 4096 nopmc" "C4 1 nopmc" "C5I 1 nopmc" "SEP 64 pmc" "CSO 4 nopmc" "CCPSO 16 nopmc"; do
  set -- $spec
  WL=$1; P=$2; PMC=$3
  if [ "$ONLY" != "  " ] && [ "${ONLY#* $WL }" = "$ONLY" ]; then continue; fi
  OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
  mkdir -p $OUT
  export TMPDIR=/tmp
  cd /tmp
  ARGS="$ROOT/bench.py --workload $WL --populations $P --steps 20 --warmup 3 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
  [ "$WL" = "C4" ] && ARGS="$ROOT/bench.py --workload $WL --populations $P --steps 5 --warmup 2 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
  echo "$WL trace rc=$?"
  if [ "$PMC" = "pmc" ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
    echo "$WL fetch rc=$?"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
    echo "$WL write rc=$?"
  fi
  # keep only the summaries (the raw traces are large)
  find $OUT -name "*kernel_trace.csv" -size +4M -delete
done
find $ROOT/gpurun_out/prof_${TAG}_* -name "*.csv" | head -40
du -sh $ROOT/gpurun_out/prof_${TAG}_* | tail -10
