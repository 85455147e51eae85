#!/bin/bash
# Run on the GPU box (through gpurun): SQ counter pass (MFMA busy cycles, instruction counts,
# wait buckets, LDS bank conflicts) for one bench workload; lands in gpurun_out/prof_<tag>_<WL>/sq/
# usage: scripts/profile_mfma.sh <tag> <workload> <populations>
TAG=$1; WL=$2; P=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$ROOT/bench.py --workload $WL --populations $P --steps 10 --warmup 3 --no-cpu-baseline --no-single --no-bipop --no-convergence --no-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/sq -o sq -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
echo "sq pass rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq2 -o sq2 -- python3 $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
echo "sq2 pass rc=$?"
ls -la $OUT/sq $OUT/sq2 | head -20
