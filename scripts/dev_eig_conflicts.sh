#!/bin/bash
# dev (GPU box): where the n = 128 eigensolver's LDS bank-conflict cycles come from -- the counter
# for the whole kernel and for runs that stop early (diagnostic bits: 12 = no leaves, no merges:
# reduction + reflector stage; 4 = + leaves; 8192 / 16384 = + the first / first two merge levels)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/eig_conflicts
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for dbg in 12 4 8192 16384 0; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE \
        --output-format csv -d $OUT/d$dbg -o sq -- python3 $ROOT/scripts/dev_eig_phase_only.py 128 256 4 $dbg > $OUT/run_$dbg.txt 2> $OUT/run_$dbg.err || exit 1
done
python3 - <<PY
import csv, glob, collections
for dbg in (12, 4, 8192, 16384, 0):
    f = glob.glob("$OUT/d%d/**/*counter_collection.csv" % dbg, recursive=True)
    acc = collections.defaultdict(float); n = 0
    for row in csv.DictReader(open(f[0])):
        if row["Kernel_Name"].startswith("bbo::cma_eigen("):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            n += row["Counter_Name"] == "SQ_INSTS_LDS"
    print("dbg %6d: per launch of 256 matrices:" % dbg, {c: "%.4g" % (v / max(n, 1)) for c, v in sorted(acc.items())}, "launches", n)
PY
