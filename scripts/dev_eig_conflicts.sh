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
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f[0])):
        if row["Kernel_Name"].startswith("bbo::cma_eigen("):
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    ids = sorted(per)[-4:]          # the four launches made under the diagnostic bits
    acc = collections.defaultdict(float)
    for i in ids:
        for c, v in per[i].items():
            acc[c] += v / len(ids)
    print("bits %6d: " % dbg + "  ".join("%s %.4g" % (c, v) for c, v in sorted(acc.items())))
PY
