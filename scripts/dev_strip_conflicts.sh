#!/bin/bash
# dev (GPU box): the sampler's LDS bank conflicts with the ziggurat's strip reads as they are and
# with a diagnostic build whose strip reads cannot conflict (-DBBO_DIAG_STRIP_NOCONFLICT, wrong
# normals, timing / counters only): HIP-event kernel times + SQ_LDS_BANK_CONFLICT of both.
# usage: scripts/dev_strip_conflicts.sh <variant .so>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$ROOT/$1
OUT=$ROOT/gpurun_out/strip_conflicts
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for tag in base noconf; do
    if [ $tag = noconf ]; then export BBO_LIB=$VAR; else unset BBO_LIB; fi
    timeout -k 10 300 python3 $ROOT/scripts/dev_kernel_times.py 128 4096 256 40 > $OUT/times_$tag.txt 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE \
        --output-format csv -d $OUT/$tag -o sq -- python3 $ROOT/scripts/dev_kernel_times.py 128 4096 256 10 > $OUT/prof_$tag.txt 2> $OUT/prof_$tag.err || exit 1
done
python3 - <<PY
import csv, glob, collections
for tag in ("base", "noconf"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
    for k in acc:
        if "sample_eval128" in k or "gram128s" in k:
            print(tag, k[:40], {c: "%.4g" % (v / max(cnt[k], 1)) for c, v in acc[k].items()}, "launches", cnt[k])
PY
cat $OUT/times_base.txt $OUT/times_noconf.txt
