#!/bin/bash
# dev (build container): a VARIANT of libbbopt_hip.so with extra compiler flags for the translation
# unit that holds the CMA / eigensolver kernels -- for A/B timing on the GPU box through BBO_LIB
# (scripts/dev_kernel_times.py, dev_plain_run.py).  Output: scripts/_variants/lib<name>.so
# (git-ignored; travels with gpurun).     usage: scripts/build_variant.sh <name> "<flags>" [tu.hip]
NAME=$1; FLAGS=$2; TU=${3:-bbo_cma.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/bboptpy_amd/csrc
OUT=$ROOT/scripts/_variants
mkdir -p $OUT/$NAME
make -s -C $SRC >/dev/null || exit 1
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $FLAGS \
    -c $SRC/$TU -o $OUT/$NAME/${TU%.hip}.o || exit 1
OBJS=""
for o in $SRC/_build/*.o; do
  b=$(basename $o)
  if [ "$b" = "${TU%.hip}.o" ]; then OBJS="$OBJS $OUT/$NAME/$b"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OBJS && echo "built $OUT/lib$NAME.so"
