#!/bin/bash
# Development aid (no GPU): build variants/libnmpc_<name>.so = the column kernel compiled with extra switches + the other objects of the last
# full build (nmpc_amd/build/*.o).  For tools/ab_bench.sh.   bash tools/mk_variant.sh <name> [-DNMPC_...=.. ...]   (ONLY_M=6 by default: seconds)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$(ls -d $ROOT/nonlinear*_amd)
mkdir -p $ROOT/variants
OBJ=$ROOT/variants/col_$NAME.o
ONLY=${ONLY_M:-6}
DEFS="-DNMPC_SRC_HASH=\"variant_$NAME\""
if [ "$ONLY" != "all" ]; then DEFS="$DEFS -DNMPC_COL_ONLY_M=$ONLY"; fi
if [ -n "$PROFILE" ]; then DEFS="$DEFS -DNMPC_PROFILE"; fi
(cd $PKG/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEFS "$@" -mllvm -disable-machine-licm ${SAVE_TEMPS:+-save-temps=obj} -c nmpc_solve_col.hip -o $OBJ)
OTHERS=$(ls $PKG/build/*.o | grep -v nmpc_solve_col.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/libnmpc_$NAME.so $OBJ $OTHERS
echo "built variants/libnmpc_$NAME.so"
