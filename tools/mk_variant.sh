#!/bin/bash
# Development aid (no GPU): build variants/libnmpc_<name>.so = the column kernel compiled with extra switches + the other objects of the last
# full build (nmpc_amd/build/*.o).  For tools/ab_bench.sh.   bash tools/mk_variant.sh <name> [-DNMPC_...=.. ...]   (ONLY_M=6 by default: seconds)
# SRC=nmpc_lidar.hip builds a variant of the LIDAR kernel instead (no ONLY_M).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$(ls -d $ROOT/nonlinear*_amd)
mkdir -p $ROOT/variants
SRC=${SRC:-nmpc_solve_col.hip}
OBJ=$ROOT/variants/obj_$NAME.o
ONLY=${ONLY_M:-6}
if [ "$SRC" != "nmpc_solve_col.hip" ]; then ONLY=all; fi
DEFS="-DNMPC_SRC_HASH=\"variant_$NAME\""
if [ "$ONLY" != "all" ]; then DEFS="$DEFS -DNMPC_COL_ONLY_M=$ONLY"; fi
if [ -n "$PROFILE" ]; then DEFS="$DEFS -DNMPC_PROFILE"; fi
(cd $PKG/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DEFS "$@" -mllvm -disable-machine-licm ${SAVE_TEMPS:+-save-temps=obj} -c $SRC -o $OBJ)
OTHERS=$(ls $PKG/build/*.o | grep -v "/${SRC%.hip}\(_p[0-9]\)\?\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/libnmpc_$NAME.so $OBJ $OTHERS
echo "built variants/libnmpc_$NAME.so"
