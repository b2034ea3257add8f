#!/bin/bash
# Experiment build of SEVERAL sources of the library under the same extra flags, into benchmarks/_alt/lib_<name>.so (git-ignored;
# selected at run time with VRWKV_HIP_LIB=...).   bash benchmarks/build_alt_multi.sh <name> "<flags>" <source.hip> [<source.hip> ...]
set -e
R=$(cd $(dirname $0)/.. && pwd); NAME=$1; FLAGS=$2; shift; shift
python -c "from visualrwkv_amd import build; build.build()" > /dev/null 2>&1
mkdir -p $R/benchmarks/_alt
OBJS=""; OTHERS=$(ls $R/visualrwkv_amd/_build/*.o)
for SRC in "$@"; do
  OBJ=$R/benchmarks/_alt/${SRC%.hip}_$NAME.o
  EXTRA=""; [ "$SRC" = "wkv7_capi.hip" ] && EXTRA="-fno-slp-vectorize"; [ "$SRC" = "attention.hip" ] && EXTRA="-fno-honor-nans"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/visualrwkv_amd/csrc -I $R/include -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
        $EXTRA $FLAGS -c $R/visualrwkv_amd/csrc/$SRC -o $OBJ &
  OBJS="$OBJS $OBJ"; OTHERS=$(echo "$OTHERS" | grep -v "/${SRC}\.")
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS $OTHERS -o $R/benchmarks/_alt/lib_$NAME.so
rm -f $OBJS
echo $R/benchmarks/_alt/lib_$NAME.so
