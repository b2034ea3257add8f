#!/bin/bash
# Experiment builds of ONE source of the library under extra flags, into benchmarks/_alt/lib_<name>.so (git-ignored; selected at run
# time with VRWKV_HIP_LIB=...).   bash benchmarks/build_alt_src.sh <name> <source.hip> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd); NAME=$1; SRC=$2; shift; shift
python -c "from visualrwkv_amd import build; build.build()" > /dev/null 2>&1
mkdir -p $R/benchmarks/_alt
OBJ=$R/benchmarks/_alt/${SRC%.hip}_$NAME.o
EXTRA=""; [ "$SRC" = "wkv7_capi.hip" ] && EXTRA="-fno-slp-vectorize"; [ "$SRC" = "attention.hip" ] && EXTRA="-fno-honor-nans"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/visualrwkv_amd/csrc -I $R/include -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
      $EXTRA "$@" -c $R/visualrwkv_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls $R/visualrwkv_amd/_build/*.o | grep -v "/${SRC}\.")
hipcc --offload-arch=gfx950 -fPIC -shared $OBJ $OTHERS -o $R/benchmarks/_alt/lib_$NAME.so
rm -f $OBJ
echo $R/benchmarks/_alt/lib_$NAME.so
