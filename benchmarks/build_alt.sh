#!/bin/bash
# Experiment builds: the library with wkv7_capi.hip compiled under extra flags, into benchmarks/_alt/lib_<name>.so
# (git-ignored; selected at run time with VRWKV_HIP_LIB=...).   bash benchmarks/build_alt.sh <name> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd); NAME=$1; shift
python -m visualrwkv_amd.build > /dev/null
mkdir -p $R/benchmarks/_alt
OBJ=$R/benchmarks/_alt/wkv7_capi_$NAME.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/visualrwkv_amd/csrc -I $R/include -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
      -fno-slp-vectorize "$@" -c $R/visualrwkv_amd/csrc/wkv7_capi.hip -o $OBJ
OTHERS=$(ls $R/visualrwkv_amd/_build/*.o | grep -v wkv7_capi)
hipcc --offload-arch=gfx950 -fPIC -shared $OBJ $OTHERS -o $R/benchmarks/_alt/lib_$NAME.so
rm -f $OBJ
echo $R/benchmarks/_alt/lib_$NAME.so
