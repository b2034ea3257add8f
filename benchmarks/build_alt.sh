#!/bin/bash
# Experiment builds: the library with the WKV7 launchers (wkv7_capi.hip, wkv7_profile.hip) compiled under extra flags and WITH the
# experiment variants of benchmarks/experiments/ (wkv7_bwd_v7.h, role-skip builds, the tail on the J waves), into
# benchmarks/_alt/lib_<name>.so (git-ignored; selected at run time with VRWKV_HIP_LIB=...).   bash benchmarks/build_alt.sh <name> <flags...>
set -e
R=$(cd $(dirname $0)/.. && pwd); NAME=$1; shift
python -m visualrwkv_amd.build > /dev/null
mkdir -p $R/benchmarks/_alt
OBJS=""
for src in wkv7_capi wkv7_profile; do
  OBJ=$R/benchmarks/_alt/${src}_$NAME.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/visualrwkv_amd/csrc -I $R/include -I $R/benchmarks/experiments -Wno-unused-result -Wno-inline-asm \
        -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DVRWKV_V6_EXPERIMENTS "$@" -c $R/visualrwkv_amd/csrc/$src.hip -o $OBJ &
  OBJS="$OBJS $OBJ"
done
wait
OTHERS=$(ls $R/visualrwkv_amd/_build/*.o | grep -v "wkv7_capi\|wkv7_profile")
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS $OTHERS -o $R/benchmarks/_alt/lib_$NAME.so
rm -f $OBJS
echo $R/benchmarks/_alt/lib_$NAME.so
