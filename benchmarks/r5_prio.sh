#!/bin/bash
# micro-benchmark A/B of library builds: bash benchmarks/r5_prio.sh <tag> "<variants>" <lib-glob>
OUT=gpurun_out/${1:-r5}_wkv7_libs_ab.jsonl; : > $OUT
for lib in $3; do
  echo -n "{\"lib\": \"$(basename $lib)\", \"ab\": [" >> $OUT
  VRWKV_HIP_LIB=$lib python benchmarks/wkv7_ab.py --B 16 --bwd $2 --rounds 4 2>/dev/null | paste -sd, >> $OUT
  echo "]}" >> $OUT
done
cat $OUT
