#!/bin/bash
# per-wave phase stamps of the v8 backward schedules (profiling libraries built by benchmarks/build_alt.sh prof_v<variant>_w<wave>)
OUT=gpurun_out/${1:-r5}_wkv7_phases_waves.jsonl; : > $OUT
for lib in benchmarks/_alt/lib_prof_v*_w*.so; do
  echo -n "{\"lib\": \"$(basename $lib)\", \"stamps\": " >> $OUT
  VRWKV_PHASES_ONLY=4 VRWKV_HIP_LIB=$lib python benchmarks/wkv7_phases.py 16 >> $OUT
  echo "}" >> $OUT
done
cat $OUT
