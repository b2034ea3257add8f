#!/bin/bash
# in-step A/B of library builds x backward variants, alternating processes: bash benchmarks/r5_step_libs.sh <tag> "<lib:variant> ..." [reps]
TAG=${1:-r5}; CASES=$2; REPS=${3:-2}; O=gpurun_out; : > $O/${TAG}_step_libs_ab.txt
for rep in $(seq $REPS); do
  for c in $CASES; do
    lib=${c%%:*}; v=${c##*:}
    VRWKV_HIP_LIB=benchmarks/_alt/lib_$lib.so VRWKV_BWD_VARIANT=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$lib variant $v', 'ms_per_step', round(d['ms_per_step'], 2), 'bwd_ms', round(r['avg_ms'], 4), 'frac', round(r['frac'], 4), r['kernel'], 'fwd_frac', round(r['fwd_kernel']['frac'], 4))
" >> $O/${TAG}_step_libs_ab.txt
  done
done
cat $O/${TAG}_step_libs_ab.txt
