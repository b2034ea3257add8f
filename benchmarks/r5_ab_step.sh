#!/bin/bash
# Same-box A/B of WKV7 backward variants: micro-benchmark, then the training step with alternating processes.
# usage: bash benchmarks/r5_ab_step.sh <tag> "<variants>" [steps]
TAG=${1:-r5}; VARS=${2:-"9 10"}; STEPS=${3:-8}
OUT=gpurun_out; mkdir -p $OUT
python benchmarks/wkv7_ab.py --B 8 16 --bwd $VARS --rounds 5 > $OUT/${TAG}_wkv7_ab.jsonl 2> $OUT/${TAG}_wkv7_ab.err
cat $OUT/${TAG}_wkv7_ab.jsonl
: > $OUT/${TAG}_step_ab.txt
for rep in 1 2; do
  for v in $VARS; do
    VRWKV_BWD_VARIANT=$v python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('variant $v', 'ms_per_step', round(d['ms_per_step'], 2), 'bwd_ms', round(r['avg_ms'], 4), 'frac', round(r['frac'], 4), r['kernel'], 'fwd_frac', round(r['fwd_kernel']['frac'], 4))
" >> $OUT/${TAG}_step_ab.txt
  done
done
cat $OUT/${TAG}_step_ab.txt
