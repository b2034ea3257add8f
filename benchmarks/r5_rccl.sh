#!/bin/bash
# (1) the forced-collective path of one rank under rocprofv3 (kernel + memory-copy trace) -> profiles-ready summary;
# (2) same-box A/B of the plain one-rank step against the forced-collective step (the product path, in place), alternating processes.
TAG=${1:-r5}; R=$PWD; O=$R/gpurun_out; mkdir -p $O/rccl
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/rccl -o rccl -- python $R/benchmarks/rccl_overlap.py run --steps 3 --warmup 2 --no-cpu-baseline --no-grad-cp-companion --fast-init > $O/${TAG}_rccl_run.log 2>&1
cd $R
python benchmarks/rccl_overlap.py analyze $O/rccl $O/${TAG}_rccl_overlap.json
rm -rf $O/rccl
: > $O/${TAG}_rccl_ab.txt
for rep in 1 2; do
  for mode in plain forced; do
    F=0; [ $mode = forced ] && F=1
    VRWKV_FORCE_COLLECTIVES=$F python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$mode', 'ms_per_step', round(d['ms_per_step'], 2), 'tokens_per_s', round(d['value']), 'backend', d.get('backend'))
" >> $O/${TAG}_rccl_ab.txt
  done
done
cat $O/${TAG}_rccl_ab.txt
