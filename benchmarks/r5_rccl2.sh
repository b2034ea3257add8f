#!/bin/bash
# The communication stream under load on one GPU: per-bucket out-of-place one-rank RCCL collectives, issued 1x and 20x (a stand-in for the
# duration of an 8-rank collective), HIP-event timeline of one step + step time against the plain one-rank step on the same box.
TAG=${1:-r5}; O=gpurun_out; mkdir -p $O
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step', round(d['ms_per_step'], 2), 'tokens_per_s', round(d['value']), 'backend', d.get('backend'))
"; }
: > $O/${TAG}_rccl_proxy_ab.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | line plain >> $O/${TAG}_rccl_proxy_ab.txt
for rep in 1 20; do
  VRWKV_COMM_PROXY_REPEAT=$rep VRWKV_RCCL_TIMELINE=$O/${TAG}_rccl_timeline_x$rep.json python benchmarks/rccl_overlap.py run --steps 8 --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | line "forced_oop_x$rep" >> $O/${TAG}_rccl_proxy_ab.txt
done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-grad-cp-companion 2>/dev/null | line plain >> $O/${TAG}_rccl_proxy_ab.txt
cat $O/${TAG}_rccl_proxy_ab.txt
