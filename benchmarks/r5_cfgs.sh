#!/bin/bash
# Round 5: the grad_cp modes beside the headline, and the other BASELINE configurations at the micro-batches that fill the chip.
TAG=${1:-r5}; O=gpurun_out; mkdir -p $O
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lm_forward or selective" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${TAG}_bench_gradcp.json
python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_gradcp.json")); c = d["config"]
print("headline", round(d["value"]), "ms", round(d["ms_per_step"], 1), "peak", c["peak_mem_GB"], "bwd frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"])
print("grad_cp1", c["grad_cp1_same_run"]); print("grad_cp2", c["grad_cp2_same_run"])
PY
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-grad-cp-companion --fast-init"
: > $O/${TAG}_bench_cfg2_micro_batches.jsonl
for mb in 21 32 42; do
  timeout 400 $B --model 0b1 --towers siglip --ctx-len 1600 --img-tokens 576 --micro-bsz $mb 2>/dev/null | grep '^{"metric"' | tail -1 >> $O/${TAG}_bench_cfg2_micro_batches.jsonl
done
python - <<PY
import json
for l in open("$O/${TAG}_bench_cfg2_micro_batches.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print("cfg2 mb", d["config"]["micro_bsz"], round(d["value"]), "tok/s  bwd", round(r["frac"], 3), "fwd", round(r["fwd_kernel"]["frac"], 3), "peak GB", d["config"]["peak_mem_GB"])
PY
timeout 600 python benchmarks/bench_v6.py --fused 1 --micro-bsz 4 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg4.json
cat $O/${TAG}_bench_cfg4.json
