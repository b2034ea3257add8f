#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
timeout 300 python -m pytest tests/test_wkv7_gpu.py -q -k backward_parity 2>&1 | tail -1
timeout 300 python benchmarks/wkv7_ab.py --B 8 16 --bwd 5 6 --rounds 4 2>&1 | grep -v amdgpu | tee $O/ab.jsonl
for V in 5 6; do
VRWKV_BWD_VARIANT=$V timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-grad-cp-companion 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_v$V.json
python - <<PY
import json; r=json.load(open("$O/bench_v$V.json")); print($V, r["value"], r["ms_per_step"], r["roofline"]["avg_ms"], r["roofline"]["frac"], r["roofline"]["fwd_kernel"]["avg_ms"])
PY
done
