#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 300 python benchmarks/dvfs_probe.py 16 2>&1 | grep -v amdgpu | tail -1 > $O/dvfs_probe.json
timeout 300 python benchmarks/wkv7_phases.py 16 2>&1 | tail -1 > $O/wkv7_phases_b16.json
timeout 300 python benchmarks/wkv7_phases.py 8 2>&1 | tail -1 > $O/wkv7_phases_b8.json
VRWKV_HIP_LIB=$PWD/benchmarks/_alt/lib_v6exp.so timeout 300 python benchmarks/wkv7_ab.py --B 16 --bwd 5 6 61 62 63 64 65 66 67 --rounds 3 2>&1 | grep -v amdgpu > $O/wkv7_roles.jsonl
timeout 600 python -m pytest tests/test_visual_gpu.py tests/test_attention_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3 > $O/pytest_vit.txt
for F in 1; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-grad-cp-companion 2>&1 | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['avg_ms'], r['roofline']['fwd_kernel']['avg_ms'])"; done > $O/bench_short.txt
cat $O/dvfs_probe.json | cut -c1-1200; cat $O/wkv7_roles.jsonl $O/pytest_vit.txt $O/bench_short.txt
