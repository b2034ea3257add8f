#!/bin/bash
# round 3, first GPU call: validate the CPU-side changes on the box + fresh baselines
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench.json
timeout 300 python benchmarks/dvfs_probe.py 16 2>&1 | grep -v amdgpu | tail -1 > $O/dvfs_probe.json
timeout 300 python benchmarks/wkv7_phases.py 8 2>&1 | tail -1 > $O/wkv7_phases_b8.json
timeout 600 python bench.py --steps 5 --warmup 2 --data loader --no-cpu-baseline --no-grad-cp-companion 2>&1 | grep -v amdgpu.ids | tail -3 > $O/bench_loader.json
cat $O/pytest_gpu.txt; cut -c1-1500 $O/bench.json; cat $O/dvfs_probe.json; cat $O/wkv7_phases_b8.json; cut -c1-700 $O/bench_loader.json
