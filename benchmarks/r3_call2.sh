#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_wkv7_gpu.py -x -q -k "backward_parity" 2>&1 | tail -5 > $O/pytest_v6.txt
timeout 300 python benchmarks/wkv7_ab.py --B 8 16 --bwd 5 6 --rounds 4 2>&1 | grep -v amdgpu > $O/ab_v6.jsonl
timeout 300 python benchmarks/wkv7_phases.py 8 2>&1 | tail -1 > $O/phases_b8.json
timeout 300 python benchmarks/wkv7_phases.py 16 2>&1 | tail -1 > $O/phases_b16.json
cat $O/pytest_v6.txt $O/ab_v6.jsonl $O/phases_b8.json $O/phases_b16.json
