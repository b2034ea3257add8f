#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout 300 python benchmarks/wkv7_ab.py --B 16 --bwd 5 6 61 62 63 64 65 66 67 --rounds 3 2>&1 | grep -v amdgpu > $O/ab_prio.jsonl
BWDVAR=6 timeout 600 bash benchmarks/wkv7_pmc.sh 8 gpurun_out/r3c/pmc_v6 -1 notcc > $O/pmc_v6_b8.txt 2>&1
cat $O/ab_prio.jsonl; grep -A30 bwd_kernel_v6 $O/pmc_v6_b8.txt
