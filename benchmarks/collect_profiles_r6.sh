#!/bin/bash
# Round-6 evidence at the final tree, run on the GPU box from the repo root:  bash benchmarks/collect_profiles_r6.sh [tag]
# (the experiments of the round -- LDS layouts, S0 register prefetch, cache policies, full-row tail stores -- have their own files under
# profiles/r6b_*; this is the state of the product.)
TAG=${1:-r6}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/pytest_gpu.txt
VRWKV_TEST_NOTES=1 timeout 600 python -m pytest tests -m gpu -q -s -k "wkv7 or wkv6 or model or wgrad or fused" 2>&1 | grep '^\.*\[parity\]\|^\[parity\]' | sed 's/^\.*//' | sort | uniq > $O/parity_notes.txt
# counters of the DEFAULT backward of the bench shape (variant 9 at B = 16, variant 8 at B = 8) -> profiles/wkv7_pmc.json (bench.py's roofline.traffic)
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 16 gpurun_out/$TAG/pmc16 > $O/wkv7_pmc_b16.txt 2>&1
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 8 gpurun_out/$TAG/pmc8 > $O/wkv7_pmc_b8.txt 2>&1
rm -rf $O/pmc16 $O/pmc8; cp profiles/wkv7_pmc.json $O/wkv7_pmc.json
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric"' | tail -1 > $O/bench.json
timeout 600 python bench.py --grad-cp 2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_gradcp2_selective.json
timeout 600 python bench.py --grad-cp 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_gradcp1_reference_recipe.json
bash benchmarks/roofline_evidence.sh $TAG > $O/roofline_evidence_summary.json 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/step_pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grad-cp-companion --fast-init > $O/step_pmc.log 2>&1
cd $R
python benchmarks/mfma_util.py $O/step_pmc > $O/step_mfma_util.json 2>&1; rm -rf $O/step_pmc $O/step_pmc.log
python benchmarks/wkv7_ab.py --B 8 16 --fwd 4 -1 --bwd 5 8 9 --rounds 4 2>&1 | grep -v amdgpu > $O/wkv7_ab.jsonl
python benchmarks/wkv7_micro.py --B 8 16 32 --iters 20 2>&1 | grep -v amdgpu > $O/wkv7_micro.jsonl
python benchmarks/wkv6_micro.py 2 4 8 16 2>&1 | grep '^{' > $O/wkv6_micro.jsonl
python benchmarks/eltwise_micro.py 16 2>&1 | grep '^{' > $O/eltwise_micro.json
python benchmarks/attention_micro.py 2>&1 | grep '^{' > $O/attention_micro.jsonl
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-grad-cp-companion --fast-init"
timeout 400 $B --model 0b1 --towers siglip --ctx-len 1600 --img-tokens 576 --micro-bsz 42 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg2.json
timeout 400 $B --model 0b1 --towers siglip --ctx-len 1600 --img-tokens 576 --micro-bsz 32 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg2_mb32.json
timeout 400 $B --model 1b5 --towers dino,siglip,sam --ctx-len 6400 --img-tokens 2304 --micro-bsz 8 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg5.json
timeout 600 python benchmarks/bench_v6.py --fused 1 --micro-bsz 4 2>&1 | grep -v amdgpu | tail -1 > $O/bench_cfg4.json
cat $O/pytest_gpu.txt; cut -c1-900 $O/bench.json
