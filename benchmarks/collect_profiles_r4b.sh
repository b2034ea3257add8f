#!/bin/bash
# Round-4 evidence at the final tree (second collection: after the streaming launch shapes, the WKV6 backward v2 and the re-dealt score
# pieces of the WKV7 backward), run on the GPU box from the repo root:  bash benchmarks/collect_profiles_r4b.sh [tag]
# The probes whose subject did not change since benchmarks/collect_profiles_r4.sh (memory-role probe, role-off builds, phase stamps,
# LoRA / big weight-gradient micro, sequence-parallel, decode, attention) are not repeated.
TAG=${1:-r4b}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/pytest_gpu.txt
VRWKV_TEST_NOTES=1 timeout 600 python -m pytest tests -m gpu -q -s -k "wkv7 or wkv6 or model or wgrad or fused" 2>&1 | grep '^\.*\[parity\]\|^\[parity\]' | sed 's/^\.*//' | sort | uniq > $O/parity_notes.txt
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep '^{"metric"' | tail -1 > $O/bench.json
bash benchmarks/roofline_evidence.sh $TAG > $O/roofline_evidence_summary.json 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/step_pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grad-cp-companion --fast-init > $O/step_pmc.log 2>&1
cd $R
python benchmarks/mfma_util.py $O/step_pmc > $O/step_mfma_util.json 2>&1; rm -rf $O/step_pmc
python benchmarks/wkv7_ab.py --B 8 16 --fwd 4 -1 --bwd 5 6 7 8 9 --rounds 4 2>&1 | grep -v amdgpu > $O/wkv7_ab.jsonl
python benchmarks/wkv7_micro.py --B 8 16 32 --iters 20 2>&1 | grep -v amdgpu > $O/wkv7_micro.jsonl
for v in 1 2; do VRWKV_WKV6_BWD_VARIANT=$v python benchmarks/wkv6_micro.py 2 4 8 16 2>&1 | grep '^{' | sed "s/^{/{\"bwd_variant\": $v, /"; done > $O/wkv6_micro.jsonl
python benchmarks/eltwise_micro.py 16 2>&1 | grep '^{' > $O/eltwise_micro.json
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-grad-cp-companion --fast-init"
timeout 400 $B --model 0b1 --towers siglip --ctx-len 1600 --img-tokens 576 --micro-bsz 16 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg2.json
timeout 400 $B --model 1b5 --towers dino,siglip,sam --ctx-len 6400 --img-tokens 2304 --micro-bsz 8 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg5.json
timeout 600 python benchmarks/bench_v6.py --fused 1 2>&1 | grep -v amdgpu | tail -1 > $O/bench_cfg4.json
cat $O/pytest_gpu.txt; cut -c1-700 $O/bench.json
