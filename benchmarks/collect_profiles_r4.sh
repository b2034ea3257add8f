#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root:  bash benchmarks/collect_profiles_r4.sh [tag]
# (benchmarks/_alt/lib_exp.so = the library built with -DVRWKV_V6_EXPERIMENTS by benchmarks/build_alt_src.sh exp wkv7_capi.hip ...,
#  benchmarks/_alt/mem_role_probe = benchmarks/mem_role_probe.hip: both built before the call, they travel with the snapshot)
TAG=${1:-r4}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/pytest_gpu.txt
VRWKV_TEST_NOTES=1 timeout 600 python -m pytest tests -m gpu -q -s -k "wkv7 or wkv6 or model or wgrad or fused" 2>&1 | grep '^\.*\[parity\]\|^\[parity\]' | sed 's/^\.*//' | sort | uniq > $O/parity_notes.txt
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep '^{"metric"' | tail -1 > $O/bench.json
timeout 600 python bench.py --steps 5 --warmup 2 --data loader --no-cpu-baseline --no-grad-cp-companion 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_loader.json
bash benchmarks/roofline_evidence.sh $TAG > $O/roofline_evidence_summary.json 2>&1
timeout 300 python benchmarks/hbm_mix_probe.py 2>&1 | grep -v amdgpu | tail -1 > $O/hbm_mix_probe.json
timeout 240 benchmarks/_alt/mem_role_probe 16 > $O/mem_role_probe.jsonl 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/step_pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grad-cp-companion --fast-init > $O/step_pmc.log 2>&1
cd $R
python benchmarks/mfma_util.py $O/step_pmc > $O/step_mfma_util.json 2>&1; rm -rf $O/step_pmc
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 16 gpurun_out/$TAG/pmc_b16 -1 > $O/wkv7_pmc_b16.txt 2>&1
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 8 gpurun_out/$TAG/pmc_b8 -1 > $O/wkv7_pmc_b8.txt 2>&1
cp profiles/wkv7_pmc.json $O/wkv7_pmc.json; rm -rf gpurun_out/$TAG/pmc_b16 gpurun_out/$TAG/pmc_b8
VRWKV_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_rccl_1rank.json
python benchmarks/wkv7_phases.py 16 2>&1 | tail -1 > $O/wkv7_phases_b16.json
python benchmarks/wkv7_ab.py --B 8 16 --fwd 4 -1 --bwd 5 6 7 8 --rounds 4 2>&1 | grep -v amdgpu > $O/wkv7_ab.jsonl
VRWKV_HIP_LIB=benchmarks/_alt/lib_exp.so python benchmarks/wkv7_ab.py --B 16 --bwd 6 61 62 63 64 65 66 67 8 81 82 83 84 85 86 87 88 --rounds 3 2>&1 | grep -v amdgpu > $O/wkv7_roles.jsonl
python benchmarks/wkv7_micro.py --B 8 16 32 --iters 20 2>&1 | grep -v amdgpu > $O/wkv7_micro.jsonl
python benchmarks/wgrad_big_micro.py --head 2>&1 | grep -v amdgpu > $O/wgrad_big_micro.jsonl
python benchmarks/fused_micro.py 2>&1 | grep -v amdgpu | tail -1 > $O/fused_micro.json
python benchmarks/wkv6_micro.py 2>&1 | grep -v amdgpu > $O/wkv6_micro.jsonl
python benchmarks/wgrad_micro.py 2>&1 | grep -v amdgpu > $O/wgrad_micro.jsonl
python benchmarks/tpar_micro.py 2>&1 | grep -v amdgpu > $O/tpar_micro.jsonl
python benchmarks/attention_micro.py 2>&1 | grep -v amdgpu > $O/attention_micro.jsonl
python benchmarks/decode_micro.py 64 1 2>&1 | grep -v amdgpu | tail -1 > $O/decode_micro.jsonl
python benchmarks/decode_micro.py 64 4 2>&1 | grep -v amdgpu | tail -1 >> $O/decode_micro.jsonl
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-grad-cp-companion --fast-init"
timeout 400 $B --model 0b1 --towers siglip --ctx-len 1600 --img-tokens 576 --micro-bsz 16 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg2.json
timeout 400 $B --model 1b5 --towers dino,siglip,sam --ctx-len 6400 --img-tokens 2304 --micro-bsz 8 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg5.json
timeout 400 $B --model 1b5 --towers dino,siglip,sam --ctx-len 6400 --img-tokens 2304 --micro-bsz 4 2>&1 | grep '^{"metric"' | tail -1 > $O/bench_cfg5_mb4.json
timeout 600 python benchmarks/bench_v6.py --fused 1 2>&1 | grep -v amdgpu | tail -1 > $O/bench_cfg4.json
cat $O/pytest_gpu.txt; cut -c1-700 $O/bench.json
