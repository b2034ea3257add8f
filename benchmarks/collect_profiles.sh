#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash benchmarks/collect_profiles.sh <tag>
TAG=${1:-r3}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench.json
timeout 600 python bench.py --steps 5 --warmup 2 --data loader --no-cpu-baseline --no-grad-cp-companion 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_loader.json
# the bench line and the rocprofv3 kernel statistics of the SAME process + clock / power state + the DVFS probe
bash benchmarks/roofline_evidence.sh $TAG > $O/roofline_evidence_summary.json 2>&1
timeout 300 python benchmarks/hbm_mix_probe.py 2>&1 | grep -v amdgpu | tail -1 > $O/hbm_mix_probe.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/step.log 2>&1
# matrix-core utilisation per kernel of the same step (counters in their own pass: no --stats, no other trace domain)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/step_pmc -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grad-cp-companion --fast-init > $O/step_pmc.log 2>&1
cd $R
python benchmarks/mfma_util.py $O/step_pmc > $O/step_mfma_util.json 2>&1; rm -rf $O/step_pmc
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 16 gpurun_out/$TAG/pmc_b16 -1 > $O/wkv7_pmc_b16.txt 2>&1
PMC_MERGE=1 bash benchmarks/wkv7_pmc.sh 8 gpurun_out/$TAG/pmc_b8 -1 > $O/wkv7_pmc_b8.txt 2>&1
cp profiles/wkv7_pmc.json $O/wkv7_pmc.json
# the N>1 code path (hooks, side stream, RCCL reduce-scatter / all-gather) on the one GPU of this box
VRWKV_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_rccl_1rank.json
python benchmarks/wkv7_phases.py 8 2>&1 | tail -1 > $O/wkv7_phases_b8.json
python benchmarks/wkv7_phases.py 16 2>&1 | tail -1 > $O/wkv7_phases_b16.json
python benchmarks/wkv7_ab.py --B 8 16 --fwd 1 -1 --bwd 5 6 --rounds 4 2>&1 | grep -v amdgpu > $O/wkv7_ab.jsonl
python benchmarks/wkv7_micro.py --B 8 16 32 --iters 20 2>&1 | grep -v amdgpu > $O/wkv7_micro.jsonl
# stateful generation: decode step (B = 1, 4), per-kernel breakdown of the captured step, prompt ingestion
python benchmarks/decode_micro.py 64 1 2>&1 | grep -v amdgpu | tail -1 > $O/decode_micro.jsonl
python benchmarks/decode_micro.py 64 4 2>&1 | grep -v amdgpu | tail -1 >> $O/decode_micro.jsonl
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/decprof -o dec -- python $R/benchmarks/decode_micro.py 64 > $O/decprof.log 2>&1)
python benchmarks/decode_profile.py $O/decprof/dec_results.db 4000 > $O/decode_kernels.txt 2>&1; rm -rf $O/decprof
python benchmarks/prefill_micro.py 2>&1 | grep -v amdgpu > $O/prefill_micro.jsonl
python benchmarks/wkv6_micro.py 2>&1 | grep -v amdgpu > $O/wkv6_micro.jsonl
python benchmarks/wgrad_micro.py 2>&1 | grep -v amdgpu > $O/wgrad_micro.jsonl
python benchmarks/tpar_micro.py 2>&1 | grep -v amdgpu > $O/tpar_micro.jsonl
python benchmarks/attention_micro.py 2>&1 | grep -v amdgpu > $O/attention_micro.jsonl
python benchmarks/patch_embed_micro.py 2>&1 | grep -v amdgpu > $O/patch_embed_micro.jsonl
python benchmarks/dgrad_layout_micro.py 2>&1 | grep -v amdgpu > $O/dgrad_layout_micro.jsonl
python benchmarks/image_micro.py 2>&1 | grep -v amdgpu > $O/image_micro.jsonl
bash benchmarks/attention_pmc.sh gpurun_out/$TAG/attn_pmc > $O/attention_pmc.txt 2>&1
cat $O/pytest_gpu.txt; cut -c1-600 $O/bench.json; cat $O/decode_micro.jsonl
