#!/bin/bash
# PMC passes for the ViT attention kernels (run on the GPU box from the repo root):
#   bash benchmarks/attention_pmc.sh <outdir> [case substring]
# Counters in separate passes (SQ has 8 slots); prints per-kernel averages and the MFMA utilisation
# SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) per kernel.
OUT=${1:-gpurun_out/attn_pmc}; R=$PWD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$n -o p -- python $R/benchmarks/attention_micro.py --hip-only --iters 2 > /dev/null 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
cd $R
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vattn" not in k: continue
        key = k.split("(")[0] + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CU_CYCLES"):
        # MFMA-pipe busy cycles summed over SIMDs / (CU-busy cycles x 4 SIMDs per CU)
        m["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_BUSY_CU_CYCLES"])
    out[k] = m
    print(k)
    for c, v in sorted(m.items()):
        print(f"   {c:28s} {v:16.3f}")
json.dump(out, open("$OUT/attention_pmc.json", "w"), indent=1)
PY
