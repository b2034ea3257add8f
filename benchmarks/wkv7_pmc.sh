#!/bin/bash
# PMC passes for the WKV7 kernels (run on the GPU box from the repo root):
#   bash benchmarks/wkv7_pmc.sh <B> <outdir> [fwd variant]
# Counters are collected in separate passes (SQ has 8 slots, TCC 4; FETCH_SIZE costs 3, WRITE_SIZE 2).
B=${1:-8}; OUT=${2:-gpurun_out/pmc}; VAR=${3:--1}; R=$PWD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$n -o p -- python $R/benchmarks/wkv7_micro.py --B $B --iters 2 --variants $VAR --bwd-variant ${BWDVAR:--1} > /dev/null 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE
if [ "${4:-}" != "notcc" ]; then
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
fi
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wkv7" not in k: continue
        agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
# HBM traffic per launch for bench.py's roofline.traffic (merged into profiles/wkv7_pmc.json by hand-off below)
import json, os
B, T, H = $B, 2624, 32
elems = B * T * H * 64
rec = {}
for k, d in agg.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d: continue
    kind = "bwd" if "bwd" in k or "backward" in k else "fwd"
    f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    rec[f"{kind}_B{B}_T{T}_H{H}"] = {
        "kernel": k, "FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w),
        "hbm_bytes_per_launch": round((2 * f + w) * 1024), "algorithmic_bytes": elems * (46 if kind == "bwd" else 34),
        "note": "separate --pmc passes (benchmarks/wkv7_pmc.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                "(gfx950 reports half of wide coalesced reads); KB -> bytes x1024"}
json.dump(rec, open("$OUT/pmc.json", "w"), indent=1)
if os.environ.get("PMC_MERGE"):
    path = "profiles/wkv7_pmc.json"
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur.update(rec)
    json.dump(cur, open(path, "w"), indent=1)
PY
