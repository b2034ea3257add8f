#!/bin/bash
# HBM read-traffic cross-check for the WKV7 backward: FETCH_SIZE vs EA read requests at several batch sizes.
R=$PWD; O=$R/gpurun_out/fetchprobe; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/tcc_counters.txt
for B in 8 12 16 32; do
  for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/B${B}_$n -o p -- python $R/benchmarks/wkv7_micro.py --B $B --iters 1 --variants -1 > $O/B${B}_$n.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/fetchprobe/*/*counter_collection.csv")):
    tag = f.split("/")[2].split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wkv7" not in k: continue
        kind = "bwd" if "bwd" in k else "fwd"
        agg[(tag, kind, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(k, sum(v) / len(v), len(v))
PY
