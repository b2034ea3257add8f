"""Per-kernel matrix-core utilisation of a training step from one rocprofv3 counter pass:
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d DIR -o p -- \
        python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python benchmarks/mfma_util.py DIR
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), summed over the launches of each kernel; kernels are
listed by their share of all MFMA-busy cycles (GEMMs, attention, the WKV7 kernels, the skinny weight gradients)."""
import collections, csv, glob, json, sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:96]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            calls[k] += 1
tot = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in agg.values()) or 1.0
tot_cu = sum(v.get("SQ_BUSY_CU_CYCLES", 0.0) for v in agg.values()) or 1.0
rows = []
for k, v in agg.items():
    m, cu = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CU_CYCLES", 0.0)
    if cu:
        rows.append({"kernel": k, "launches": calls[k], "mfma_util": round(m / (4 * cu), 4), "share_of_mfma_cycles": round(m / tot, 4),
                     "share_of_cu_busy_cycles": round(cu / tot_cu, 4)})
rows.sort(key=lambda r: -r["share_of_cu_busy_cycles"])
print(json.dumps({"whole_step_mfma_util": round(tot / (4 * tot_cu), 4), "kernels": rows[:40]}, indent=1))
