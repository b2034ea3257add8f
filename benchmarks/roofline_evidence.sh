#!/bin/bash
# One box, one commit: the bench line AND the rocprofv3 kernel statistics of the SAME process, plus the clock / power state of
# the box, so that roofline.frac can be recomputed from profiles/ (VERDICT r2, weak #2).   bash benchmarks/roofline_evidence.sh r3
TAG=${1:-r3}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
git -C $R rev-parse HEAD > $O/commit.txt 2>/dev/null || echo "snapshot without .git (gpurun)" > $O/commit.txt
rocm-smi --showclocks --showpower --showtemp > $O/smi_before.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-grad-cp-companion > $O/bench_under_rocprof.log 2>&1
cd $R
rocm-smi --showclocks --showpower --showtemp > $O/smi_after.txt 2>&1
grep '^{"metric"' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
timeout 300 python benchmarks/dvfs_probe.py 16 2>&1 | grep -v amdgpu | tail -1 > $O/dvfs_probe.json
python - <<PY
import csv, glob, json
O = "$O"
line = json.load(open(f"{O}/bench_under_rocprof.json"))
stats = {}
for f in glob.glob(f"{O}/step/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wkv7" in r["Name"]:
            stats[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
rf = line["roofline"]
bwd = next((v for k, v in stats.items() if "bwd_kernel" in k), None)
fwd = next((v for k, v in stats.items() if "fwd_kernel" in k), None)
ev = {"what": "bench.py under rocprofv3 --kernel-trace --stats, same process: HIP-event average (bench line) vs rocprofv3 average (kernel_stats.csv)",
      "commit": open(f"{O}/commit.txt").read().strip(), "bench_line": line, "rocprof_wkv7_kernels": stats}
if bwd:
    ms = bwd["avg_ns"] * 1e-6
    ev["bwd"] = {"events_avg_ms": rf["avg_ms"], "rocprof_avg_ms": ms, "ratio": rf["avg_ms"] / ms, "frac_from_rocprof": rf["algorithmic_bytes"] / (ms * 1e-3) / 8e12,
                 "frac_in_line": rf["frac"]}
if fwd and "fwd_kernel" in rf:
    ms = fwd["avg_ns"] * 1e-6
    ev["fwd"] = {"events_avg_ms": rf["fwd_kernel"]["avg_ms"], "rocprof_avg_ms": ms, "ratio": rf["fwd_kernel"]["avg_ms"] / ms, "frac_in_line": rf["fwd_kernel"]["frac"]}
ev["dvfs_probe"] = json.load(open(f"{O}/dvfs_probe.json"))
ev["smi_before"] = [l.strip() for l in open(f"{O}/smi_before.txt") if any(k in l for k in ("sclk", "mclk", "Power", "junction"))]
ev["smi_after"] = [l.strip() for l in open(f"{O}/smi_after.txt") if any(k in l for k in ("sclk", "mclk", "Power", "junction"))]
json.dump(ev, open(f"{O}/roofline_evidence.json", "w"), indent=1)
print(json.dumps({k: ev.get(k) for k in ("bwd", "fwd")}))
PY
# keep only the per-kernel summary of the trace (the full trace is tens of MB)
for f in $(find $O/step -name '*kernel_stats.csv'); do cp $f $O/step_kernel_stats.csv; done
rm -rf $O/step
