"""Where do the library's stream-K GEMMs (`_SK3_` in the kernel name) sit in a traced step, and do two of them ever overlap in time?
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 1 --warmup 1 ... ;  python benchmarks/streamk_in_step.py DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sk = [i for i, r in enumerate(rows) if "_SK3_" in r["Kernel_Name"] or "StreamK" in r["Kernel_Name"]]
print(list(rows[0].keys()))
print("kernels", len(rows), "stream-K launches", len(sk), "queues", collections.Counter(rows[i]["Queue_Id"] for i in sk))
ctx = collections.Counter()
for i in sk:
    prev = next((rows[j]["Kernel_Name"][:60] for j in range(i - 1, -1, -1) if rows[j]["Queue_Id"] == rows[i]["Queue_Id"]), "")
    r = rows[i]
    ctx[(r["Queue_Id"], r.get("Grid_Size", r.get("Grid_Size_X")), r.get("Workgroup_Size", r.get("Workgroup_Size_X")), r["Kernel_Name"][:48] + ".." + r["Kernel_Name"][-40:], prev)] += 1
for k, n in ctx.most_common(12):
    print(n, k)
over = 0
for a, i in enumerate(sk):
    for j in sk[a + 1:a + 4]:
        if int(rows[j]["Start_Timestamp"]) < int(rows[i]["End_Timestamp"]):
            over += 1
print("pairs of stream-K launches overlapping in time:", over)
