"""GPU idle time inside a traced step: union of the kernel intervals of all queues vs the wall time between the first and the last kernel of the step.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 3 --warmup 2 ... ;  python benchmarks/step_idle.py DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps: from one adamw launch group to the next (the optimizer ends a step)
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
groups = []
for i in ends:
    if not groups or i - groups[-1][-1] > 50:
        groups.append([i])
    else:
        groups[-1].append(i)
bounds = [g[-1] for g in groups]
for a, b in zip(bounds[:-1], bounds[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, cur_s, cur_e = 0, None, None
    gaps = collections.Counter()
    for s, e, n in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps[n[:50]] += s - cur_e
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"step: wall {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms ({100 * (1 - busy / (t1 - t0)):.1f} %), kernels {len(seg)}")
    print("   largest idle before:", [(k, round(v / 1e6, 2)) for k, v in gaps.most_common(8)])
