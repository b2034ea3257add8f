"""What the N > 1 code path of dp.Zero1Engine puts on the communication stream, on the ONE GPU a gpurun box has.

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -o rccl -- python benchmarks/rccl_overlap.py run [bench.py flags]
  python benchmarks/rccl_overlap.py analyze <dir> <out.json>

`run` = bench.py with VRWKV_FORCE_COLLECTIVES=1 (hooks, side stream, RCCL reduce-scatter per 200 MB bucket, sharded AdamW, RCCL
all-gather) on a one-rank communicator.  RCCL short-cuts a one-rank collective: in place it launches nothing, out of place it is a device
copy.  With VRWKV_COLLECTIVE_OOP=1 (the default of `run`) the engine's two collectives are therefore issued OUT OF PLACE through a
scratch buffer, so that every bucket leaves a record on the communication stream -- RCCL's copy of the bucket plus the copy back -- whose
start / end can be laid against the compute-stream kernels of the backward pass (reduce-scatter) and of the next step's ViT encode
(all-gather).  That shows stream placement and concurrency, which is what one GPU can show; it is NOT a measurement of xGMI traffic.

`analyze` reads rocprofv3's kernel and memory-copy traces: the stream with the most kernel time is "compute"; everything on another
stream (RCCL kernels if any, copies, the engine's own kernels under `with torch.cuda.stream(comm)`) is "comm".  Per comm record: start,
end, and the part of its duration during which a compute-stream kernel was running."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    os.environ["VRWKV_FORCE_COLLECTIVES"] = "1"
    oop = os.environ.setdefault("VRWKV_COLLECTIVE_OOP", "1") == "1"
    rep = int(os.environ.get("VRWKV_COMM_PROXY_REPEAT", "1"))      # issue each bucket's out-of-place collective this many times: a stand-in for the
    timeline = os.environ.get("VRWKV_RCCL_TIMELINE")               # duration of a real 8-rank collective (a 200 MB bucket: ~1-2 ms over xGMI)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from visualrwkv_amd import dp

    marks = []          # (label, start event, end event) of the current step, on whatever stream they were recorded

    def mark(label, fn):
        if timeline is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        marks.append((label, e0, e1))
        return out

    if oop:
        scratch = {}

        def buf_like(t, key):
            b = scratch.get(key)
            if b is None or b.numel() < t.numel() or b.dtype != t.dtype:
                b = scratch[key] = torch.empty(t.numel(), dtype=t.dtype, device=t.device)
            return b[:t.numel()]

        def reduce_scatter(self, buf, b):
            piece = buf[self.rank * b.piece:(self.rank + 1) * b.piece]
            tmp = buf_like(piece, "rs")

            def go():
                for _ in range(rep):
                    dist.reduce_scatter_tensor(tmp, buf, op=dist.ReduceOp.SUM, group=self.pg)
                piece.copy_(tmp)
            mark(f"reduce_scatter bucket {self.buckets.index(b)}", go)

        def all_gather(self):
            for i, b in enumerate(self.buckets):
                buf = self.flat_param[b.start:b.end]
                piece = buf[self.rank * b.piece:(self.rank + 1) * b.piece]
                tmp = buf_like(buf, "ag")

                def go():
                    for _ in range(rep):
                        dist.all_gather_into_tensor(tmp, piece, group=self.pg)
                    buf.copy_(tmp)
                mark(f"all_gather bucket {i}", go)

        dp.Zero1Engine._reduce_scatter = reduce_scatter
        dp.Zero1Engine._all_gather = all_gather
    if timeline is not None:
        # compute-stream marks: the whole backward (zero_grad -> step() entry) and the optimizer step; the communication-stream marks above are
        # recorded under `with torch.cuda.stream(comm)` by the engine, i.e. on the communication stream
        steps = []
        zg, st = dp.Zero1Engine.zero_grad, dp.Zero1Engine.step

        def zero_grad(self, *a, **k):
            if marks:
                steps.append(list(marks))
                marks.clear()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(("step origin (zero_grad, compute stream)", e, e))
            return zg(self, *a, **k)

        def step(self, *a, **k):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(("backward enqueued: engine.step() entered (compute stream)", e, e))
            out = st(self, *a, **k)
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            marks.append(("engine.step() returned (compute stream: AdamW done, all-gather left on the communication stream)", e2, e2))
            return out

        dp.Zero1Engine.zero_grad, dp.Zero1Engine.step = zero_grad, step
        import atexit

        def dump():
            torch.cuda.synchronize()
            if marks:
                steps.append(list(marks))
            full = [s for s in steps if len(s) > 3]
            if len(full) < 2:
                return
            cur, nxt = full[-2], full[-1]            # the all-gather of a step is recorded by that step; its consumers run in the next one
            org = cur[0][1]
            rows = [{"what": l, "start_ms": round(org.elapsed_time(a), 3), "end_ms": round(org.elapsed_time(b), 3)} for l, a, b in cur]
            rows.append({"what": "next step origin", "start_ms": round(org.elapsed_time(nxt[0][1]), 3), "end_ms": round(org.elapsed_time(nxt[0][1]), 3)})
            json.dump({"what": "HIP-event timeline of one training step, ms from zero_grad: compute-stream marks and, per bucket, the communication-stream "
                               f"work (out-of-place one-rank RCCL collective x {rep} + copy back)", "repeat": rep, "rows": rows}, open(timeline, "w"), indent=1)
        atexit.register(dump)
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    import bench
    bench.main()


def _col(row, *names):
    for n in names:
        for k in row:
            if k.lower() == n.lower():
                return row[k]
    return None


def _load(dirname):
    recs = []
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append({"kind": "kernel", "name": _col(r, "Kernel_Name") or "", "stream": _col(r, "Stream_Id") or _col(r, "Queue_Id") or "?",
                         "queue": _col(r, "Queue_Id") or "?", "t0": int(_col(r, "Start_Timestamp")), "t1": int(_col(r, "End_Timestamp"))})
    for f in glob.glob(os.path.join(dirname, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append({"kind": "copy", "name": (_col(r, "Direction") or "copy") + " " + (_col(r, "Size") or _col(r, "Bytes") or ""),
                         "stream": _col(r, "Stream_Id") or "copy-engine", "queue": "-", "t0": int(_col(r, "Start_Timestamp")), "t1": int(_col(r, "End_Timestamp"))})
    return recs


def analyze(dirname, out):
    recs = _load(dirname)
    assert recs, f"no rocprofv3 traces under {dirname}"
    busy = {}
    for r in recs:
        if r["kind"] == "kernel":
            busy[r["stream"]] = busy.get(r["stream"], 0) + r["t1"] - r["t0"]
    compute = max(busy, key=busy.get)
    comp = sorted((r["t0"], r["t1"]) for r in recs if r["kind"] == "kernel" and r["stream"] == compute)
    # merged busy intervals of the compute stream
    merged = []
    for a, b in comp:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    starts = [m[0] for m in merged]
    import bisect

    def overlap(a, b):
        i = max(bisect.bisect_right(starts, a) - 1, 0)
        tot = 0
        while i < len(merged) and merged[i][0] < b:
            tot += max(0, min(b, merged[i][1]) - max(a, merged[i][0]))
            i += 1
        return tot

    def concurrent_names(a, b, k=3):
        names = {}
        for r in recs:
            if r["kind"] == "kernel" and r["stream"] == compute and r["t0"] < b and r["t1"] > a:
                n = r["name"].split("(")[0][-60:]
                names[n] = names.get(n, 0) + min(b, r["t1"]) - max(a, r["t0"])
        return [n for n, _ in sorted(names.items(), key=lambda kv: -kv[1])[:k]]

    comm = sorted((r for r in recs if not (r["kind"] == "kernel" and r["stream"] == compute)), key=lambda r: r["t0"])
    # the last full step: from the last adamw kernel backwards to the adamw kernel group before it
    adam = [r["t0"] for r in recs if "adamw" in r["name"]]
    t_end = max(r["t1"] for r in recs)
    groups = []
    for t in sorted(adam):
        if not groups or t - groups[-1][-1] > 50_000_000:        # > 50 ms apart: another step
            groups.append([t])
        else:
            groups[-1].append(t)
    win = (groups[-2][0], groups[-1][0]) if len(groups) >= 2 else (min(r["t0"] for r in recs), t_end)
    t0w = win[0]
    rows = []
    for r in comm:
        if r["t0"] < win[0] or r["t0"] >= win[1]:
            continue
        d = r["t1"] - r["t0"]
        ov = overlap(r["t0"], r["t1"])
        rows.append({"kind": r["kind"], "name": r["name"].split("(")[0][-70:], "stream": r["stream"], "start_ms": round((r["t0"] - t0w) * 1e-6, 3),
                     "end_ms": round((r["t1"] - t0w) * 1e-6, 3), "us": round(d * 1e-3, 1), "hidden_frac": round(ov / d, 3) if d else None,
                     "beside": concurrent_names(r["t0"], r["t1"]) if d > 20_000 else None})
    tot = sum(x["us"] for x in rows)
    hid = sum(x["us"] * (x["hidden_frac"] or 0) for x in rows)
    big = [x for x in rows if x["us"] >= 20]
    res = {"what": "one optimizer step (from one AdamW group to the next) of bench.py with VRWKV_FORCE_COLLECTIVES=1 on one MI355X under "
                   "rocprofv3 --kernel-trace --memory-copy-trace: every record that is NOT a kernel of the compute stream",
           "compute_stream": compute, "streams_kernel_busy_ms": {k: round(v * 1e-6, 2) for k, v in busy.items()},
           "step_window_ms": round((win[1] - win[0]) * 1e-6, 2), "comm_records": len(rows), "comm_total_us": round(tot, 1),
           "comm_hidden_frac": round(hid / tot, 3) if tot else None, "records_of_20us_and_more": big[:80]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "records_of_20us_and_more"}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) > 3 and sys.argv[1] == "analyze":
        analyze(sys.argv[2], sys.argv[3])
    else:
        print(__doc__)
