"""Which hardware queue does each HIP stream of this process land on?  Run under rocprofv3 --kernel-trace: one marker kernel per stream
(a fill of a distinct size) -> (Stream_Id, Queue_Id) pairs in the trace.  python benchmarks/stream_queues.py analyze <dir> prints them."""
import csv, glob, json, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "analyze":
    pairs = {}
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r.get("Stream_Id"), r.get("Queue_Id"))
            pairs.setdefault(k, set()).add(r.get("Kernel_Name", "")[:40] + " grid " + str(r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    print(json.dumps({"stream_queue_pairs": [{"stream": k[0], "queue": k[1], "kernels": sorted(v)[:4]} for k, v in sorted(pairs.items())]}))
    sys.exit(0)
import torch
x = torch.zeros(1 << 20, device="cuda:0")
torch.cuda.synchronize()
x[: 1000].fill_(1.0)                                   # default stream: 1000 elements
streams = [torch.cuda.Stream() for _ in range(6)] + [torch.cuda.Stream(priority=-1) for _ in range(3)]
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        x[: 2000 + 1000 * i].fill_(float(i))           # stream i: 2000 + 1000 i elements
torch.cuda.synchronize()
