"""HBM streaming ceilings of this box by read : write mix (random data): what a kernel with the WKV7 forward's (12 : 22) or
backward's (34 : 12) traffic split can expect at best, next to the 1 : 1 copy."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd import hip_lib  # noqa: E402


def run(nbytes=1 << 30, iters=10):
    lib = hip_lib.load()
    dev = "cuda:0"
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    d0, d1 = torch.empty_like(a), torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, mode, moved in (("copy 1:1", 0, 2), ("fill 0:1", 1, 1), ("1 read : 2 writes", 2, 3), ("read only", 3, 1), ("2 reads : 1 write", 4, 3)):
        def go():
            if mode == 0:
                assert lib.vrwkv_stream_copy(a.data_ptr(), d0.data_ptr(), nbytes, st) == 0
            else:
                assert lib.vrwkv_stream_probe(mode, a.data_ptr(), b.data_ptr(), d0.data_ptr(), d1.data_ptr(), nbytes, st) == 0
        go(); torch.cuda.synchronize()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                go()
            e1.record(); torch.cuda.synchronize()
            best = max(best, moved * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        out[name] = round(best, 1)
    return out


if __name__ == "__main__":
    print(json.dumps({"hbm_GBps_by_mix": run()}))
