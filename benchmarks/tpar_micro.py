"""Sequence-parallel WKV7 training op (forward with checkpoints + two-pass backward) against the sequential kernels for
few heads: python benchmarks/tpar_micro.py [T] [H] -- prints one JSON line per (B, segments)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs  # noqa: E402
from visualrwkv_amd import wkv7  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = "cuda:0"
    for B in (1, 2, 4):
        w, q, k, v, z, a, dy = synth_inputs(B, T, H, dev)
        y = torch.empty_like(v)
        s = torch.empty(B, H, T // 16, 64, 64, dtype=torch.float32, device=dev)
        sa = torch.empty(B, T, H, 64, dtype=torch.float32, device=dev)
        g = [torch.empty_like(w) for _ in range(6)]
        fwd = lambda: torch.ops.wind_backstepping.forward(w, q, k, v, z, a, y, s, sa)
        bwd = lambda: torch.ops.wind_backstepping.backward(w, q, k, v, z, a, dy, s, sa, *g)
        fwd()
        base = {"fwd_ms": timeit(fwd), "bwd_ms": timeit(bwd)}
        for P in (2, 4, 8, 16):
            if T % P or (T // P) % 16:
                continue
            r = {"B": B, "T": T, "H": H, "segments": P, "seq_fwd_ms": round(base["fwd_ms"], 3), "seq_bwd_ms": round(base["bwd_ms"], 3),
                 "tpar_fwd_ms": round(timeit(lambda: wkv7.wkv7_forward_tparallel(w, q, k, v, z, a, segments=P, train=True)), 3),
                 "tpar_bwd_ms": round(timeit(lambda: wkv7.wkv7_backward_tparallel(w, q, k, v, z, a, dy, s, sa, P)), 3)}
            print(json.dumps(r))


if __name__ == "__main__":
    main()
