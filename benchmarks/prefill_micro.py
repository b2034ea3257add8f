"""Inference prefill of one long prompt (B = 1, 32 heads = 32 workgroups for 256 CUs): the training forward, the
inference forward (no checkpoints / sa) and the sequence-parallel forward.  python benchmarks/prefill_micro.py [T]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs  # noqa: E402
from visualrwkv_amd import wkv7  # noqa: E402


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2624
    for B in (1, 2, 4):
        H = 32
        w, q, k, v, z, a, _ = synth_inputs(B, T, H, "cuda:0")
        y = torch.empty_like(v)
        s = torch.empty(B, H, T // 16, 64, 64, device="cuda")
        sa = torch.empty(B, T, H, 64, device="cuda")
        res = {"B": B, "T": T, "H": H,
               "train_fwd_ms": round(t(lambda: torch.ops.wind_backstepping.forward(w, q, k, v, z, a, y, s, sa)), 4),
               "infer_fwd_ms": round(t(lambda: wkv7.wkv7_forward_state(w, q, k, v, z, a)), 4)}
        P = wkv7.tparallel_segments(B, H, T)
        res["segments"] = P
        res["tparallel_ms"] = round(t(lambda: wkv7.wkv7_forward_tparallel(w, q, k, v, z, a)), 4)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
