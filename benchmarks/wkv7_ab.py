"""Same-process A/B of the kernel generations (interleaved, best of several rounds):
python benchmarks/wkv7_ab.py --B 8 16 --bwd 4 -1 --fwd 0 -1   (4 / 0 = the predecessors kept for comparison)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs  # noqa: E402
from visualrwkv_amd import hip_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--fwd", type=int, nargs="+", default=[-1])
    ap.add_argument("--bwd", type=int, nargs="+", default=[-1])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = hip_lib.load()
    T, H, dev = 2624, 32, "cuda:0"
    for B in a.B:
        w, q, k, v, z, aa, dy = synth_inputs(B, T, H, dev)
        y = torch.empty_like(v)
        s = torch.empty(B, H, T // 16, 64, 64, dtype=torch.float32, device=dev)
        sa = torch.empty(B, T, H, 64, dtype=torch.float32, device=dev)
        g = [torch.empty_like(w) for _ in range(6)]
        st = torch.cuda.current_stream().cuda_stream

        def fwd():
            assert lib.vrwkv_wkv7_forward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, aa, y, s, sa)], st) == 0

        def bwd():
            assert lib.vrwkv_wkv7_backward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, aa, dy, s, sa, *g)], st) == 0

        def t(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters

        fwd(); bwd(); torch.cuda.synchronize()
        best = {}
        for _ in range(a.rounds):
            for fv in a.fwd:
                lib.vrwkv_wkv7_set_forward_variant(fv)
                fwd()
                best[("fwd", fv)] = min(best.get(("fwd", fv), 1e9), t(fwd))
            lib.vrwkv_wkv7_set_forward_variant(-1)
            fwd()
            for bv in a.bwd:
                lib.vrwkv_wkv7_set_backward_variant(bv)
                bwd()
                best[("bwd", bv)] = min(best.get(("bwd", bv), 1e9), t(bwd))
            lib.vrwkv_wkv7_set_backward_variant(-1)
        print(json.dumps({"B": B, **{f"{k}{v}": round(ms, 4) for (k, v), ms in best.items()}}))


if __name__ == "__main__":
    main()
