"""Timing of the fused element-wise kernels at the benchmark shape (B x 2624 x 2048): ms per call and the
effective HBM rate for the algorithmic bytes.  python benchmarks/fused_micro.py [B]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd import fused  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    T, C = 2624, 2048
    dev = "cuda"
    x = torch.randn(B, T, C, device=dev, dtype=torch.bfloat16)
    mus = [torch.rand(1, 1, C, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(6)]
    n = B * T * C
    res = {}
    xr = x.clone().requires_grad_(True)
    outs = fused.mix(xr, *mus)
    gouts = [torch.randn_like(x) for _ in range(6)]
    res["mix6_fwd_ms"] = timeit(lambda: fused.mix(x, *[m.detach() for m in mus]))
    res["mix6_fwd_GBps"] = n * 14 / res["mix6_fwd_ms"] / 1e6

    def bwd():
        xr.grad = None
        torch.autograd.backward(outs, gouts, retain_graph=True)
    res["mix6_bwd_ms"] = timeit(bwd)
    res["mix6_bwd_GBps"] = n * 16 / res["mix6_bwd_ms"] / 1e6
    xr1 = x.clone().requires_grad_(True)
    o1 = fused.mix(xr1, mus[0])
    res["mix1_bwd_ms"] = timeit(lambda: torch.autograd.backward(o1, gouts[:1], retain_graph=True))
    res["mix1_bwd_GBps"] = n * 6 / res["mix1_bwd_ms"] / 1e6
    # kva (a-gate, value residual, kk normalise, k modulation) and post (GroupNorm + bonus + gate), training form with the aliased outputs
    kk_, vv_, vf_, vl_, al_ = [torch.randn(B, T, C, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(5)]
    kp = [torch.randn(1, 1, C, device=dev, dtype=torch.bfloat16).mul_(0.5).requires_grad_(True) for _ in range(4)]
    kouts = fused.kva(kk_, vv_, vf_, vl_, al_, *kp, True)
    kg = [gouts[i % 6] for i in range(len(kouts))]
    res["kva_fwd_ms"] = timeit(lambda: fused.kva(kk_.detach(), vv_.detach(), vf_.detach(), vl_.detach(), al_.detach(), *[q.detach() for q in kp]))
    res["kva_bwd_ms"] = timeit(lambda: torch.autograd.backward(kouts, kg, retain_graph=True))
    yy, rr, gg = [torch.randn(B, T, C, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(3)]
    lw, lb_, rk_ = [torch.randn(C, device=dev, dtype=torch.bfloat16).mul_(0.5).requires_grad_(True) for _ in range(3)]
    pout = fused.post(yy, rr, kk_, vv_, gg, lw, lb_, rk_.view(C // 64, 64), 64e-5)
    res["post_bwd_ms"] = timeit(lambda: torch.autograd.backward(pout, gouts[0], retain_graph=True))
    # LayerNorm + lerps in one kernel against the two-kernel path (bytes: the two-kernel path's algorithmic bytes)
    ln = torch.nn.LayerNorm(C).to(dev).bfloat16()
    delta = torch.randn_like(x)
    gres = torch.randn_like(x)
    for M in (6, 1):
        ms = mus[:M]
        for name, fwd in (("ln_mix", lambda xx, dd: fused.add_ln_mix(xx, dd, ln, ms, M == 6)),
                          ("add_ln+mix", lambda xx, dd: (lambda xn, h: (xn, list((fused.mix_dup3 if M == 6 else fused.mix)(h, *ms))))(*fused.add_ln(xx, dd, ln)))):
            xg, dg = x.clone().requires_grad_(True), delta.clone().requires_grad_(True)
            with torch.no_grad():
                res[f"{name}{M}_fwd_ms"] = timeit(lambda: fwd(x, delta))
            xn, outs = fwd(xg, dg)
            go = [gouts[i % 6] for i in range(len(outs))]
            res[f"{name}{M}_bwd_ms"] = timeit(lambda: torch.autograd.backward([xn, *outs], [gres, *go], retain_graph=True))
    print(json.dumps({k: round(v, 3) for k, v in res.items()}))


if __name__ == "__main__":
    main()
