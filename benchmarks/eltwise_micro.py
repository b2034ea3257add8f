"""Per-kernel timing of the training step's streaming element-wise kernels at the benchmark shape (B x 2624 tokens, C = 2048, FFN 8192) through
the C-ABI launchers, one kernel per timed call where the launcher allows it; effective HBM rate for the algorithmic bytes.
VRWKV_HIP_LIB selects an experiment build (benchmarks/build_alt_src.sh).   python benchmarks/eltwise_micro.py [B]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd import fused  # noqa: E402
from visualrwkv_amd.hip_lib import load  # noqa: E402


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
    lib = load()
    n = B * T * C
    bf = lambda *sh: torch.randn(*sh, device=dev, dtype=torch.bfloat16)
    res = {}

    def rec(name, ms, nbytes):
        res[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}

    st = torch.cuda.current_stream().cuda_stream
    # relu^2 (FFN width 4C)
    h, dy = bf(B * T, 4 * C), bf(B * T, 4 * C)
    y = torch.empty_like(h)
    rec("relusq_fwd", timeit(lambda: lib.vrwkv_relusq_fwd_bf16(4 * n, h.data_ptr(), y.data_ptr(), st)), 4 * n * 4)
    rec("relusq_bwd", timeit(lambda: lib.vrwkv_relusq_bwd_bf16(4 * n, h.data_ptr(), dy.data_ptr(), y.data_ptr(), st)), 4 * n * 6)
    del h, dy, y
    # AdamW on a 100 M element bucket
    ne = 100 * 1024 * 1024
    master, m, v = [torch.zeros(ne, device=dev) for _ in range(3)]
    g, p = torch.zeros(ne, device=dev, dtype=torch.bfloat16), torch.zeros(ne, device=dev, dtype=torch.bfloat16)
    rec("adamw", timeit(lambda: lib.vrwkv_adamw_step_bf16(ne, master.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), p.data_ptr(), 1e-4, 0.9, 0.99,
                                                           1e-8, 0.0, 1, 1.0, 0, 0, st)), ne * 28)
    del master, m, v, g, p
    x = bf(B, T, C)
    mus = [torch.rand(1, 1, C, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(6)]
    gouts = [torch.randn_like(x) for _ in range(6)]
    with torch.no_grad():
        rec("mix6_fwd", timeit(lambda: fused.mix(x, *[q.detach() for q in mus])), n * 14)
    xr = x.clone().requires_grad_(True)
    outs = fused.mix(xr, *mus)
    rec("mix6_bwd(+colsum)", timeit(lambda: torch.autograd.grad(outs, [xr, *mus], gouts, retain_graph=True)), n * 16)
    del outs
    kk_, vv_, vf_, vl_, al_ = [bf(B, T, C).requires_grad_(True) for _ in range(5)]
    kp = [bf(1, 1, C).mul_(0.5).requires_grad_(True) for _ in range(4)]
    with torch.no_grad():
        rec("kva_fwd", timeit(lambda: fused.kva(kk_.detach(), vv_.detach(), vf_.detach(), vl_.detach(), al_.detach(), *[q.detach() for q in kp])), n * 2 * 9)
    kouts = fused.kva(kk_, vv_, vf_, vl_, al_, *kp, True)
    kg = [gouts[i % 6] for i in range(len(kouts))]
    rec("kva_bwd(+colsum)", timeit(lambda: torch.autograd.grad(kouts, [kk_, vv_, vf_, vl_, al_, *kp], kg, retain_graph=True)), n * 2 * 14)
    del kouts
    yy, rr, gg = [bf(B, T, C).requires_grad_(True) for _ in range(3)]
    lw, lb_, rk_ = [bf(C).mul_(0.5).requires_grad_(True) for _ in range(3)]
    with torch.no_grad():
        rec("post_fwd", timeit(lambda: fused.post(yy.detach(), rr.detach(), kk_.detach(), vv_.detach(), gg.detach(), lw.detach(), lb_.detach(),
                                                  rk_.detach().view(C // 64, 64), 64e-5)), n * 2 * 6)
    pout = fused.post(yy, rr, kk_, vv_, gg, lw, lb_, rk_.view(C // 64, 64), 64e-5)
    rec("post_bwd(+colsum)", timeit(lambda: torch.autograd.grad(pout, [yy, rr, kk_, vv_, gg, lw, lb_, rk_], gouts[0], retain_graph=True)), n * 2 * 11)
    del pout, yy, rr, gg, kk_, vv_, vf_, vl_, al_
    ln = torch.nn.LayerNorm(C).to(dev).bfloat16()
    delta, gres = torch.randn_like(x), torch.randn_like(x)
    for M in (6, 1):
        ms = mus[:M]
        with torch.no_grad():
            rec(f"ln_mix{M}_fwd", timeit(lambda: fused.add_ln_mix(x, delta, ln, ms, M == 6)), n * 2 * (3 + M))
        xg, dg = x.clone().requires_grad_(True), delta.clone().requires_grad_(True)
        xn, outs = fused.add_ln_mix(xg, dg, ln, ms, M == 6)
        go = [gouts[i % 6] for i in range(len(outs))]
        rec(f"ln_mix{M}_bwd(all kernels)", timeit(lambda: torch.autograd.grad([xn, *outs], [xg, dg, ln.weight, ln.bias, *ms], [gres, *go], retain_graph=True)), n * 2 * (4 + len(outs)))
        del xn, outs
    with torch.no_grad():
        rec("add_ln_fwd", timeit(lambda: fused.add_ln(x, delta, ln)), n * 2 * 4)
    xg, dg = x.clone().requires_grad_(True), delta.clone().requires_grad_(True)
    xn, yo = fused.add_ln(xg, dg, ln)
    rec("add_ln_bwd(+colsum)", timeit(lambda: torch.autograd.grad([xn, yo], [xg, dg, ln.weight, ln.bias], [gres, gouts[0]], retain_graph=True)), n * 2 * 4)
    print(json.dumps({"B": B, "lib": os.environ.get("VRWKV_HIP_LIB", "default"), **res}))


if __name__ == "__main__":
    main()
