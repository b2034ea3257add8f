"""Generate tests/golden/wkv7_simple_ref.pt by RUNNING the reference's own file
VisualRWKV-v6/v6.xx/RWKV-v7_simple.py (the only kernel-independent statement of the WKV7
recurrence in the reference) in this container.

The reference script hard-codes B,H,N,T = 2,3,4,5 and draws its inputs with torch.randn /
torch.ones.  We execute the file unmodified with `runpy`, but hand it a `torch` whose
randn/ones return *seeded leaf tensors that require grad*, so that after the script ends we
can (a) read its inputs and its `out`, and (b) call autograd through the script's own graph
to obtain reference gradients.  Nothing from the reference is copied into this repo: only
the numeric inputs/outputs are stored.

Run (only where /root/reference exists):  python tests/golden/make_golden_wkv7.py
"""
import os
import runpy
import sys

import torch

REF = "/root/reference/VisualRWKV-v6/v6.xx/RWKV-v7_simple.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wkv7_simple_ref.pt")


def main():
    g = torch.Generator().manual_seed(20260926)
    leaves = []
    real_randn, real_ones = torch.randn, torch.ones

    def leaf(*shape, **kw):
        # float64 leaf so the script's .double() is an identity op in the graph
        x = real_randn(*shape, generator=g, dtype=torch.float64)
        if len(leaves) == 1:          # 2nd tensor the script creates is `w` (torch.ones)
            x = -torch.nn.functional.softplus(-x) - 0.5
        x.requires_grad_(True)
        leaves.append(x)
        return x

    torch.randn = lambda *s, **kw: leaf(*s)
    torch.ones = lambda *s, **kw: leaf(*s)
    try:
        ns = runpy.run_path(REF)
    finally:
        torch.randn, torch.ones = real_randn, real_ones
    # creation order in the script: r, w, k, v, a, b   (RWKV-v7_simple.py:8-13)
    r, w, k, v, a, b = leaves
    B, T, H, N = r.shape
    out = ns["out"]                                  # (B,T,H*N)
    dy = real_randn(B, T, H * N, generator=g, dtype=torch.float64)
    grads = torch.autograd.grad(out, [r, w, k, v, a, b], grad_outputs=dy)
    torch.save({
        "source": "RWKV-v7_simple.py executed unmodified via runpy",
        "r": r.detach(), "w_raw": w.detach(), "k": k.detach(), "v": v.detach(),
        "a": a.detach(), "b": b.detach(), "out": out.detach(), "dy": dy,
        "dr": grads[0], "dw_raw": grads[1], "dk": grads[2], "dv": grads[3],
        "da": grads[4], "db": grads[5],
    }, OUT)
    print("wrote", OUT, "out", tuple(out.shape))


if __name__ == "__main__":
    sys.exit(main())
