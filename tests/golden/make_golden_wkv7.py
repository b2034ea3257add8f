"""Generate tests/golden/wkv7_simple_ref.pt by RUNNING the reference's own file
VisualRWKV-v6/v6.xx/RWKV-v7_simple.py (the only kernel-independent statement of the WKV7
recurrence in the reference) in this container.

The reference script hard-codes B,H,N,T = 2,3,4,5 and draws its inputs with torch.randn /
torch.ones.  We execute the file unmodified with `runpy`, but hand it a `torch` whose
randn/ones return *seeded leaf tensors that require grad*, so that after the script ends we
can (a) read its inputs and its `out`, and (b) call autograd through the script's own graph
to obtain reference gradients.  Nothing from the reference is copied into this repo: only
the numeric inputs/outputs are stored.

A second fixture pins the oracle DIRECTLY at the kernels' head size and across 16-token checkpoint boundaries
(wkv7_simple_n64_ref.pt): the recurrence loop of the same file (RWKV-v7_simple.py:20-32, the `for t in range(T):`
statement) is extracted from the reference's source text with `ast` at generation time and executed, unmodified, in fp64
at (B,T,H,N) = (1,48,2,64) on inputs with the model's structure (oracle.wkv7_oracle.make_inputs: bf16-representable, so
the bf16 kernels and the C oracle see the same numbers); gradients come from autograd through that loop.  Only tensors
and a provenance string are stored.

Run (only where /root/reference exists):  python tests/golden/make_golden_wkv7.py
"""
import ast
import os
import runpy
import sys

import torch

REF = "/root/reference/VisualRWKV-v6/v6.xx/RWKV-v7_simple.py"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "wkv7_simple_ref.pt")
OUT64 = os.path.join(HERE, "wkv7_simple_n64_ref.pt")


def extract_recurrence_loop(path):
    """Source text of the module-level `for t in range(T):` loop of the reference script."""
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Call) and getattr(node.iter.func, "id", "") == "range":
            return ast.get_source_segment(src, node), node.lineno, node.end_lineno
    raise RuntimeError("recurrence loop not found in " + path)


def make_n64(B=1, T=48, H=2, N=64, seed=42):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.wkv7_oracle import make_inputs            # structured inputs of SURVEY.md 8(d), bf16
    loop_src, l0, l1 = extract_recurrence_loop(REF)
    w_raw, q, k, v, z, a, dy = make_inputs(B, T, H, N=N, seed=seed)
    leaves = [x.double().clone().requires_grad_(True) for x in (q, w_raw, k, v, z, a)]
    r_, wraw_, k_, v_, a_, b_ = leaves                    # the script's names: r, w, k, v, a (= op's z), b (= op's a)
    ns = {"torch": torch, "T": T, "r": r_, "k": k_, "v": v_, "a": a_, "b": b_,
          "w": torch.exp(-torch.exp(wraw_)),             # RWKV-v7_simple.py:15
          "out": torch.zeros(B, T, H, N, dtype=torch.float64),          # :17
          "state": torch.zeros(B, H, N, N, dtype=torch.float64)}        # :18
    exec(compile(loop_src, REF + ":loop", "exec"), ns)
    out, state = ns["out"], ns["state"]
    grads = torch.autograd.grad(out, leaves, grad_outputs=dy.double())
    torch.save({
        "provenance": f"RWKV-v7_simple.py:{l0}-{l1} (the recurrence loop, ast-extracted and executed unmodified, fp64) "
                      f"at B,T,H,N={B},{T},{H},{N}; inputs oracle.wkv7_oracle.make_inputs(seed={seed})",
        "w_raw": w_raw, "q": q, "k": k, "v": v, "z": z, "a": a, "dy": dy,            # bf16, op-schema names
        "out": out.detach(), "final_state": state.detach(),
        "dq": grads[0], "dw_raw": grads[1], "dk": grads[2], "dv": grads[3], "dz": grads[4], "da": grads[5],
    }, OUT64)
    print("wrote", OUT64, "out", tuple(out.shape))


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
    make_n64()


if __name__ == "__main__":
    sys.exit(main())
