"""Pin the WKV6 oracle to the REFERENCE'S OWN pure-PyTorch statements of the RWKV-6 recurrence.

Two functions of the reference are executed here, unmodified, by extracting their source text from the reference
files at generation time (nothing of it is stored in this repository -- only the tensors they compute):

  1. `naive_recurrent_rwkv6_fla` + `run_naive_recurrent_fla`   VisualRWKV-v6/v6.xx/test_kernel.py:175-215
     (the file itself cannot be imported: it JIT-builds CUDA and imports `fla` at module level);
  2. the per-token loop of `RWKV.att_seq_v6_0`                  VisualRWKV-v7/v7.00/app/modeling_rwkv.py:891-897
     (the CPU sequence path of the demo app), as a cross-check of (1).

Inputs follow the reference's own test procedure (test_kernel.py:43-55: seed 42, r,k,v,u ~ U(-1,1), w ~ U(-8,1),
bf16-rounded), at a size the CPU handles; outputs and autograd gradients (loss = sum(y*y - tanh(y)), test_kernel.py:36)
are recorded in fp64.

Run where /root/reference exists:   python tests/golden/make_golden_wkv6.py
"""
import ast
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TEST = "/root/reference/VisualRWKV-v6/v6.xx/test_kernel.py"
REF_APP = "/root/reference/VisualRWKV-v7/v7.00/app/modeling_rwkv.py"


def extract_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            out[node.name] = ast.get_source_segment(src, node)
    assert set(out) == set(names), (names, list(out))
    return out


def extract_loop(path, func_name):
    """Source of the `for t in range(T):` loop inside method `func_name` (dedented)."""
    src = open(path).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == func_name:
            for sub in ast.walk(node):
                if isinstance(sub, ast.For) and isinstance(sub.iter, ast.Call) and getattr(sub.iter.func, "id", "") == "range":
                    seg = ast.get_source_segment(src, sub)
                    lines = seg.split("\n")
                    indent = len(lines[1]) - len(lines[1].lstrip()) - 4
                    return "\n".join([lines[0]] + [l[indent:] for l in lines[1:]])
    raise RuntimeError("loop not found")


def main():
    fns = extract_functions(REF_TEST, ["naive_recurrent_rwkv6_fla", "run_naive_recurrent_fla"])
    ns = {"torch": torch, "Optional": __import__("typing").Optional}
    for name in ("naive_recurrent_rwkv6_fla", "run_naive_recurrent_fla"):
        exec(compile(fns[name], REF_TEST + ":" + name, "exec"), ns)

    B, T, H, N = 2, 32, 2, 64
    C = H * N
    torch.manual_seed(42)
    mk = lambda lo, hi, *shape: torch.empty(*shape).uniform_(lo, hi).bfloat16().double()
    r, k, v = mk(-1, 1, B, T, C), mk(-1, 1, B, T, C), mk(-1, 1, B, T, C)
    w = mk(-8, 1, B, T, C)
    u = mk(-1, 1, H, N)
    s0 = mk(-1, 1, B, H, N, N)
    gy = mk(-1, 1, B, T, C)            # upstream gradient, bf16-representable so that the bf16 GPU op sees the same numbers
    out = {"provenance": "VisualRWKV-v6/v6.xx/test_kernel.py:175-215 naive_recurrent_rwkv6_fla / run_naive_recurrent_fla "
                         "(executed unmodified, fp64 inputs); cross-check VisualRWKV-v7/v7.00/app/modeling_rwkv.py:891-897",
           "B": B, "T": T, "H": H, "N": N, "r": r, "k": k, "v": v, "w": w, "u": u}
    for tag, state in (("zero_state", None), ("with_state", s0)):
        leaves = [x.clone().requires_grad_(True) for x in (r, k, v, w, u)]
        st = state.clone().requires_grad_(True) if state is not None else None
        # the reference function converts to float32 internally (`x.float()`): keep fp64 by making .float() a no-op view
        y, fin = ns["run_naive_recurrent_fla"](B, T, C, H, *[_F64(x) for x in leaves], _F64(st) if st is not None else None)
        y = y.as_subclass(torch.Tensor)
        loss = (y * gy).sum()
        loss.backward()
        out[tag] = {"s0": state, "y": y.detach().clone(), "final_state": fin.detach().as_subclass(torch.Tensor).clone(),
                    "gy": gy.clone(),
                    "gr": leaves[0].grad.clone(), "gk": leaves[1].grad.clone(), "gv": leaves[2].grad.clone(),
                    "gw": leaves[3].grad.clone(), "gu": leaves[4].grad.clone(),
                    "gs0": st.grad.clone() if st is not None else None}

    # cross-check with the demo app's CPU loop (one sample; its layout: r (H,T,N), k (H,N,T), v (H,T,N), w (T,H,N,1), s (H,N,N))
    loop_src = extract_loop(REF_APP, "att_seq_v6_0")
    b = 0
    f = lambda x: x[b].view(T, H, N)
    env = {"torch": torch, "T": T, "matmul": torch.matmul,
           "r": f(r).transpose(0, 1).contiguous(), "k": f(k).permute(1, 2, 0).contiguous(), "v": f(v).transpose(0, 1).contiguous(),
           "w": torch.exp(-torch.exp(f(w))).view(T, H, N, 1), "t_first": u.view(H, N, 1), "s": torch.zeros(H, N, N, dtype=torch.float64),
           "out": torch.empty(T, H, N, dtype=torch.float64)}
    exec(compile(loop_src, REF_APP + ":att_seq_v6_0.loop", "exec"), env)
    app_y = env["out"].reshape(T, C)
    err = (app_y - out["zero_state"]["y"][b]).norm() / out["zero_state"]["y"][b].norm()
    assert err < 1e-12, err
    out["app_loop_vs_naive_rel_err"] = float(err)
    torch.save(out, os.path.join(HERE, "wkv6_naive_ref.pt"))
    print("wrote wkv6_naive_ref.pt; app loop vs naive:", float(err))


class _F64(torch.Tensor):
    """fp64 tensor whose .float() stays fp64 (the reference function up-casts its inputs with .float())."""

    @staticmethod
    def __new__(cls, x):
        return x.as_subclass(cls)

    def float(self):
        return self


if __name__ == "__main__":
    sys.exit(main())
