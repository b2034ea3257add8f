"""CPU restatement of one RWKV-7 block (time-mix + channel-mix) around the WKV7 oracle.
TEST INFRASTRUCTURE ONLY (parity checker for the GPU modules, and bench.py's cpu_baseline leg).

Follows VisualRWKV-v7/v7.00/src/model.py:163-195 (RWKV_Tmix_x070.forward), :221-227
(RWKV_CMix_x070.forward), :247-254 (Block.forward) on plain tensors taken from a module's state
(parameter names as in the reference).  The WKV7 call goes to the C oracle through an autograd
Function with the reference's bf16 contract (WindBackstepping, model.py:45-65)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import wkv7_c


class OracleWKV7(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, q, k, v, z, b):
        ins = [x.to(torch.bfloat16).contiguous() for x in (w, q, k, v, z, b)]
        y, s, sa = wkv7_c.forward(*ins)
        ctx.save_for_backward(*ins, s, sa)
        ctx.in_dtype = w.dtype
        return y.to(w.dtype)

    @staticmethod
    def backward(ctx, dy):
        *ins, s, sa = ctx.saved_tensors
        grads = wkv7_c.backward(*ins, dy.to(torch.bfloat16).contiguous(), s, sa)
        return tuple(g.to(ctx.in_dtype) for g in grads)


def run_wkv7(r, w, k, v, a, b):
    B, T, HC = r.shape
    r, w, k, v, a, b = [x.view(B, T, HC // 64, 64) for x in (r, w, k, v, a, b)]
    return OracleWKV7.apply(w, r, k, v, a, b).view(B, T, HC)       # (w,q,...) re-order of model.py:69-70


def shift(x):
    return F.pad(x, (0, 0, 1, -1))


def tmix(P, x, v_first, layer_id, n_head, eps):
    """P: dict of the att.* tensors (x_r ... ln_x.weight/bias)."""
    B, T, C = x.shape
    H = n_head
    xx = shift(x) - x
    xr, xw, xk, xv, xa, xg = [x + xx * P[n] for n in ("x_r", "x_w", "x_k", "x_v", "x_a", "x_g")]
    r = xr @ P["receptance.weight"].t()
    w = -F.softplus(-(P["w0"] + torch.tanh(xw @ P["w1"]) @ P["w2"])) - 0.5
    k = xk @ P["key.weight"].t()
    v = xv @ P["value.weight"].t()
    if layer_id == 0:
        v_first = v
    else:
        v = v + (v_first - v) * torch.sigmoid(P["v0"] + (xv @ P["v1"]) @ P["v2"])
    a = torch.sigmoid(P["a0"] + (xa @ P["a1"]) @ P["a2"])
    g = torch.sigmoid(xg @ P["g1"]) @ P["g2"]
    kk = F.normalize((k * P["k_k"]).view(B, T, H, -1), dim=-1, p=2.0).view(B, T, C)
    k = k * (1 + (a - 1) * P["k_a"])
    y = run_wkv7(r, w, k, v, -kk, kk * a)
    y = F.group_norm(y.view(B * T, C), H, P["ln_x.weight"], P["ln_x.bias"], eps).view(B, T, C)
    y = y + ((r.view(B, T, H, -1) * k.view(B, T, H, -1) * P["r_k"]).sum(dim=-1, keepdim=True) * v.view(B, T, H, -1)).view(B, T, C)
    return (y * g) @ P["output.weight"].t(), v_first


def cmix(P, x):
    xx = shift(x) - x
    k = x + xx * P["x_k"]
    k = torch.relu(k @ P["key.weight"].t()) ** 2
    return k @ P["value.weight"].t()


def block(state, prefix, x, v_first, layer_id, n_head, eps=64e-5):
    """state: flat dict with keys like f'{prefix}att.x_r'.  Returns (x, v_first)."""
    sub = lambda p: {k[len(prefix + p):]: v for k, v in state.items() if k.startswith(prefix + p)}
    C = x.shape[-1]
    if layer_id == 0:
        x = F.layer_norm(x, (C,), state[prefix + "ln0.weight"], state[prefix + "ln0.bias"])
    h = F.layer_norm(x, (C,), state[prefix + "ln1.weight"], state[prefix + "ln1.bias"])
    y, v_first = tmix(sub("att."), h, v_first, layer_id, n_head, eps)
    x = x + y
    h = F.layer_norm(x, (C,), state[prefix + "ln2.weight"], state[prefix + "ln2.bias"])
    return x + cmix(sub("ffn."), h), v_first
