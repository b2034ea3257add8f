"""Single-token decode step of RWKV_Tmix_x070 / RWKV_CMix_x070 with carried state, built from batched GEMV launches
(csrc/gemv_decode.hip) and the fused glue kernels: 13 launches per layer instead of the ~25 the module-level stateful
path issues (every nn.Linear and LoRA product of src/model.py:175-194,222-225 is a GEMV at T = 1 and the activations
tanh / sigmoid / relu^2 and the output residual ride in the GEMV epilogue).  Inference only; weights are read through
cached, pre-transposed copies of the LoRA factors (rebuilt when a parameter is modified in place)."""
from __future__ import annotations

import ctypes

import torch

from . import fused, hip_lib

ACT_NONE, ACT_TANH, ACT_SIGMOID, ACT_RELUSQ = 0, 1, 2, 3
MAX_B = 4


def gemv_multi(jobs, B, device):
    """jobs: list of (W (N,K), x (B,K), res (B,N) or None, act) -> list of y (B,N) bf16, one launch."""
    n = len(jobs)
    ys = [torch.empty(B, W.shape[0], dtype=torch.bfloat16, device=device) for W, _, _, _ in jobs]
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    rc = hip_lib.load().vrwkv_gemv_multi_bf16(
        n, B, vp(*[W.data_ptr() for W, _, _, _ in jobs]), vp(*[x.data_ptr() for _, x, _, _ in jobs]),
        vp(*[(r.data_ptr() if r is not None else 0) for _, _, r, _ in jobs]), vp(*[y.data_ptr() for y in ys]),
        ip(*[W.shape[0] for W, _, _, _ in jobs]), ip(*[W.shape[1] for W, _, _, _ in jobs]), ip(*[a for _, _, _, a in jobs]),
        torch.cuda.current_stream(device).cuda_stream)
    hip_lib.check(rc, "vrwkv_gemv_multi_bf16")
    return ys


def _transposed(m, names):
    """(N,K)-major copies of the LoRA factors used as `x @ p`; cached on the module, keyed by the parameters' versions."""
    key = tuple((getattr(m, n).data_ptr(), getattr(m, n)._version) for n in names)
    cache = getattr(m, "_decode_cache", None)
    if cache is None or cache[0] != key:
        cache = (key, {n: getattr(m, n).detach().t().contiguous() for n in names})
        m._decode_cache = cache
    return cache[1]


def supported(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and x.shape[1] == 1 and x.shape[0] <= MAX_B


@torch.no_grad()
def tmix_decode(m, x, v_first, state):
    """x (B,1,C): one token through RWKV_Tmix_x070 continuing from `state`; returns (out (B,1,C), v_first)."""
    lid = m.layer_id
    B, _, C = x.shape
    dev = x.device
    t = _transposed(m, ("w1", "w2", "a1", "a2", "g1", "g2") + (("v1", "v2") if lid > 0 else ()))
    xr, xw, xk, xv, xa, xg = [o.view(B, C) for o in fused.mix_prev(x, state.att_x[lid], m.x_r, m.x_w, m.x_k, m.x_v, m.x_a, m.x_g)]
    state.att_x[lid].copy_(x[:, 0])
    jobs = [(m.receptance.weight, xr, None, ACT_NONE), (m.key.weight, xk, None, ACT_NONE), (m.value.weight, xv, None, ACT_NONE),
            (t["w1"], xw, None, ACT_TANH), (t["a1"], xa, None, ACT_NONE), (t["g1"], xg, None, ACT_SIGMOID)]
    if lid > 0:
        jobs.append((t["v1"], xv, None, ACT_NONE))
    outs = gemv_multi(jobs, B, dev)
    r, k, v, hw, ha, hg = outs[:6]
    jobs2 = [(t["w2"], hw, None, ACT_NONE), (t["a2"], ha, None, ACT_NONE), (t["g2"], hg, None, ACT_NONE)]
    if lid > 0:
        jobs2.append((t["v2"], outs[6], None, ACT_NONE))
    outs2 = gemv_multi(jobs2, B, dev)
    sh = (B, 1, C)
    w = fused.decay(outs2[0].view(sh), m.w0)
    al, g = outs2[1].view(sh), outs2[2].view(sh)
    k, v, r = k.view(sh), v.view(sh), r.view(sh)
    if lid == 0:
        v_first = v
        k2, z, b = fused.kva(k, None, None, None, al, m.k_k, m.k_a, m.a0, None)
        v2 = v
    else:
        k2, v2, z, b = fused.kva(k, v, v_first, outs2[3].view(sh), al, m.k_k, m.k_a, m.a0, m.v0)
    y = state.wkv(lid, r, w, k2, v2, z, b)
    y = fused.post(y, r, k2, v2, g, m.ln_x.weight, m.ln_x.bias, m.r_k, m.ln_x.eps)
    return y, v_first                                    # the output projection is applied by the caller with the residual


@torch.no_grad()
def block_decode(block, x, v_first, state):
    """One Block (src/model.py:247-254) for one token: x (B,1,C) residual stream in, residual stream out."""
    B, _, C = x.shape
    dev = x.device
    att, ffn = block.att, block.ffn
    if block.layer_id == 0:
        x = block.ln0(x)
    h = block.ln1(x)
    y, v_first = tmix_decode(att, h, v_first, state)
    (x,) = gemv_multi([(att.output.weight, y.view(B, C), x.view(B, C), ACT_NONE)], B, dev)         # x + output(y)
    x = x.view(B, 1, C)
    h = block.ln2(x)
    (kx,) = fused.mix_prev(h, state.ffn_x[block.layer_id], ffn.x_k)
    state.ffn_x[block.layer_id].copy_(h[:, 0])
    (kk,) = gemv_multi([(ffn.key.weight, kx.view(B, C), None, ACT_RELUSQ)], B, dev)
    (x2,) = gemv_multi([(ffn.value.weight, kk, x.view(B, C), ACT_NONE)], B, dev)                     # x + value(relu(key)^2)
    return x2.view(B, 1, C), v_first
