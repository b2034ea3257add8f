"""Single-token decode step of a Block (RWKV_Tmix_x070 + RWKV_CMix_x070, src/model.py:166-194,221-227,247-254) with
carried state: 5 launches per layer instead of the ~25 the module-level stateful path issues.  Every nn.Linear and
first-stage LoRA product is a row of a batched GEMV launch with its activation / residual in the epilogue
(csrc/gemv_decode.hip); LayerNorm + token shift + lerps are folded into the GEMV that consumes them (a separate kernel
for widths the fold does not cover), and everything that is per-head -- the second
LoRA stage, decay, k/v/a glue, the WKV7 state step, GroupNorm + bonus + gate -- is another (csrc/decode_fused.hip).
Inference only; first-stage LoRA factors are read through cached (N,K)-major copies (rebuilt when a parameter is
modified in place)."""
from __future__ import annotations

import ctypes

import torch

from . import hip_lib

ACT_NONE, ACT_TANH, ACT_SIGMOID, ACT_RELUSQ = 0, 1, 2, 3
MAX_B = 4


def _stream(dev):
    return hip_lib.launch_stream(dev)


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
        _stream(device))
    hip_lib.check(rc, "vrwkv_gemv_multi_bf16")
    return ys


def gemv_multi_copy(jobs, B, device, copy_src, copy_dst):
    """gemv_multi plus the side job copy_dst[:] = copy_src (the carried token-shift row, see gemv_ln_multi)."""
    n = len(jobs)
    ys = [torch.empty(B, W.shape[0], dtype=torch.bfloat16, device=device) for W, _, _, _ in jobs]
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    rc = hip_lib.load().vrwkv_gemv_multi_copy_bf16(
        n, B, vp(*[W.data_ptr() for W, _, _, _ in jobs]), vp(*[x.data_ptr() for _, x, _, _ in jobs]),
        vp(*[(r.data_ptr() if r is not None else 0) for _, _, r, _ in jobs]), vp(*[y.data_ptr() for y in ys]),
        ip(*[W.shape[0] for W, _, _, _ in jobs]), ip(*[W.shape[1] for W, _, _, _ in jobs]), ip(*[a for _, _, _, a in jobs]),
        copy_src.data_ptr(), copy_dst.data_ptr(), copy_src.numel(), _stream(device))
    hip_lib.check(rc, "vrwkv_gemv_multi_copy_bf16")
    return ys


def ln_fold_supported(C):
    return 512 <= C <= 4096 and C % 8 == 0


def gemv_ln_multi(jobs, x, ln, x_prev):
    """jobs: list of (W (N,K), mu (K), act); x (B,K) raw residual row.  y_j = act_j(W_j (h + (x_prev - h) mu_j)) with
    h = LayerNorm(x), one launch.  Returns (ys, h); x_prev is NOT updated (the caller hands h to a later launch's
    copy side job, because workgroups of this launch read x_prev until it ends)."""
    B, K = x.shape
    n = len(jobs)
    ys = [torch.empty(B, W.shape[0], dtype=torch.bfloat16, device=x.device) for W, _, _ in jobs]
    h = torch.empty_like(x)
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    rc = hip_lib.load().vrwkv_gemv_ln_multi_bf16(
        n, B, K, vp(*[W.data_ptr() for W, _, _ in jobs]), x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps),
        x_prev.data_ptr(), vp(*[m.data_ptr() for _, m, _ in jobs]), h.data_ptr(), vp(*[y.data_ptr() for y in ys]),
        ip(*[W.shape[0] for W, _, _ in jobs]), ip(*[a for _, _, a in jobs]), _stream(x.device))
    hip_lib.check(rc, "vrwkv_gemv_ln_multi_bf16")
    return ys, h


def ln_mix(x, ln, x_prev, mus):
    """x (B,C) -> [LN(x) + (x_prev - LN(x)) * mu for mu in mus]; x_prev (B,C) is replaced by LN(x) in place."""
    B, C = x.shape
    n = len(mus)
    outs = [torch.empty_like(x) for _ in mus]
    vp = ctypes.c_void_p * n
    rc = hip_lib.load().vrwkv_decode_ln_mix_bf16(B, C, n, x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps),
                                                 x_prev.data_ptr(), vp(*[m.data_ptr() for m in mus]),
                                                 vp(*[o.data_ptr() for o in outs]), _stream(x.device))
    hip_lib.check(rc, "vrwkv_decode_ln_mix_bf16")
    return outs


def tmix_head(m, r, k, v, v_first, hidden, S, carry=None):
    """Everything of RWKV_Tmix_x070.forward between the first-stage products and the output projection, for one token.
    hidden: [tanh(xw w1), xa a1, sigmoid(xg g1)] (+ [xv v1] on layers > 0), each (B,D); S (B,H,64,64) fp32, in place.
    carry: optional (src, dst) pair of (B,C) rows copied as a side job."""
    B, C = r.shape
    names = ("w2", "a2", "g2") + (("v2",) if v_first is not None else ())
    t = _transposed(m, names, "_decode_cache2")
    W2 = [t[n] for n in names]                       # (C, D): a channel's D weights are contiguous
    n = len(W2)
    out = torch.empty_like(r)
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    rc = hip_lib.load().vrwkv_decode_tmix_head_bf16(
        B, C // 64, r.data_ptr(), k.data_ptr(), v.data_ptr(), v_first.data_ptr() if v_first is not None else 0,
        vp(*[h.data_ptr() for h in hidden]), vp(*[w.data_ptr() for w in W2]), ip(*[w.shape[1] for w in W2]),
        m.w0.data_ptr(), m.a0.data_ptr(), m.v0.data_ptr() if v_first is not None else 0, m.k_k.data_ptr(), m.k_a.data_ptr(),
        m.r_k.data_ptr(), m.ln_x.weight.data_ptr(), m.ln_x.bias.data_ptr(), float(m.ln_x.eps), S.data_ptr(), out.data_ptr(),
        carry[0].data_ptr() if carry else 0, carry[1].data_ptr() if carry else 0, _stream(r.device))
    hip_lib.check(rc, "vrwkv_decode_tmix_head_bf16")
    return out


def _transposed(m, names, slot="_decode_cache"):
    """(N,K)-major copies of the LoRA factors used as `x @ p`; cached on the module, keyed by the parameters' versions."""
    from . import param_state
    key = (param_state.generation(),) + tuple((getattr(m, n).data_ptr(), getattr(m, n)._version) for n in names)
    cache = getattr(m, slot, None)
    if cache is None or cache[0] != key:
        cache = (key, {n: getattr(m, n).detach().t().contiguous() for n in names})
        setattr(m, slot, cache)
    return cache[1]


def supported(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and x.shape[1] == 1 and x.shape[0] <= MAX_B


def _bf16_contig(m, names):
    return all(getattr(m, n).dtype == torch.bfloat16 and getattr(m, n).is_contiguous() for n in names)


@torch.no_grad()
def block_decode(block, x, v_first, state):
    """One Block (src/model.py:247-254) for one token: x (B,1,C) residual stream in, residual stream out."""
    B, _, C = x.shape
    dev = x.device
    att, ffn, lid = block.att, block.ffn, block.layer_id
    assert _bf16_contig(att, ("w0", "a0", "k_k", "k_a", "r_k")), "decode step: bf16 contiguous parameters"
    if lid == 0:
        x = block.ln0(x)
    x = x.reshape(B, C)
    t = _transposed(att, ("w1", "a1", "g1") + (("v1",) if lid > 0 else ()))
    fold = ln_fold_supported(C)                  # LayerNorm + shift + lerps inside the consuming GEMV launch
    mats = [(att.receptance.weight, att.x_r, ACT_NONE), (att.key.weight, att.x_k, ACT_NONE), (att.value.weight, att.x_v, ACT_NONE),
            (t["w1"], att.x_w, ACT_TANH), (t["a1"], att.x_a, ACT_NONE), (t["g1"], att.x_g, ACT_SIGMOID)]
    if lid > 0:
        mats.append((t["v1"], att.x_v, ACT_NONE))
    carry = None
    if fold:
        outs, h = gemv_ln_multi(mats, x, block.ln1, state.att_x[lid])
        carry = (h, state.att_x[lid])
    else:
        ins = dict(zip(("x_r", "x_w", "x_k", "x_v", "x_a", "x_g"),
                       ln_mix(x, block.ln1, state.att_x[lid], [att.x_r, att.x_w, att.x_k, att.x_v, att.x_a, att.x_g])))
        names = ["x_r", "x_k", "x_v", "x_w", "x_a", "x_g", "x_v"]
        outs = gemv_multi([(W, ins[nm], None, act) for (W, _, act), nm in zip(mats, names)], B, dev)
    r, k, v = outs[:3]
    if lid == 0:
        v_first = v.view(B, 1, C)
    y = tmix_head(att, r, k, v, v_first.view(B, C) if lid > 0 else None, outs[3:], state.S[lid], carry)
    (x,) = gemv_multi([(att.output.weight, y, x, ACT_NONE)], B, dev)                                # x + output(y)
    if fold:
        (kk,), h = gemv_ln_multi([(ffn.key.weight, ffn.x_k, ACT_RELUSQ)], x, block.ln2, state.ffn_x[lid])
        (x2,) = gemv_multi_copy([(ffn.value.weight, kk, x, ACT_NONE)], B, dev, h, state.ffn_x[lid])
    else:
        (kx,) = ln_mix(x, block.ln2, state.ffn_x[lid], [ffn.x_k])
        (kk,) = gemv_multi([(ffn.key.weight, kx, None, ACT_RELUSQ)], B, dev)
        (x2,) = gemv_multi([(ffn.value.weight, kk, x, ACT_NONE)], B, dev)                            # x + value(relu(key)^2)
    return x2.view(B, 1, C), v_first


@torch.no_grad()
def head_decode(rwkv, x):
    """ln_out + head for one token: x (B,1,C) -> logits (B,V).  The head is one GEMV over the (V,C) matrix."""
    B, _, C = x.shape
    h = rwkv.ln_out(x).view(B, C)
    (logits,) = gemv_multi([(rwkv.head.weight, h, None, ACT_NONE)], B, x.device)
    return logits
