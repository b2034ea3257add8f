"""CPU oracle for the WKV7 ("wind_backstepping") operator.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (visualrwkv_amd) never does: it fails loudly when the HIP
library is missing.

Three restatements of the reference math, all on CPU tensors:

* ``wkv7_forward_ref`` / ``wkv7_backward_ref`` -- literal restatements of the reference
  CUDA kernels ``forward_kernel`` (VisualRWKV-v7/v7.00/cuda/wkv7_cuda.cu:10-52) and
  ``backward_kernel`` (same file :54-130): same recurrence order, same fp32 state, same
  16-token transposed checkpoints, same inverse-decay "un-step" in the backward.  They
  are vectorised over (B, H) but keep the per-thread summation structure (one sum over
  j per output element).
* ``wkv7_naive`` -- the kernel-independent statement of the recurrence, following
  VisualRWKV-v6/v6.xx/RWKV-v7_simple.py:15-32 (state[b,h,i,j], i = value index,
  j = key index).  Differentiable, so ``torch.autograd`` gives an independent backward.

Pinning: tests/golden/make_golden.py runs the reference file RWKV-v7_simple.py itself
(in this container) and stores its inputs/outputs/autograd grads; tests check
``wkv7_naive`` and the literal restatements against those vectors.

Argument naming follows the op schema (wkv7_op.cpp:22-23): (w, q, k, v, z, a) where
z = the kernel's ``a`` (= -kk) and a = the kernel's ``b`` (= kk * a_gate).
"""
from __future__ import annotations

import torch

CHUNK_LEN = 16


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16 and back (what `to_bf` does, wkv7_cuda.cu:6)."""
    return x.to(torch.bfloat16).to(x.dtype)


def wkv7_naive(w, q, k, v, z, a, state0=None):
    """RWKV-v7_simple.py:20-32.  Inputs (B,T,H,N) in any float dtype; w is w_raw.

    Returns y (B,T,H,N) in the input dtype, un-rounded, and the final state.
    """
    B, T, H, N = w.shape
    decay = torch.exp(-torch.exp(w))
    state = (torch.zeros(B, H, N, N, dtype=w.dtype) if state0 is None else state0)
    ys = []
    for t in range(T):
        kk, rr, vv = k[:, t], q[:, t], v[:, t]
        aa, bb = z[:, t], a[:, t]
        sa = state @ aa.unsqueeze(-1)                              # (B,H,N,1)   :27
        sab = sa @ bb.unsqueeze(-2)                                # (B,H,N,N)   :28
        state = state * decay[:, t, :, None, :] + sab + vv.unsqueeze(-1) * kk.unsqueeze(-2)  # :30
        ys.append((state @ rr.unsqueeze(-1)).squeeze(-1))          # :31
    return torch.stack(ys, dim=1), state


def wkv7_forward_ref(w, q, k, v, z, a, chunk_len=CHUNK_LEN, dtype=torch.float32, round_y=True):
    """Literal restatement of forward_kernel (wkv7_cuda.cu:10-52).

    Inputs: (B,T,H,N) tensors (bf16 or float); converted with `to_float` semantics.
    Returns y (B,T,H,N) [dtype, bf16-rounded when round_y], s (B,H,T/chunk,N,N) holding
    S^T (s[b,h,c,j,i] = S[i,j]) and sa (B,T,H,N), both in `dtype`.
    """
    B, T, H, N = w.shape
    assert T % chunk_len == 0
    w, q, k, v, z, a = [x.to(dtype) for x in (w, q, k, v, z, a)]
    state = torch.zeros(B, H, N, N, dtype=dtype)          # state[b,h,i,j]  (:14, thread i)
    y = torch.empty(B, T, H, N, dtype=dtype)
    sa_out = torch.empty(B, T, H, N, dtype=dtype)
    s = torch.empty(B, H, T // chunk_len, N, N, dtype=dtype)
    decay = torch.exp(-torch.exp(w))                       # :21
    for t in range(T):
        sa = (state * z[:, t, :, None, :]).sum(-1)         # :27-31  sa_i = sum_j a_j S_ij
        sa_out[:, t] = sa
        state = state * decay[:, t, :, None, :] + sa[..., None] * a[:, t, :, None, :] \
            + k[:, t, :, None, :] * v[:, t, :, :, None]    # :38
        y[:, t] = (state * q[:, t, :, None, :]).sum(-1)    # :39-41
        if (t + 1) % chunk_len == 0:
            s[:, :, t // chunk_len] = state.transpose(-1, -2)   # :44-50 (stored transposed)
    if round_y:
        y = bf16_round(y)
    return y, s, sa_out


def wkv7_backward_ref(w, q, k, v, z, a, dy, s, sa, chunk_len=CHUNK_LEN, dtype=torch.float32,
                      round_out=True):
    """Literal restatement of backward_kernel (wkv7_cuda.cu:54-130).

    Thread i of the reference owns column i of S as stateT[j] = S[j,i], row i of dS as
    dstate[j] = dS[i,j] and column i of dS as dstateT[j] = dS[j,i].  Here
    ST[b,h,i,j] = S[j,i]; dS[b,h,i,j] = dS[i,j]; dST[b,h,i,j] = dS[j,i].
    """
    B, T, H, N = w.shape
    assert T % chunk_len == 0
    w, q, k, v, z, a, dy = [x.to(dtype) for x in (w, q, k, v, z, a, dy)]
    s, sa = s.to(dtype), sa.to(dtype)
    ST = torch.zeros(B, H, N, N, dtype=dtype)
    dS = torch.zeros(B, H, N, N, dtype=dtype)
    dST = torch.zeros(B, H, N, N, dtype=dtype)
    dw, dq, dk, dv, dz, da = [torch.empty(B, T, H, N, dtype=dtype) for _ in range(6)]
    for t in range(T - 1, -1, -1):
        qt, kt, vt, zt, at, dyt, sat = q[:, t], k[:, t], v[:, t], z[:, t], a[:, t], dy[:, t], sa[:, t]
        wfac = -torch.exp(w[:, t])                          # :66
        wt = torch.exp(wfac)                                # :67
        if (t + 1) % chunk_len == 0:
            ST = s[:, :, t // chunk_len].clone()            # :76-82  stateT[j] = s[..., i, j]
        dq[:, t] = (ST * dyt[:, :, None, :]).sum(-1)        # :84-89
        iw = 1.0 / wt                                       # :91
        # :93-97  (thread i, loop j): stateT[j] = (stateT[j] - k_i v_j - b_i sa_j) * iw_i
        ST = (ST - kt[..., None] * vt[:, :, None, :] - at[..., None] * sat[:, :, None, :]) * iw[..., None]
        dS = dS + dyt[..., None] * qt[:, :, None, :]        # dstate[j]  += dy_i q_j
        dST = dST + qt[..., None] * dyt[:, :, None, :]      # dstateT[j] += q_i dy_j
        dwt = (dST * ST).sum(-1)                            # :101
        dk[:, t] = (dST * vt[:, :, None, :]).sum(-1)        # :102
        dv[:, t] = (dS * kt[:, :, None, :]).sum(-1)         # :103
        dSb = (dS * at[:, :, None, :]).sum(-1)              # :104
        da[:, t] = (dST * sat[:, :, None, :]).sum(-1)       # :105  (db in the kernel)
        dw[:, t] = dwt * wt * wfac                          # :107
        dz[:, t] = (ST * dSb[:, :, None, :]).sum(-1)        # :117-122 (da in the kernel)
        dS = dS * wt[:, :, None, :] + dSb[..., None] * zt[:, :, None, :]       # :126
        dST = dST * wt[..., None] + zt[..., None] * dSb[:, :, None, :]         # :127
    outs = [dw, dq, dk, dv, dz, da]
    if round_out:
        outs = [bf16_round(x) for x in outs]
    return tuple(outs)


def wkv7_autograd(w, q, k, v, z, a, dy, dtype=torch.float64):
    """Independent backward: torch.autograd through `wkv7_naive` (fp64 by default)."""
    leaves = [x.detach().to(dtype).requires_grad_(True) for x in (w, q, k, v, z, a)]
    y, _ = wkv7_naive(*leaves)
    y.backward(dy.to(dtype))
    return y.detach(), tuple(x.grad for x in leaves)


def make_inputs(B, T, H, N=64, seed=42, dtype=torch.bfloat16):
    """Synthetic op inputs with the structure RWKV_Tmix_x070 feeds the op
    (src/model.py:175-190): w_raw <= -0.5, ||kk||_2 = 1 per head, a_gate in (0,1),
    z = -kk, a = kk * a_gate.  Unstructured random z/a make the recurrence diverge
    (SURVEY.md 8d)."""
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(B, T, H, N, generator=g) * 0.5
    k = torch.randn(B, T, H, N, generator=g) * 0.5
    v = torch.randn(B, T, H, N, generator=g) * 0.5
    w = -torch.nn.functional.softplus(-torch.randn(B, T, H, N, generator=g)) - 0.5
    kk = torch.nn.functional.normalize(torch.randn(B, T, H, N, generator=g), dim=-1, p=2.0)
    gate = torch.sigmoid(torch.randn(B, T, H, N, generator=g))
    z = -kk
    a = kk * gate
    dy = torch.randn(B, T, H, N, generator=g)
    return tuple(x.to(dtype).contiguous() for x in (w, r, k, v, z, a, dy))


def rel_rms(x, ref):
    """||x - ref||_2 / ||ref||_2  (VisualRWKV-v6/v6.xx/test_kernel.py:27-30)."""
    x, ref = x.double(), ref.double()
    return ((x - ref).pow(2).sum() / ref.pow(2).sum().clamp_min(1e-300)).sqrt().item()
