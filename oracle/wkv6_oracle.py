"""CPU restatement of the reference's WKV6 operator (BASELINE config 4).  TEST INFRASTRUCTURE ONLY.

Follows VisualRWKV-v6/v6.0/cuda/wkv6_cuda.cu:7-61 (kernel_forward), :64-159 (kernel_backward_111: gr, gu, gk, gv),
:162-227 (kernel_backward_222: gw) and the wrapper VisualRWKV-v6/v6.0/src/model.py:43-88 (WKV_6: the kernels get
ew = -exp(w.float()); gu is summed over the batch in Python; gw is the gradient with respect to the RAW w).

Per head (N = 64), state S[i][j] (i = value index = the reference's thread, j = key index), decay d_t = exp(ew_t):
    y_t[i] = sum_j r_t[j] (u[j] k_t[j] v_t[i] + S[i][j])          S[i][j] <- S[i][j] d_t[j] + k_t[j] v_t[i]

Pinning: tests/golden/make_golden_wkv6.py executes the reference's own pure-PyTorch statement of the recurrence
(`naive_recurrent_rwkv6_fla`, VisualRWKV-v6/v6.xx/test_kernel.py:175-215, cross-checked against the demo app's CPU loop
VisualRWKV-v7/v7.00/app/modeling_rwkv.py:891-897) in fp64 and stores inputs, outputs, final state and autograd gradients;
tests/test_v6_cpu.py::test_oracle_reproduces_the_references_own_recurrence holds `wkv6_naive` to those numbers at 1e-12.

`wkv6_naive` is that recurrence vectorised over (B,H) in any float dtype; gradients come from autograd through it
(the reference's three backward sweeps are the hand-derived form of exactly this; `wkv6_backward_ref` restates them
literally for small cases).  `wkv6_chunked` is the 16-token chunk formulation the HIP kernels use, with a pluggable
matmul so that the bf16x3 split can be modelled.
"""
from __future__ import annotations

import torch

CHUNK = 16


def wkv6_naive(r, k, v, w_raw, u, state0=None):
    """r,k,v,w_raw: (B,T,H,N); u: (H,N).  Returns y (B,T,H,N) un-rounded and the final state (B,H,N,N)."""
    B, T, H, N = r.shape
    decay = torch.exp(-torch.exp(w_raw))
    S = torch.zeros(B, H, N, N, dtype=r.dtype) if state0 is None else state0
    ys = []
    for t in range(T):
        kv = v[:, t].unsqueeze(-1) * k[:, t].unsqueeze(-2)                       # [i][j] = v_i k_j      (:43-46)
        ys.append(((u.unsqueeze(-2) * kv + S) * r[:, t].unsqueeze(-2)).sum(-1))   # y_i                   (:48-51)
        S = S * decay[:, t].unsqueeze(-2) + kv                                    # (:53-56)
    return torch.stack(ys, dim=1), S


def wkv6_autograd(r, k, v, w_raw, u, gy, dtype=torch.float64):
    ins = [x.detach().to(dtype).requires_grad_(True) for x in (r, k, v, w_raw, u)]
    y, _ = wkv6_naive(*ins)
    y.backward(gy.to(dtype))
    return y.detach(), [x.grad if x.grad is not None else torch.zeros_like(x) for x in ins]


def wkv6_backward_ref(r, k, v, ew, u, gy):
    """Literal restatement of kernel_backward_111 / _222 for ONE (b,h): tensors (T,N), u (N); fp64 in, fp64 out.
    Returns gr, gk, gv, gw, gu (gu is this (b,h)'s row of the reference's (B,C) buffer)."""
    T, N = r.shape
    w = torch.exp(ew)
    gr, gk, gv, gw = (torch.zeros_like(r) for _ in range(4))
    state = torch.zeros(N, N, dtype=r.dtype)          # [i][j]: thread i = key index here (cuda:66-110), j = value index
    gu = torch.zeros(N, dtype=r.dtype)
    for t in range(T):
        x = k[t].unsqueeze(1) * v[t].unsqueeze(0)
        gr[t] = ((u.unsqueeze(1) * x + state) * gy[t].unsqueeze(0)).sum(1)
        gu += r[t] * (x * gy[t].unsqueeze(0)).sum(1)
        state = state * w[t].unsqueeze(1) + x
    sc = torch.zeros(N, N, dtype=r.dtype)
    for t in range(T - 1, -1, -1):
        x = r[t].unsqueeze(1) * gy[t].unsqueeze(0)
        gk[t] = ((u.unsqueeze(1) * x + sc) * v[t].unsqueeze(0)).sum(1)
        sc = x + sc * w[t].unsqueeze(1)
    sd = torch.zeros(N, N, dtype=r.dtype)             # thread i = value index, j = key index (:139-158)
    for t in range(T - 1, -1, -1):
        x = gy[t].unsqueeze(1) * r[t].unsqueeze(0)
        gv[t] = ((u.unsqueeze(0) * x + sd) * k[t].unsqueeze(0)).sum(1)
        sd = x + sd * w[t].unsqueeze(0)
    # kernel_backward_222
    if T >= 3:
        sa = torch.zeros(N, N, dtype=r.dtype)
        sb = torch.zeros(max(T - 2, 0), N, dtype=r.dtype)
        for t in range(T - 1, 1, -1):
            x = r[t].unsqueeze(1) * gy[t].unsqueeze(0)
            sa = (sa + x) * w[t - 1].unsqueeze(1)
            sb[t - 2] = (sa * v[t - 2].unsqueeze(0)).sum(1) * k[t - 2]
        sss = sb[0].clone()
        gw[1] = sss * ew[1]
        sc2 = torch.zeros(N, N, dtype=r.dtype)
        for t in range(2, T - 1):
            x = k[t - 2].unsqueeze(1) * v[t - 2].unsqueeze(0)
            sc2 = (sc2 + x) * w[t - 1].unsqueeze(1)
            sss = sss + sb[t - 1] - (sc2 * gy[t].unsqueeze(0)).sum(1) * r[t]
            gw[t] = sss * ew[t]
    return gr, gk, gv, gw, gu


def wkv6_chunked(r, k, v, ew, u, gy=None, mm=torch.matmul, clamp=80.0):
    """Chunk formulation (chunk length 16), one pass forward and -- if gy is given -- one reverse pass.
    r,k,v,ew,(gy): (B,T,H,N) float; ew = log decay (= -exp(w_raw)); u (H,N).
    With x_t = sum_{s<=t} ew_s inside a chunk and m = x at the chunk's midpoint (keeps both exponentials in range):
        Rt = r exp(x_{t-1} - m)   Kh = k exp(m - x_t)   Kb = k exp(x_L - x_t)   c_L = exp(x_L)
        A  = tril_strict(Rt Kh^T) + diag(sum_j r u k)
        Y  = Rt (S0 diag(e^m))^T ... written as  (Rt e^m) S0^T + A V ;   S_L = S0 diag(c_L) + V^T Kb
    Returns y, final state and (if gy) the gradients gr, gk, gv, g_ew (w.r.t. the log decay), gu (B,H,N)."""
    B, T, H, N = r.shape
    L = CHUNK
    nch = (T + L - 1) // L
    pad = nch * L - T

    def padt(x):
        return torch.cat([x, torch.zeros(B, pad, H, N, dtype=x.dtype)], dim=1) if pad else x

    r_, k_, v_, ew_ = (padt(x).view(B, nch, L, H, N).permute(0, 3, 1, 2, 4) for x in (r, k, v, ew))   # (B,H,nch,L,N)
    x = torch.cumsum(ew_, dim=3)                                     # inclusive
    xprev = x - ew_
    xL = x[:, :, :, -1:, :]
    m = x[:, :, :, L // 2 - 1: L // 2, :]
    Rt = r_ * torch.exp(torch.clamp(xprev - m, max=clamp))
    Kh = k_ * torch.exp(torch.clamp(m - x, max=clamp))
    Kb = k_ * torch.exp(xL - x)
    em = torch.exp(m)                                                # (…,1,N) <= 1
    cL = torch.exp(xL)
    uu = u.view(1, H, 1, 1, N)
    d = (r_ * uu * k_).sum(-1)                                       # (B,H,nch,L)
    tri = torch.tril(torch.ones(L, L, dtype=r.dtype), -1)
    A = mm(Rt, Kh.transpose(-1, -2)) * tri + torch.diag_embed(d)
    S = torch.zeros(B, H, N, N, dtype=r.dtype)
    ys, S0s = [], []
    for c in range(nch):
        S0s.append(S)
        y = mm(Rt[:, :, c] * em[:, :, c], S.transpose(-1, -2)) + mm(A[:, :, c], v_[:, :, c])
        ys.append(y)
        S = S * cL[:, :, c] + mm(v_[:, :, c].transpose(-1, -2), Kb[:, :, c])
    y = torch.stack(ys, dim=2).permute(0, 2, 3, 1, 4).reshape(B, nch * L, H, N)[:, :T]
    if gy is None:
        return y, S
    gy_ = padt(gy).view(B, nch, L, H, N).permute(0, 3, 1, 2, 4)
    dS = torch.zeros(B, H, N, N, dtype=r.dtype)                      # dL/dS_L of the current chunk, [i][j]
    outs = {n: [] for n in ("gr", "gk", "gv", "gx")}
    gu = torch.zeros(B, H, N, dtype=r.dtype)
    for c in range(nch - 1, -1, -1):
        S0, dY, V = S0s[c], gy_[:, :, c], v_[:, :, c]
        Rtc, Khc, Kbc, Ac = Rt[:, :, c], Kh[:, :, c], Kb[:, :, c], A[:, :, c]
        dV = mm(Ac.transpose(-1, -2), dY) + mm(Kbc, dS.transpose(-1, -2))
        dA = mm(dY, V.transpose(-1, -2))
        dd = torch.diagonal(dA, dim1=-2, dim2=-1)                    # (B,H,L)
        dAl = dA * tri
        dRte = mm(dY, S0)                                            # gradient w.r.t. (Rt e^m)
        dRt = dRte * em[:, :, c] + mm(dAl, Khc)
        dKh = mm(dAl.transpose(-1, -2), Rtc)
        dKb = mm(V, dS)
        glast = (dS * S0).sum(-2, keepdim=True) * cL[:, :, c]       # d/dx_L through S0 diag(c_L)
        dS = dS * cL[:, :, c] + mm(dY.transpose(-1, -2), Rtc * em[:, :, c])
        rr, kk = r_[:, :, c], k_[:, :, c]
        e_r = torch.exp(torch.clamp(xprev[:, :, c] - m[:, :, c], max=clamp))
        e_h = torch.exp(torch.clamp(m[:, :, c] - x[:, :, c], max=clamp))
        e_b = torch.exp(xL[:, :, c] - x[:, :, c])
        ddu = dd.unsqueeze(-1) * uu[:, :, 0]
        outs["gr"].append(dRt * e_r + ddu * kk)
        outs["gk"].append(dKh * e_h + dKb * e_b + ddu * rr)
        outs["gv"].append(dV)
        gu += (dd.unsqueeze(-1) * rr * kk).sum(-2)
        # dL/dx_t: Rt_t depends on x_{t-1} (shifted), Kh_t on -x_t, Kb_t on x_L - x_t, the e^m factor on x_mid
        pr, ph, pb = dRt * Rtc, dKh * Khc, dKb * Kbc
        gx = -ph - pb
        gx[:, :, :-1] += pr[:, :, 1:]                                # Rt_{t+1} uses x_t
        gx[:, :, -1:] += pb.sum(-2, keepdim=True) + glast
        mid = (dRte * Rtc * em[:, :, c]).sum(-2, keepdim=True) - pr.sum(-2, keepdim=True) + ph.sum(-2, keepdim=True)
        gx[:, :, L // 2 - 1: L // 2] += mid                          # everything that carries +m / -m
        # x_t = sum_{s<=t} ew_s  =>  dL/dew_s = sum_{t>=s} dL/dx_t
        outs["gx"].append(torch.flip(torch.cumsum(torch.flip(gx, dims=[-2]), dim=-2), dims=[-2]))
    res = {}
    for n, lst in outs.items():
        res[n] = torch.stack(lst[::-1], dim=2).permute(0, 2, 3, 1, 4).reshape(B, nch * L, H, N)[:, :T]
    return y, S, res["gr"], res["gk"], res["gv"], res["gx"], gu


def make_inputs6(B, T, H, N=64, seed=42, dtype=torch.bfloat16, w_lo=-8.0, w_hi=1.0):
    """The distributions of the reference's own kernel test (VisualRWKV-v6/v6.xx/test_kernel.py:45-50):
    r,k,v,u ~ U(-1,1), w ~ U(-8,1), seed 42."""
    g = torch.Generator().manual_seed(seed)
    uni = lambda *s, lo=-1.0, hi=1.0: torch.rand(*s, generator=g) * (hi - lo) + lo
    r, k, v = uni(B, T, H, N), uni(B, T, H, N), uni(B, T, H, N)
    w = uni(B, T, H, N, lo=w_lo, hi=w_hi)
    u = uni(H, N)
    gy = uni(B, T, H, N)
    return tuple(x.to(dtype).contiguous() for x in (r, k, v, w, u, gy))
