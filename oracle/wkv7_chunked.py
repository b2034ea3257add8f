"""Chunked (matmul-form) restatement of WKV7 forward/backward.  TEST INFRASTRUCTURE ONLY.

This is the algorithm the MFMA kernels implement (DESIGN.md "chunked WKV7"); it is mathematically
identical to the reference recurrence (wkv7_cuda.cu:17-51 / :62-129) but groups L=16 tokens so that
all work is dense products.  Used (a) to validate the derivation against fp64 autograd, (b) to study
the numerics of reduced-precision products (`mm` is pluggable) before committing to a kernel design.

Per chunk, per head (rows = time, S0 in R^{N_i x N_j}):
    c_t = prod_{r<=t} w_r,  c_0 = 1
    Zt = z * c_{t-1}   Qt = q * c_t   Ah = a / c_t   Kh = k / c_t
    M_za = tril_(Zt Ah^T)  M_zk = tril_(Zt Kh^T)  M_qa = tril(Qt Ah^T)  M_qk = tril(Qt Kh^T)
    SA = (I - M_za)^-1 (Zt S0^T + M_zk V)
    Y  = Qt S0^T + M_qa SA + M_qk V
    S_L = (S0 + SA^T Ah + V^T Kh) diag(c_L)
"""
from __future__ import annotations

import torch

L = 16


def mm_exact(a, b):
    return a @ b


def split_bf16(x, terms=2):
    parts, r = [], x
    for _ in range(terms):
        h = r.to(torch.bfloat16).to(x.dtype)
        parts.append(h)
        r = r - h
    return parts


def make_mm_bf16x3(acc_dtype=torch.float32):
    """a@b with both operands split into hi+lo bf16 and the lo*lo term dropped (3 bf16 products,
    fp32 accumulation) -- what three bf16 MFMAs compute."""
    def mm(a, b):
        ah, al = split_bf16(a.to(acc_dtype))
        bh, bl = split_bf16(b.to(acc_dtype))
        return (ah @ bh + (ah @ bl + al @ bh)).to(a.dtype)
    return mm


def make_mm_bf16x1():
    def mm(a, b):
        return (a.to(torch.bfloat16).float() @ b.to(torch.bfloat16).float()).to(a.dtype)
    return mm


def _chunk_quantities(w_raw, q, k, z, a):
    """inputs (..., L, N) -> decayed operands."""
    lw = -torch.exp(w_raw)                    # log w
    cl = torch.cumsum(lw, dim=-2)             # log c_t
    c = torch.exp(cl)
    cprev = torch.exp(cl - lw)                # c_{t-1}
    ic = torch.exp(-cl)
    return lw, c, cprev, ic, z * cprev, q * c, a * ic, k * ic


def _solve_unit_lower(M, R, mm):
    """(I - M)^-1 R for strictly-lower M (L x L): forward substitution, row by row."""
    rows = []
    for t in range(M.shape[-1]):
        acc = R[..., t, :]
        for s in range(t):
            acc = acc + M[..., t, s, None] * rows[s]
        rows.append(acc)
    return torch.stack(rows, dim=-2)


def _solve_unit_lower_T(M, R):
    """(I - M)^-T R: back substitution."""
    n = M.shape[-1]
    rows = [None] * n
    for t in range(n - 1, -1, -1):
        acc = R[..., t, :]
        for s in range(t + 1, n):
            acc = acc + M[..., s, t, None] * rows[s]
        rows[t] = acc
    return torch.stack(rows, dim=-2)


def forward(w, q, k, v, z, a, mm=mm_exact, dtype=torch.float32):
    """(B,T,H,N) inputs (any float dtype) -> y (B,T,H,N), s (B,H,T/16,N,N) [S^T], sa (B,T,H,N), in `dtype`."""
    B, T, H, N = w.shape
    tr = lambda x: x.to(dtype).permute(0, 2, 1, 3).reshape(B, H, T // L, L, N)
    W, Q, K, V, Z, A = map(tr, (w, q, k, v, z, a))
    S = torch.zeros(B, H, N, N, dtype=dtype)
    tril = torch.tril(torch.ones(L, L, dtype=dtype))
    tril_s = torch.tril(torch.ones(L, L, dtype=dtype), -1)
    ys, sas, ss = [], [], []
    for c_ in range(T // L):
        lw, c, cprev, ic, Zt, Qt, Ah, Kh = _chunk_quantities(W[:, :, c_], Q[:, :, c_], K[:, :, c_], Z[:, :, c_], A[:, :, c_])
        Vc = V[:, :, c_]
        M_za = mm(Zt, Ah.transpose(-1, -2)) * tril_s
        M_zk = mm(Zt, Kh.transpose(-1, -2)) * tril_s
        M_qa = mm(Qt, Ah.transpose(-1, -2)) * tril
        M_qk = mm(Qt, Kh.transpose(-1, -2)) * tril
        R = mm(Zt, S.transpose(-1, -2)) + mm(M_zk, Vc)
        SA = _solve_unit_lower(M_za, R, mm)
        Y = mm(Qt, S.transpose(-1, -2)) + mm(M_qa, SA) + mm(M_qk, Vc)
        U = S + mm(SA.transpose(-1, -2), Ah) + mm(Vc.transpose(-1, -2), Kh)
        S = U * c[..., -1:, :]
        ys.append(Y); sas.append(SA); ss.append(S.transpose(-1, -2))
    y = torch.stack(ys, 2).reshape(B, H, T, N).permute(0, 2, 1, 3)
    sa = torch.stack(sas, 2).reshape(B, H, T, N).permute(0, 2, 1, 3)
    s = torch.stack(ss, 2)
    return y.contiguous(), s.contiguous(), sa.contiguous()


def backward(w, q, k, v, z, a, dy, s, sa, mm=mm_exact, dtype=torch.float32):
    """Chunked backward.  Returns dw, dq, dk, dv, dz, da (B,T,H,N) in `dtype` (un-rounded)."""
    B, T, H, N = w.shape
    tr = lambda x: x.to(dtype).permute(0, 2, 1, 3).reshape(B, H, T // L, L, N)
    W, Q, K, V, Z, A, DY, SA_ = map(tr, (w, q, k, v, z, a, dy, sa))
    s = s.to(dtype)
    tril = torch.tril(torch.ones(L, L, dtype=dtype))
    tril_s = torch.tril(torch.ones(L, L, dtype=dtype), -1)
    dS = torch.zeros(B, H, N, N, dtype=dtype)       # dL/dS at the end of the current chunk
    outs = {n: [] for n in ("dw", "dq", "dk", "dv", "dz", "da")}
    T_ = lambda x: x.transpose(-1, -2)
    for c_ in range(T // L - 1, -1, -1):
        S0 = T_(s[:, :, c_ - 1]) if c_ > 0 else torch.zeros(B, H, N, N, dtype=dtype)
        SL = T_(s[:, :, c_])
        lw, c, cprev, ic, Zt, Qt, Ah, Kh = _chunk_quantities(W[:, :, c_], Q[:, :, c_], K[:, :, c_], Z[:, :, c_], A[:, :, c_])
        Vc, SA, dY = V[:, :, c_], SA_[:, :, c_], DY[:, :, c_]
        cL = c[..., -1:, :]
        M_za = mm(Zt, T_(Ah)) * tril_s
        M_zk = mm(Zt, T_(Kh)) * tril_s
        M_qa = mm(Qt, T_(Ah)) * tril
        M_qk = mm(Qt, T_(Kh)) * tril
        dU = dS * cL                                           # (N_i, N_j)
        g_last = (dS * SL).sum(-2)                             # sum_i dS_L * S_L   -> (N_j)
        dSA = mm(Ah, T_(dU)) + mm(T_(M_qa), dY)
        dV = mm(Kh, T_(dU)) + mm(T_(M_qk), dY)
        dAh = mm(SA, dU)
        dKh = mm(Vc, dU)
        dQt = mm(dY, S0)
        dM_qa = mm(dY, T_(SA)) * tril
        dM_qk = mm(dY, T_(Vc)) * tril
        dR = _solve_unit_lower_T(M_za, dSA)
        dM_za = mm(dR, T_(SA)) * tril_s
        dM_zk = mm(dR, T_(Vc)) * tril_s
        dZt = mm(dR, S0) + mm(dM_za, Ah) + mm(dM_zk, Kh)
        dV = dV + mm(T_(M_zk), dR)
        dQt = dQt + mm(dM_qa, Ah) + mm(dM_qk, Kh)
        dAh = dAh + mm(T_(dM_za), Zt) + mm(T_(dM_qa), Qt)
        dKh = dKh + mm(T_(dM_zk), Zt) + mm(T_(dM_qk), Qt)
        dS = dU + mm(T_(dY), Qt) + mm(T_(dR), Zt)              # dL/dS_0
        # elementwise back to the raw inputs
        dz = dZt * cprev
        dq = dQt * c
        da = dAh * ic
        dk = dKh * ic
        g = dQt * Qt - dAh * Ah - dKh * Kh                     # dL/dlog c_t
        g = g + torch.cat([(dZt * Zt)[..., 1:, :], torch.zeros_like(g[..., :1, :])], dim=-2)
        g[..., -1, :] = g[..., -1, :] + g_last
        dlw = torch.flip(torch.cumsum(torch.flip(g, [-2]), dim=-2), [-2])
        dw = dlw * lw                                          # d/dw_raw of log w = -exp(w_raw) = lw
        for n, x in zip(("dw", "dq", "dk", "dv", "dz", "da"), (dw, dq, dk, dV, dz, da)):
            outs[n].append(x)
    res = []
    for n in ("dw", "dq", "dk", "dv", "dz", "da"):
        x = torch.stack(outs[n][::-1], 2).reshape(B, H, T, N).permute(0, 2, 1, 3).contiguous()
        res.append(x)
    return tuple(res)
