"""The LayerNorm kernels (csrc/ln_kernels.h) on the host lockstep emulator: residual add + LayerNorm fused with the token shift and
lerps of the time-mix / channel-mix (ln_mix_*), against a torch statement that rounds where the kernels round.  Covers what needs no
GPU to go wrong: token ranges per workgroup (longer than, equal to and shorter than one row), the row before / after a range,
sample boundaries inside a range, inactive lanes (C / 8 not a multiple of 64), the partial-row column sums."""
import ctypes

import pytest
import torch


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def PA(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _inputs(B, T, C, M, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * T, C, generator=g).bfloat16()
    d = (0.5 * torch.randn(B * T, C, generator=g)).bfloat16()
    w = (1 + 0.2 * torch.randn(C, generator=g)).bfloat16()
    b = (0.1 * torch.randn(C, generator=g)).bfloat16()
    mus = [torch.rand(C, generator=g).bfloat16() for _ in range(M)]
    return x, d, w, b, mus


def _ln_ref(x, d, w, b, eps):
    """xn = bf16(x + d); y = bf16(LN(xn)) with fp32 statistics of the rounded xn (what add_ln_fwd_kernel does)."""
    xn = (x.float() + d.float()).bfloat16() if d is not None else x
    v = xn.float()
    mu = v.mean(-1, keepdim=True)
    var = ((v - mu) ** 2).mean(-1, keepdim=True)
    rs = torch.rsqrt(var + eps)
    y = ((v - mu) * rs * w.float() + b.float()).bfloat16()
    return xn, y, mu.squeeze(-1).contiguous(), rs.squeeze(-1).contiguous()


def _mix_ref(y, T, mus):
    yf = y.float().view(-1, T, y.shape[-1])
    prev = torch.cat([torch.zeros_like(yf[:, :1]), yf[:, :-1]], dim=1)
    xx = prev - yf
    return [(yf + xx * m.float()).bfloat16().view(-1, y.shape[-1]) for m in mus]


@pytest.mark.parametrize("B,T,C,M,grid,has_delta", [(2, 7, 128, 6, 3, True), (3, 5, 64, 1, 15, True), (1, 33, 512, 6, 4, False),
                                                    (4, 4, 192, 1, 5, True), (2, 16, 128, 6, 40, True)])
def test_ln_mix_forward(emu_lib, B, T, C, M, grid, has_delta):
    x, d, w, b, mus = _inputs(B, T, C, M, seed=B * 100 + T + C)
    d = d if has_delta else None
    ntok = B * T
    grid = min(grid, ntok)
    xn = torch.zeros_like(x)
    outs = [torch.zeros_like(x) for _ in range(M)]
    mean, rstd = torch.zeros(ntok), torch.zeros(ntok)
    f = emu_lib.emu_ln_mix_fwd
    f.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int]
    assert f(ntok, T, C, 1e-5, M, P(x), P(d), P(w), P(b), PA(mus), P(xn) if has_delta else None, PA(outs), P(mean), P(rstd), grid) == 0
    xn_r, y_r, mu_r, rs_r = _ln_ref(x, d, w, b, 1e-5)
    if has_delta:
        assert torch.equal(xn, xn_r)
    assert torch.allclose(mean, mu_r, rtol=1e-5, atol=1e-6) and torch.allclose(rstd, rs_r, rtol=1e-5, atol=1e-6)
    for o, r in zip(outs, _mix_ref(y_r, T, mus)):
        # the reference's LayerNorm output may differ from the kernel's by an ulp where the summation order moves a statistic
        diff = (o.float() - r.float()).abs()
        assert float((diff > 0).float().mean()) < 0.02 and float(diff.max()) <= 2 ** -6 * float(r.float().abs().max())


@pytest.mark.parametrize("B,T,C,grid,has_res", [(2, 7, 128, 3, True), (3, 5, 64, 15, True), (1, 33, 512, 4, False), (2, 16, 192, 40, True)])
def test_ln_mix_backward_channel_mix(emu_lib, B, T, C, grid, has_res):
    """ln_mix_bwd_kernel<1> + the column sums against autograd through an fp32 statement of LayerNorm -> shift -> lerp."""
    x, d, w, b, mus = _inputs(B, T, C, 1, seed=B * 10 + T + C)
    ntok = B * T
    grid = min(grid, ntok)
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(ntok, C, generator=g).bfloat16()
    dres = torch.randn(ntok, C, generator=g).bfloat16() if has_res else None
    xn_r, y_r, mu_r, rs_r = _ln_ref(x, d, w, b, 1e-5)
    dx = torch.zeros_like(x)
    dwb, dmu = torch.zeros(2 * C), torch.zeros(C)
    f = emu_lib.emu_ln_mix_bwd1
    f.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 11 + [ctypes.c_int]
    assert f(ntok, T, C, P(xn_r), P(mu_r), P(rs_r), P(w), P(b), P(mus[0]), P(dout), P(dres), P(dx), P(dwb), P(dmu), grid) == 0
    # reference: autograd in fp64 from the rounded xn
    xv = xn_r.double().requires_grad_(True)
    wv, bv, mv = w.double().requires_grad_(True), b.double().requires_grad_(True), mus[0].double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xv, (C,), wv, bv, 1e-5).view(B, T, C)
    prev = torch.cat([torch.zeros_like(y[:, :1]), y[:, :-1]], dim=1)
    out = (y + (prev - y) * mv).view(ntok, C)
    out.backward(dout.double())
    dx_ref = xv.grad + (dres.double() if has_res else 0)
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert rel(dx, dx_ref) < 6e-3              # bf16 output, and the lerp's input gradient is rounded to bf16 on the way (as the two-kernel path does)
    assert rel(dwb[:C], wv.grad) < 6e-3 and rel(dwb[C:], bv.grad) < 6e-3 and rel(dmu, mv.grad) < 6e-3


def test_add_ln_kernels_on_the_emulator(emu_lib):
    """The plain add + LayerNorm pair through the same harness (forward bit-exact against the rounding-faithful statement up to
    statistic ulps, backward against fp64 autograd)."""
    B, T, C, grid = 3, 6, 128, 4
    x, d, w, b, _ = _inputs(B, T, C, 1, seed=77)
    ntok = B * T
    xn, y = torch.zeros_like(x), torch.zeros_like(x)
    mean, rstd = torch.zeros(ntok), torch.zeros(ntok)
    f = emu_lib.emu_add_ln_fwd
    f.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 8 + [ctypes.c_int]
    assert f(ntok, C, 1e-5, P(x), P(d), P(w), P(b), P(xn), P(y), P(mean), P(rstd), grid) == 0
    xn_r, y_r, mu_r, rs_r = _ln_ref(x, d, w, b, 1e-5)
    assert torch.equal(xn, xn_r)
    assert float((y.float() != y_r.float()).float().mean()) < 0.02
    g = torch.Generator().manual_seed(6)
    dy, dres = torch.randn(ntok, C, generator=g).bfloat16(), torch.randn(ntok, C, generator=g).bfloat16()
    dx, dwb = torch.zeros_like(x), torch.zeros(2 * C)
    f = emu_lib.emu_add_ln_bwd
    f.argtypes = [ctypes.c_long, ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int]
    assert f(ntok, C, P(dy), P(dres), P(xn), P(mean), P(rstd), P(w), P(dx), P(dwb), grid) == 0
    xv, wv, bv = xn.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xv, (C,), wv, bv, 1e-5).backward(dy.double())
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert rel(dx, xv.grad + dres.double()) < 4e-3 and rel(dwb[:C], wv.grad) < 1e-4 and rel(dwb[C:], bv.grad) < 1e-4
