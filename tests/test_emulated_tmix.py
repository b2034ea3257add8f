"""csrc/tmix_fused.hip -- kernels and C entry points -- compiled for the host lockstep emulator (tests/emu/emu_tmix.cpp): the
element-wise glue of the time-mix / channel-mix through the same vrwkv_* symbols as the product library, on CPU tensors.
Checks what this round added without a GPU: the lerps' backward with the LayerNorm output recomputed (vrwkv_mix_bwd_ln_bf16)
against the stored-input form, the v_first gradient chain of kva (vrwkv_kva_bwd3_bf16), and the kernels they are built on against
fp64 autograd."""
import ctypes

import pytest
import torch
import torch.nn.functional as F


L, I, F32 = ctypes.c_long, ctypes.c_int, ctypes.c_float
VP = ctypes.c_void_p


def P(t):
    return VP(t.data_ptr()) if t is not None else VP(0)


def PA(ts):
    return (VP * len(ts))(*[t.data_ptr() for t in ts])


def bf(*shape, g, scale=1.0):
    return (scale * torch.randn(*shape, generator=g)).bfloat16()


def rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))


def call(lib, name, argtypes, *args):
    f = getattr(lib, name)
    f.argtypes, f.restype = argtypes, I
    rc = f(*args)
    assert rc == 0, (name, rc)


def ws_floats(lib, ntok, C, nvec):
    lib.vrwkv_param_grad_ws_floats.argtypes, lib.vrwkv_param_grad_ws_floats.restype = [L, I, I], L
    return int(lib.vrwkv_param_grad_ws_floats(ntok, C, nvec))


@pytest.mark.parametrize("B,T,C,M,dup3", [(2, 5, 64, 6, True), (1, 40, 128, 6, False), (3, 1, 64, 1, False), (2, 21, 192, 1, False), (4, 9, 64, 2, False)])
def test_mix_forward_backward_against_autograd(emu_lib, B, T, C, M, dup3):
    g = torch.Generator().manual_seed(B * 31 + T + C + M)
    ntok = B * T
    x = bf(ntok, C, g=g)
    mus = [torch.rand(C, generator=g).bfloat16() for _ in range(M)]
    outs = [torch.zeros_like(x) for _ in range(M)]
    call(emu_lib, "vrwkv_mix_fwd_bf16", [L, I, I, I, VP, VP, VP, VP], ntok, T, C, M, P(x), PA(mus), PA(outs), None)
    xv = x.double().view(B, T, C).requires_grad_(True)
    mv = [m.double().requires_grad_(True) for m in mus]
    prev = torch.cat([torch.zeros_like(xv[:, :1]), xv[:, :-1]], dim=1)
    ref = [(xv + (prev - xv) * m).view(ntok, C) for m in mv]
    for o, r in zip(outs, ref):
        assert torch.equal(o, r.detach().float().bfloat16())          # one fma per element, one rounding
    douts = [bf(ntok, C, g=g) for _ in range(M)]
    d3b = bf(ntok, C, g=g) if dup3 else None
    dx, dmu = torch.zeros_like(x), torch.zeros(M, C)
    ws = torch.zeros(ws_floats(emu_lib, ntok, C, M))
    call(emu_lib, "vrwkv_mix_bwd2_bf16", [L, I, I, I] + [VP] * 8, ntok, T, C, M, P(x), PA(mus), PA(douts), P(d3b), P(dx), P(dmu), P(ws), None)
    gr = [d.double() for d in douts]
    if dup3:
        gr[3] = gr[3] + d3b.double()
    torch.autograd.backward(ref, gr)
    assert rel(dx, xv.grad.view(ntok, C)) < 3e-3
    for j in range(M):
        assert rel(dmu[j], mv[j].grad) < 1e-5


@pytest.mark.parametrize("B,T,C,dup3", [(2, 7, 64, True), (1, 33, 128, False), (3, 16, 64, True)])
def test_mix_backward_with_recomputed_layernorm_output(emu_lib, B, T, C, dup3):
    """vrwkv_mix_bwd_ln_bf16(xn, mean, rstd, gamma, beta) == vrwkv_mix_bwd2_bf16(y) with y = bf16(LN(xn)) formed the same way."""
    g = torch.Generator().manual_seed(T * 7 + C)
    ntok, M = B * T, 6
    xn = bf(ntok, C, g=g)
    w, b = (1 + 0.2 * torch.randn(C, generator=g)).bfloat16(), (0.1 * torch.randn(C, generator=g)).bfloat16()
    v = xn.float()
    mean = v.mean(-1).contiguous()
    rstd = torch.rsqrt(((v - mean[:, None]) ** 2).mean(-1) + 1e-5).contiguous()
    y = torch.addcmul(b.float(), (v - mean[:, None]) * rstd[:, None], w.float())          # fma((x - mu) rs, w, b) up to the fma's single rounding
    yb = y.bfloat16()
    mus = [torch.rand(C, generator=g).bfloat16() for _ in range(M)]
    douts = [bf(ntok, C, g=g) for _ in range(M)]
    d3b = bf(ntok, C, g=g) if dup3 else None
    res = []
    for name, extra in (("vrwkv_mix_bwd_ln_bf16", True), ("vrwkv_mix_bwd2_bf16", False)):
        dx, dmu = torch.zeros_like(xn), torch.zeros(M, C)
        ws = torch.zeros(ws_floats(emu_lib, ntok, C, M))
        if extra:
            call(emu_lib, name, [L, I, I, I] + [VP] * 12, ntok, T, C, M, P(xn), P(mean), P(rstd), P(w), P(b), PA(mus),
                 PA(douts), P(d3b), P(dx), P(dmu), P(ws), None)
        else:
            call(emu_lib, name, [L, I, I, I] + [VP] * 8, ntok, T, C, M, P(yb), PA(mus), PA(douts), P(d3b), P(dx), P(dmu), P(ws), None)
        res.append((dx, dmu))
    assert torch.equal(res[0][0], res[1][0])                    # dx does not depend on y at all
    assert rel(res[0][1], res[1][1]) < 2e-3                     # dmu: y may differ by an ulp where fma and mul+add round differently


def _kva_inputs(B, T, C, g):
    ntok = B * T
    k, v, vf, vl, al = [bf(ntok, C, g=g) for _ in range(5)]
    kk, ka, a0, v0 = [bf(C, g=g, scale=0.5) for _ in range(4)]
    return ntok, k, v, vf, vl, al, kk, ka, a0, v0


def test_kva_backward_adds_the_v_first_gradient_of_later_layers(emu_lib):
    """vrwkv_kva_bwd3_bf16 with dvfirst_in == vrwkv_kva_bwd2_bf16 + that tensor (fp32 add, one rounding); everything else equal."""
    g = torch.Generator().manual_seed(11)
    B, T, C = 2, 9, 128
    ntok, k, v, vf, vl, al, kk, ka, a0, v0 = _kva_inputs(B, T, C, g)
    dk2, dv2, dz, db, dk2b, dv2b, dvf_in = [bf(ntok, C, g=g) for _ in range(7)]
    outs = {}
    for chain in (False, True):
        dk, dv, dvf, dvl, dal = [torch.zeros_like(k) for _ in range(5)]
        pg = torch.zeros(4, C)
        ws = torch.zeros(ws_floats(emu_lib, ntok, C, 4))
        call(emu_lib, "vrwkv_kva_bwd3_bf16", [L, I, I] + [VP] * 24, ntok, C, 1, P(k), P(v), P(vf), P(vl), P(al), P(kk), P(ka), P(a0), P(v0),
             P(dk2), P(dv2), P(dz), P(db), P(dk2b), P(dv2b), P(dvf_in if chain else None), P(dk), P(dv), P(dvf), P(dvl), P(dal), P(pg), P(ws), None)
        outs[chain] = (dk, dv, dvf, dvl, dal, pg)
    for i in (0, 1, 3, 4, 5):
        assert torch.equal(outs[True][i], outs[False][i])
    # the chained dvfirst is (term + in) rounded once; the unchained one is the term rounded
    sv = torch.sigmoid(v0.float() + vl.float())
    term = (dv2.float() + dv2b.float()) * sv
    assert rel(outs[False][2], term) < 3e-3
    assert rel(outs[True][2], term + dvf_in.float()) < 3e-3


def test_kva_forward_backward_against_autograd(emu_lib):
    g = torch.Generator().manual_seed(12)
    B, T, C, H = 2, 6, 128, 2
    ntok, k, v, vf, vl, al, kk, ka, a0, v0 = _kva_inputs(B, T, C, g)
    k2, v2, z, b = [torch.zeros_like(k) for _ in range(4)]
    call(emu_lib, "vrwkv_kva_fwd_bf16", [L, I, I] + [VP] * 14, ntok, C, 1, P(k), P(v), P(vf), P(vl), P(al), P(kk), P(ka), P(a0), P(v0),
         P(k2), P(v2), P(z), P(b), None)
    xs = [t.double().requires_grad_(True) for t in (k, v, vf, vl, al, kk, ka, a0, v0)]
    kd, vd, vfd, vld, ald, kkd, kad, a0d, v0d = xs
    a = torch.sigmoid(a0d + ald)
    v2r = vd + (vfd - vd) * torch.sigmoid(v0d + vld)
    kkn = F.normalize((kd * kkd).view(ntok, H, -1), dim=-1, p=2.0).view(ntok, C)
    k2r = kd * (1 + (a - 1) * kad)
    refs = (k2r, v2r, -kkn, kkn * a)
    for o, r in zip((k2, v2, z, b), refs):
        assert rel(o, r.detach()) < 3e-3
    grads = [bf(ntok, C, g=g) for _ in range(4)]
    torch.autograd.backward(refs, [t.double() for t in grads])
    dk, dv, dvf, dvl, dal = [torch.zeros_like(k) for _ in range(5)]
    pg = torch.zeros(4, C)
    ws = torch.zeros(ws_floats(emu_lib, ntok, C, 4))
    call(emu_lib, "vrwkv_kva_bwd3_bf16", [L, I, I] + [VP] * 24, ntok, C, 1, P(k), P(v), P(vf), P(vl), P(al), P(kk), P(ka), P(a0), P(v0),
         P(grads[0]), P(grads[1]), P(grads[2]), P(grads[3]), None, None, None, P(dk), P(dv), P(dvf), P(dvl), P(dal), P(pg), P(ws), None)
    for mine, x in zip((dk, dv, dvf, dvl, dal), xs[:5]):
        assert rel(mine, x.grad) < 4e-3
    for j, x in enumerate(xs[5:]):
        assert rel(pg[j], x.grad) < 1e-4


def test_relusq_and_decay(emu_lib):
    g = torch.Generator().manual_seed(13)
    n, C = 40, 64
    h = bf(n, C, g=g)
    y = torch.zeros_like(h)
    call(emu_lib, "vrwkv_relusq_fwd_bf16", [L, VP, VP, VP], n * C, P(h), P(y), None)
    assert torch.equal(y, (torch.relu(h.float()) ** 2).bfloat16())
    dy, dh = bf(n, C, g=g), torch.zeros_like(h)
    call(emu_lib, "vrwkv_relusq_bwd_bf16", [L, VP, VP, VP, VP], n * C, P(h), P(dy), P(dh), None)
    assert torch.equal(dh, (2 * torch.relu(h.float()) * dy.float()).bfloat16())
    w0, w = bf(C, g=g), torch.zeros_like(h)
    call(emu_lib, "vrwkv_decay_fwd_bf16", [L, I, VP, VP, VP, VP], n, C, P(h), P(w0), P(w), None)
    assert rel(w, -F.softplus(-(w0.double() + h.double())) - 0.5) < 3e-3
