"""Fused element-wise HIP kernels (csrc/tmix_fused.hip) against an fp32 torch statement of the reference
formulas (src/model.py:166-194,222-225) evaluated on the same bf16 inputs; outputs/gradients compared after
one rounding to bf16 (rel-RMS <= 1e-3; 2e-3 for the atomically accumulated parameter gradients).
Token-shift indexing is checked bit-exactly."""
import pytest
import torch
import torch.nn.functional as F

from oracle.wkv7_oracle import rel_rms

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().cuda()


def _ref_grads(fn, inputs, gouts):
    """fn on fp32 copies -> outputs, grads (fp32)."""
    xs = [x.detach().float().requires_grad_(True) for x in inputs]
    outs = fn(*xs)
    outs = outs if isinstance(outs, tuple) else (outs,)
    torch.autograd.backward(outs, [g.float() for g in gouts])
    return outs, [x.grad for x in xs]


def _check(mine, ref, tol=TOL, names=None):
    for i, (a, b) in enumerate(zip(mine, ref)):
        e = rel_rms(a.float().cpu(), b.detach().bfloat16().float().cpu())
        assert e < tol, (names[i] if names else i, e)


@pytest.mark.parametrize("B,T,C,M", [(2, 5, 128, 6), (3, 33, 768, 6), (1, 40, 2048, 1), (3, 1, 64, 6), (7, 592, 256, 6),
                                     (2, 2624, 2048, 6), (1, 17, 4608, 1)])
def test_mix(B, T, C, M):
    from visualrwkv_amd import fused
    x = _rnd(B, T, C, seed=1)
    mus = [torch.rand(1, 1, C, generator=torch.Generator().manual_seed(10 + i)).bfloat16().cuda() for i in range(M)]
    gouts = [_rnd(B, T, C, seed=20 + i) for i in range(M)]

    def ref(x, *mus):
        xx = F.pad(x, (0, 0, 1, -1)) - x
        return tuple(x + xx * m for m in mus)

    xs = [x.clone().requires_grad_(True)] + [m.clone().requires_grad_(True) for m in mus]
    outs = fused.mix(*xs)
    torch.autograd.backward(outs, gouts)
    r_outs, r_grads = _ref_grads(ref, [x] + mus, gouts)
    _check(outs, r_outs)
    _check([xs[0].grad], [r_grads[0]])
    _check([t.grad for t in xs[1:]], r_grads[1:], tol=2e-3)
    # exact shift indexing: integer-valued x (all fp32 arithmetic exact) and mu = 1 give exactly the previous
    # token, and exactly 0 at t = 0 of every sample
    xi = torch.randint(-8, 9, (B, T, C), generator=torch.Generator().manual_seed(3)).bfloat16().cuda()
    ones = [torch.ones(1, 1, C, dtype=torch.bfloat16, device="cuda") for _ in range(M)]
    for o in fused.mix(xi, *ones):
        assert torch.equal(o[:, 1:], xi[:, :-1]) and torch.equal(o[:, 0], torch.zeros_like(o[:, 0]))


def test_decay():
    from visualrwkv_amd import fused
    h = _rnd(2, 17, 256, scale=3.0, seed=3)
    w0 = _rnd(1, 1, 256, scale=2.0, seed=4)
    g = _rnd(2, 17, 256, seed=5)
    ref = lambda h, w0: -F.softplus(-(w0 + h)) - 0.5
    hh, ww = h.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    w = fused.decay(hh, ww)
    w.backward(g)
    (rw,), (rdh, rdw0) = _ref_grads(ref, [h, w0], [g])
    _check([w, hh.grad], [rw, rdh])
    _check([ww.grad], [rdw0], tol=2e-3)
    assert float(w.float().max()) <= -0.5


@pytest.mark.parametrize("has_vres", [False, True])
def test_kva(has_vres):
    from visualrwkv_amd import fused
    B, T, C, H = 2, 9, 256, 4
    k, v, vf, vl, al = [_rnd(B, T, C, seed=30 + i) for i in range(5)]
    k_k, k_a, a0, v0 = [_rnd(1, 1, C, scale=0.5, seed=40 + i) for i in range(4)]
    gouts = [_rnd(B, T, C, seed=50 + i) for i in range(4)]

    def ref(k, v, vf, vl, al, k_k, k_a, a0, v0):
        a = torch.sigmoid(a0 + al)
        v2 = v + (vf - v) * torch.sigmoid(v0 + vl)
        kk = F.normalize((k * k_k).view(B, T, H, -1), dim=-1, p=2.0).view(B, T, C)
        k2 = k * (1 + (a - 1) * k_a)
        return (k2, v2, -kk, kk * a) if has_vres else (k2, -kk, kk * a)

    ins = [k, v, vf, vl, al, k_k, k_a, a0, v0]
    xs = [t.clone().requires_grad_(True) for t in ins]
    if has_vres:
        outs = fused.kva(*xs)
        go = gouts
    else:
        outs = fused.kva(xs[0], None, None, None, xs[4], xs[5], xs[6], xs[7], None)
        go = [gouts[0], gouts[2], gouts[3]]
    torch.autograd.backward(outs, go)
    r_outs, r_grads = _ref_grads(ref, ins, go)
    _check(outs, r_outs)
    idx = [0, 1, 2, 3, 4] if has_vres else [0, 4]
    _check([xs[i].grad for i in idx], [r_grads[i] for i in idx], names=idx)
    pidx = [5, 6, 7, 8] if has_vres else [5, 6, 7]
    _check([xs[i].grad for i in pidx], [r_grads[i] for i in pidx], tol=2e-3, names=pidx)


def test_post():
    from visualrwkv_amd import fused
    B, T, C, H = 2, 11, 256, 4
    y, r, k, v, g = [_rnd(B, T, C, seed=60 + i) for i in range(5)]
    ln_w, ln_b = _rnd(C, seed=70) * 0.5 + 1, _rnd(C, scale=0.1, seed=71)
    r_k = _rnd(H, 64, scale=0.3, seed=72)
    go = _rnd(B, T, C, seed=73)
    eps = 64e-5

    def ref(y, r, k, v, g, ln_w, ln_b, r_k):
        x = F.group_norm(y.view(B * T, C), H, ln_w, ln_b, eps).view(B, T, C)
        x = x + ((r.view(B, T, H, -1) * k.view(B, T, H, -1) * r_k).sum(dim=-1, keepdim=True) * v.view(B, T, H, -1)).view(B, T, C)
        return x * g

    ins = [y, r, k, v, g, ln_w, ln_b, r_k]
    xs = [t.clone().requires_grad_(True) for t in ins]
    out = fused.post(*xs, eps)
    out.backward(go)
    (r_out,), r_grads = _ref_grads(ref, ins, [go])
    _check([out], [r_out])
    _check([t.grad for t in xs[:5]], r_grads[:5], names=list("yrkvg"))
    _check([t.grad for t in xs[5:]], r_grads[5:], tol=2e-3, names=["ln_w", "ln_b", "r_k"])


def test_relu_sq():
    from visualrwkv_amd import fused
    h = _rnd(3, 7, 512, seed=80)
    g = _rnd(3, 7, 512, seed=81)
    hh = h.clone().requires_grad_(True)
    y = fused.relu_sq(hh)
    y.backward(g)
    (ry,), (rdh,) = _ref_grads(lambda h: torch.relu(h) ** 2, [h], [g])
    _check([y, hh.grad], [ry, rdh])


@pytest.mark.parametrize("ntok,C,has_delta", [(7, 64, True), (33, 512, True), (100, 768, False), (64, 2048, True),
                                              (5, 8192, True), (2624, 2048, True), (3000, 1024, False)])
def test_add_ln(ntok, C, has_delta):
    """Residual add + LayerNorm (csrc/ln_fused.hip) vs the reference sequence: bf16 add, then nn.LayerNorm."""
    from visualrwkv_amd import fused
    x = _rnd(ntok, C, seed=1, scale=2.0) + 0.5
    delta = _rnd(ntok, C, seed=2) if has_delta else None
    ln = torch.nn.LayerNorm(C).cuda().bfloat16()
    with torch.no_grad():
        ln.weight.copy_(_rnd(C, seed=3) * 0.5 + 1)
        ln.bias.copy_(_rnd(C, seed=4) * 0.1)
    g_xn, g_y = _rnd(ntok, C, seed=5), _rnd(ntok, C, seed=6)

    def ref(x, w, b, *d):
        xn = (x + d[0]).bfloat16().float() if d else x
        return xn, F.layer_norm(xn, (C,), w, b, ln.eps)

    ins = [x, ln.weight.detach(), ln.bias.detach()] + ([delta] if has_delta else [])
    xs = [x.clone().requires_grad_(True)] + ([delta.clone().requires_grad_(True)] if has_delta else [])
    ln.zero_grad()
    xn, y = fused.add_ln(xs[0], xs[1] if has_delta else None, ln)
    if has_delta:
        torch.autograd.backward([xn, y], [g_xn, g_y])
    else:
        torch.autograd.backward([y], [g_y])

    # fp32 reference with a straight-through rounding of the residual sum
    rx = [t.detach().float().requires_grad_(True) for t in ins]
    if has_delta:
        s = rx[0] + rx[3]
        r_xn = s + (s.bfloat16().float() - s).detach()
    else:
        r_xn = rx[0]
    r_y = F.layer_norm(r_xn, (C,), rx[1], rx[2], ln.eps)
    if has_delta:
        torch.autograd.backward([r_xn, r_y], [g_xn.float(), g_y.float()])
    else:
        torch.autograd.backward([r_y], [g_y.float()])
    if has_delta:
        assert torch.equal(xn, (x.float() + delta.float()).bfloat16())         # the add is exactly the bf16 add
        _check([xs[1].grad], [rx[3].grad])
    _check([y], [r_y])
    _check([xs[0].grad], [rx[0].grad])
    _check([ln.weight.grad, ln.bias.grad], [rx[1].grad, rx[2].grad], tol=2e-3)


@pytest.mark.parametrize("B,T,V", [(2, 9, 512), (3, 33, 65536), (1, 16, 1000)])
def test_fused_cross_entropy_and_l2wrap(B, T, V):
    """csrc/loss_fused.hip vs the eager statement of training_step + L2Wrap (src/model.py:418-434,257-271)."""
    from visualrwkv_amd import fused
    from visualrwkv_amd.visual import VisualRWKV
    g = torch.Generator().manual_seed(B * 100 + T)
    logits = (torch.randn(B, T, V, generator=g) * 2).bfloat16().cuda()
    targets = torch.randint(0, V, (B, T), generator=g).cuda()
    targets[:, : T // 3] = -100
    targets[0, -2] = -100
    if B > 2:
        targets[2] = -100                                             # a sample without any valid label
    a = logits.clone().requires_grad_(True)
    loss = fused.loss_from_logits(a, targets, -100)
    (loss * 1.5).backward()
    b = logits.clone().float().requires_grad_(True)                   # eager reference in fp32
    ref = VisualRWKV.loss_from_logits(b, targets)
    (ref * 1.5).backward()
    assert abs(float(loss) - float(ref)) < 1e-2 * abs(float(ref)) + 1e-3
    # the L2Wrap term is not scaled by the upstream gradient; its arg-max may differ only between exact ties
    assert rel_rms(a.grad.float().cpu(), b.grad.cpu()) < 1e-2


def test_fused_rejects_wrong_inputs():
    from visualrwkv_amd import fused
    x = torch.randn(1, 4, 128, device="cuda")          # fp32: must raise, not fall back
    with pytest.raises(ValueError):
        fused.relu_sq(x)


def test_stream_copy_moves_every_byte():
    """vrwkv_stream_copy (the on-box copy ceiling of bench.py's roofline): exact copy, ragged tail of the tile loop."""
    from visualrwkv_amd import hip_lib
    lib = hip_lib.load()
    for nbytes in (16, 32 * 1024 + 48, 5 * 1024 * 1024 + 16):
        src = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda")
        dst = torch.zeros_like(src)
        hip_lib.check(lib.vrwkv_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "copy")
        assert torch.equal(dst, src)
    assert lib.vrwkv_stream_copy(src.data_ptr(), dst.data_ptr(), 24, None) != 0        # not a multiple of 16


@pytest.mark.parametrize("M,K,N", [(4096, 2048, 96), (5000, 96, 2048), (2624, 2048, 64), (3001, 256, 2048), (41984, 2048, 256),
                                   (1000, 128, 32), (777, 160, 1024)])
def test_wgrad_skinny_matches_torch(M, K, N):
    """vrwkv_wgrad_skinny_bf16 (x^T dy for LoRA factors, both orientations, ragged M) against an fp32 product."""
    from visualrwkv_amd import fused
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dy = (torch.randn(M, N, device="cuda", generator=g) * 0.1).bfloat16()
    assert fused.wgrad_skinny_supported(x, dy)
    got = fused.wgrad_skinny(x, dy)
    ref = x.float().t() @ dy.float()
    assert got.shape == (K, N) and got.dtype == torch.bfloat16
    err = (got.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 6e-3, float(err)                    # one bf16 rounding of an fp32 accumulation


def test_lora_mm_gradients_match_plain_matmul():
    from visualrwkv_amd import fused
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 640, 256, device="cuda", generator=g).bfloat16().requires_grad_()
    w1 = (torch.randn(256, 32, device="cuda", generator=g) * 0.1).bfloat16().requires_grad_()
    w2 = (torch.randn(32, 256, device="cuda", generator=g) * 0.1).bfloat16().requires_grad_()
    dy = torch.randn(2, 640, 256, device="cuda", generator=g).bfloat16()
    def run(mm):
        for t in (x, w1, w2):
            t.grad = None
        y = mm(torch.tanh(mm(x, w1)), w2)
        y.backward(dy)
        return y.detach(), x.grad.clone(), w1.grad.clone(), w2.grad.clone()
    a, b = run(fused.lora_mm), run(torch.matmul)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for u, v in zip(a[2:], b[2:]):
        assert float((u.float() - v.float()).abs().max() / v.float().abs().max()) < 1e-2


def test_aliased_outputs_sum_gradients_in_the_kernels():
    """mix_dup3 / kva(dup=True): the aliases' gradients are summed inside mix_bwd / kva_bwd -- same result as autograd's
    own accumulation on the un-aliased functions."""
    from visualrwkv_amd import fused
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).bfloat16()
    B, T, C = 2, 48, 256
    x = rn(B, T, C).requires_grad_()
    mus = [torch.rand(1, 1, C, device="cuda", generator=g).bfloat16().requires_grad_() for _ in range(6)]
    w = [rn(B, T, C) for _ in range(7)]
    def grads(outs, second):
        for t in [x] + mus:
            t.grad = None
        loss = sum((o.float() * wi.float()).sum() for o, wi in zip(outs, w[:6])) + (second.float() * w[6].float()).sum()
        loss.backward()
        return [x.grad.clone()] + [m.grad.clone() for m in mus]
    o = fused.mix(x, *mus)
    ref = grads(o, o[3])
    o = fused.mix_dup3(x, *mus)
    assert len(o) == 7 and o[6].data_ptr() == o[3].data_ptr()
    got = grads(o[:6], o[6])
    for a, b in zip(got, ref):
        assert float((a.float() - b.float()).abs().max()) <= 0.02 * float(b.float().abs().max()) + 1e-3
    # kva
    k, v, vf, vl, al = [rn(B, T, C).requires_grad_() for _ in range(5)]
    k_k, k_a, a0, v0 = [rn(1, 1, C).requires_grad_() for _ in range(4)]
    leaves = [k, v, vf, vl, al, k_k, k_a, a0, v0]
    wk = [rn(B, T, C) for _ in range(6)]
    def kgrads(outs):
        for t in leaves:
            t.grad = None
        sum((o.float() * wi.float()).sum() for o, wi in zip(outs, wk)).backward()
        return [t.grad.clone() for t in leaves]
    k2, v2, z, b = fused.kva(k, v, vf, vl, al, k_k, k_a, a0, v0)
    ref = kgrads([k2, v2, z, b, k2, v2])
    outs = fused.kva(k, v, vf, vl, al, k_k, k_a, a0, v0, True)
    assert len(outs) == 6 and outs[4].data_ptr() == outs[0].data_ptr() and outs[5].data_ptr() == outs[1].data_ptr()
    got = kgrads(outs)
    for a, b in zip(got, ref):
        assert float((a.float() - b.float()).abs().max()) <= 0.02 * float(b.float().abs().max()) + 1e-3


@pytest.mark.parametrize("rows,cols", [(2048, 2048), (8192, 2048), (2048, 65536), (64, 192)])
def test_transpose_kernel_is_exact(rows, cols):
    from visualrwkv_amd import fused
    w = torch.randn(rows, cols, device="cuda").bfloat16()
    assert torch.equal(fused.transpose2d(w), w.t().contiguous())


def test_linear_tn_matches_autograd():
    """fused.linear: same outputs and gradients as nn.Linear's autograd (the input gradient only changes its GEMM layout)."""
    from visualrwkv_amd import fused
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 512, bias=False).cuda().bfloat16()
    x = torch.randn(4, 64, 256, device="cuda").bfloat16().requires_grad_(True)
    gy = torch.randn(4, 64, 512, device="cuda").bfloat16()
    y = fused.linear(lin, x); y.backward(gy)
    gx, gw = x.grad.clone(), lin.weight.grad.clone()
    x.grad = None; lin.weight.grad = None
    y2 = lin(x); y2.backward(gy)
    assert torch.equal(y, y2)
    assert rel_rms(gx.float(), x.grad.float()) < 2e-3 and rel_rms(gw.float(), lin.weight.grad.float()) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,M,has_delta,dup3", [(2, 5, 128, 6, True, False), (3, 33, 768, 6, True, True), (1, 40, 2048, 1, True, False),
                                                    (3, 1, 64, 6, False, False), (7, 592, 256, 6, True, True), (5, 1100, 512, 1, False, False),
                                                    (8, 2624, 2048, 6, True, True), (8, 2624, 2048, 1, True, False)])
def test_ln_mix_is_the_two_kernel_path(B, T, C, M, has_delta, dup3):
    """add + LayerNorm + token shift + lerps in one kernel (fused.add_ln_mix) against add_ln followed by mix: outputs and the
    input gradients bit-identical (same arithmetic, same roundings), parameter gradients to fp32 summation order.  Shapes with
    fewer tokens than workgroups, with ranges that straddle sample boundaries, and the benchmark shape."""
    from visualrwkv_amd import fused
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T + C + M)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    ln = torch.nn.LayerNorm(C).to(dev).bfloat16()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * rn(C)); ln.bias.copy_(0.1 * rn(C))
    mus = [torch.rand(1, 1, C, device=dev, generator=g).bfloat16() for _ in range(M)]
    x0, d0 = rn(B, T, C).bfloat16(), (0.5 * rn(B, T, C)).bfloat16()
    nout = M + (1 if dup3 else 0)
    gouts = [rn(B, T, C).bfloat16() for _ in range(nout)]
    gres = rn(B, T, C).bfloat16()

    def run(fused_path):
        x, d = x0.clone().requires_grad_(True), (d0.clone().requires_grad_(True) if has_delta else None)
        ms = [m.clone().requires_grad_(True) for m in mus]
        for p_ in ln.parameters():
            p_.grad = None
        if fused_path:
            xn, outs = fused.add_ln_mix(x, d, ln, ms, dup3)
        else:
            xn, h = fused.add_ln(x, d, ln)
            outs = list((fused.mix_dup3 if dup3 else fused.mix)(h, *ms))
        torch.autograd.backward([xn, *outs], [gres, *gouts])
        return ([xn, *outs], [x.grad] + ([d.grad] if has_delta else []), [m.grad for m in ms] + [ln.weight.grad.clone(), ln.bias.grad.clone()])

    o_f, gx_f, gp_f = run(True)
    o_r, gx_r, gp_r = run(False)
    for a, b in zip(o_f, o_r):
        assert torch.equal(a, b)
    for a, b in zip(gx_f, gx_r):
        if has_delta:
            assert torch.equal(a, b)
        else:   # without a delta xn is x itself: autograd adds the (rounded) LayerNorm gradient to the residual gradient and rounds again,
            assert rel_rms(a.float().cpu(), b.float().cpu()) < 3e-3      # the kernel adds in fp32 and rounds once
    for a, b in zip(gp_f, gp_r):
        assert rel_rms(a.float().cpu(), b.float().cpu()) < 2e-3       # fp32 partial sums in another order, then one bf16 rounding


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (2048, 512, 256), (4128, 256, 768), (41984, 2048, 2048), (10496, 8192, 2048)])
def test_big_weight_gradient_against_fp32(M, N, K):
    """csrc/wgrad_big.h (dW = dy^T x of the Linear layers, VisualRWKV-v7/v7.00/src/model.py:150-153,214-215) against the fp32
    product rounded once to bf16: one stage, ragged stage counts, the split-K shapes of the step (2048 x 2048 over 41 984 tokens:
    4 slices) and a channel-mix shape; and through fused._LinearTN into a strided slot of a flat buffer."""
    from tests.parity import bf16_close
    from visualrwkv_amd import fused
    g = torch.Generator(device="cuda").manual_seed(M + N)
    dy = (torch.randn(M, N, device="cuda", generator=g) * 0.3).bfloat16()
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    assert fused.wgrad_big_supported(dy, x)
    ref = dy.float().t() @ x.float()
    out = fused.wgrad_big(dy, x)
    bf16_close(out, ref, f"wgrad_big {M}x{N}x{K}", tol=1e-3, max_flip=0.02)
    flat = torch.zeros(N * K + 64, dtype=torch.bfloat16, device="cuda")
    fused.wgrad_big(dy, x, out=flat[32:32 + N * K].view(N, K))
    assert torch.equal(flat[32:32 + N * K].view(N, K), out) and float(flat[:32].abs().sum()) == 0 and float(flat[-32:].abs().sum()) == 0
