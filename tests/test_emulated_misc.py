"""csrc/fused_ops.hip (AdamW / squared norm), csrc/loss_fused.hip (cross-entropy + L2Wrap) and csrc/visual_ops.hip (adaptive pool,
context gate) compiled whole for the host lockstep emulator: the product's vrwkv_* entry points on CPU tensors, against torch.
The GPU suite holds the same comparisons on the device (tests/test_optim_gpu.py, test_fused_gpu.py, test_visual_gpu.py); these run
without one."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

L, I, F32, VP = ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def P(t):
    return VP(t.data_ptr()) if t is not None else VP(0)


def call(lib, name, argtypes, *args):
    f = getattr(lib, name)
    f.argtypes, f.restype = argtypes, I
    rc = f(*args)
    assert rc == 0, (name, rc)


def rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))


def test_adamw_steps_match_torch_adamw(emu_lib):
    """FusedAdam(adam_w_mode=True) semantics (src/model.py:390-410): fp32 master / m / v, bf16 gradient in, bf16 parameter out,
    weight decay only below wd_boundary; five steps against torch.optim.AdamW on the fp32 master copy."""
    g = torch.Generator().manual_seed(0)
    n, nwd = 4096, 2048
    master = torch.randn(n, generator=g)
    ref = master.clone().requires_grad_(True)
    opt_wd = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1)
    m, v = torch.zeros(n), torch.zeros(n)
    param = master.bfloat16()
    ref_nowd = master.clone().requires_grad_(True)
    opt_nowd = torch.optim.AdamW([ref_nowd], lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g).bfloat16()
        call(emu_lib, "vrwkv_adamw_step_bf16", [L, VP, VP, VP, VP, VP, F32, F32, F32, F32, F32, I, F32, L, L, VP],
             n, P(master), P(m), P(v), P(grad), P(param), 3e-3, 0.9, 0.99, 1e-8, 0.1, step, 1.0, 0, nwd, None)
        for r, o in ((ref, opt_wd), (ref_nowd, opt_nowd)):
            r.grad = grad.float()
            o.step()
    want = torch.cat([ref.detach()[:nwd], ref_nowd.detach()[nwd:]])
    assert rel(master, want) < 1e-5
    assert torch.equal(param, master.bfloat16())


def test_sqnorm_and_clipped_step(emu_lib):
    g = torch.Generator().manual_seed(1)
    n = 8192
    x = (3 * torch.randn(n, generator=g)).bfloat16()
    out = torch.zeros(1)
    call(emu_lib, "vrwkv_sqnorm_bf16", [L, VP, VP, VP], n, P(x), P(out), None)
    assert abs(float(out) - float(x.double().square().sum())) < 1e-3 * float(x.double().square().sum())
    # clip 1.0 read from device memory: the step sees grad * min(1, clip / ||g||)
    master = torch.randn(n, generator=g)
    m, v, param = torch.zeros(n), torch.zeros(n), master.bfloat16()
    m2, v2, master2, param2 = m.clone(), v.clone(), master.clone(), param.clone()
    call(emu_lib, "vrwkv_adamw_step_clip_bf16", [L, VP, VP, VP, VP, VP, F32, F32, F32, F32, F32, I, VP, F32, F32, L, L, VP],
         n, P(master), P(m), P(v), P(x), P(param), 1e-3, 0.9, 0.99, 1e-8, 0.0, 1, P(out), 1.0, 1.0, 0, 0, None)
    scale = min(1.0, 1.0 / float(out.sqrt()))
    call(emu_lib, "vrwkv_adamw_step_bf16", [L, VP, VP, VP, VP, VP, F32, F32, F32, F32, F32, I, F32, L, L, VP],
         n, P(master2), P(m2), P(v2), P(x), P(param2), 1e-3, 0.9, 0.99, 1e-8, 0.0, 1, scale, 0, 0, None)
    assert rel(m, m2) < 1e-5 and rel(master, master2) < 1e-6


@pytest.mark.parametrize("rows,V", [(5, 512), (3, 4096), (2, 1000)])
def test_cross_entropy_and_l2wrap(emu_lib, rows, V):
    """training_step's shifted CE (ignore_index -100) + L2Wrap's gradient (src/model.py:418-434,257-271), per row."""
    g = torch.Generator().manual_seed(rows + V)
    logits = (2 * torch.randn(rows, V, generator=g)).bfloat16()
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[0] = -100
    loss, mx, lse = torch.zeros(rows), torch.zeros(rows), torch.zeros(rows)
    arg = torch.zeros(rows, dtype=torch.int32)
    call(emu_lib, "vrwkv_ce_fwd_bf16", [L, I] + [VP] * 7, rows, V, P(logits), P(labels), P(loss), P(mx), P(lse), P(arg), None)
    lf = logits.float()
    want = F.cross_entropy(lf, labels, reduction="none", ignore_index=-100)
    assert torch.allclose(loss, want, rtol=1e-5, atol=1e-5)
    assert torch.equal(arg.long(), lf.argmax(-1)) and torch.equal(mx, lf.max(-1).values)
    roww = torch.tensor([0.0] + [0.25] * (rows - 1))
    dlog = torch.zeros_like(logits)
    l2 = 1e-4 / rows
    call(emu_lib, "vrwkv_ce_bwd_bf16", [L, I] + [VP] * 6 + [F32, VP, VP], rows, V, P(logits), P(labels), P(roww), P(mx), P(lse), P(arg), l2, P(dlog), None)
    ref = roww[:, None] * (torch.softmax(lf, -1) - F.one_hot(labels.clamp_min(0), V).float() * (labels >= 0)[:, None])
    ref[torch.arange(rows), lf.argmax(-1)] += lf.max(-1).values * l2
    assert rel(dlog, ref) < 3e-3


def test_adaptive_pool_and_gate(emu_lib):
    g = torch.Generator().manual_seed(4)
    B, sin, sout, D = 2, 8, 3, 64
    x = torch.randn(B, sin * sin, D, generator=g).bfloat16()
    y = torch.zeros(B, sout * sout, D, dtype=torch.bfloat16)
    call(emu_lib, "vrwkv_adaptive_pool_bf16", [I, I, I, I, VP, VP, VP], B, sin, sout, D, P(x), P(y), None)
    want = F.adaptive_avg_pool2d(x.float().view(B, sin, sin, D).permute(0, 3, 1, 2), sout).permute(0, 2, 3, 1).reshape(B, sout * sout, D)
    assert rel(y, want) < 3e-3
    n = 512
    a, gt, do = [torch.randn(n, generator=g).bfloat16() for _ in range(3)]
    out = torch.zeros_like(a)
    call(emu_lib, "vrwkv_gate_fwd_bf16", [L, VP, VP, VP, VP], n, P(a), P(gt), P(out), None)
    assert rel(out, a.float() * torch.sigmoid(gt.float())) < 3e-3
    dg, dx = torch.zeros_like(a), torch.zeros_like(a)
    call(emu_lib, "vrwkv_gate_bwd_bf16", [L, VP, VP, VP, VP, VP, VP], n, P(a), P(gt), P(do), P(dg), P(dx), None)
    s = torch.sigmoid(gt.float())
    assert rel(dx, do.float() * s) < 3e-3 and rel(dg, do.float() * a.float() * s * (1 - s)) < 4e-3


def test_gelu_exact_and_tanh_against_torch(emu_lib):
    """csrc/visual_ops.hip gelu_kernel: the towers' nn.GELU (timm Mlp via src/vision.py:123-134; src/sam.py MLPBlock), both approximations, in place.
    Tolerance: the correctly rounded bf16 of the fp64 value, except at rounding ties."""
    g = torch.Generator().manual_seed(6)
    n = 4096
    x = (torch.randn(n, generator=g) * 3).bfloat16()
    x[:16] = torch.tensor([0.0, -0.0, 1e-3, -1e-3, 0.5, -0.5, 1, -1, 2, -2, 4, -4, 8, -8, 30, -30]).bfloat16()
    for tanh in (0, 1):
        y = x.clone()
        call(emu_lib, "vrwkv_gelu_bf16", [L, VP, VP, I, VP], n, P(y), P(y), tanh, None)
        want64 = F.gelu(x.double(), approximate="tanh" if tanh else "none")     # fp64: torch's fp32 kernel forms 1 + erf(.) and loses the negative tail to cancellation
        want = want64.bfloat16()
        assert torch.isfinite(y.float()).all()
        assert ((y.double() - want64).abs() <= want64.abs() * 2.0 ** -8 * 1.01 + 1e-12).all(), "more than half a bf16 step from the exact value"
        big = want64.abs() > 1e-10                           # below, even the fp64 reference has lost 1 + erf(.) to cancellation
        assert (y != want)[big].float().mean() < 0.002       # rounding ties only
        t32 = F.gelu(x.float(), approximate="tanh" if tanh else "none")          # and torch's own fp32 kernel agrees to its accuracy
        assert ((y.float() - t32).abs() <= (t32.abs() * 2.0 ** -7).clamp_min(1e-6)).all()
    f = emu_lib.vrwkv_gelu_bf16
    assert f(12, P(y), P(y), 0, None) == -2 and f(0, P(y), P(y), 0, None) == -1 and f(8, None, P(y), 0, None) == -1


def test_wkv7_single_token_step_against_the_recurrence(emu_lib):
    """csrc/wkv7_step.hip: five tokens stepped one by one through the carried (B,H,64,64) state against the reference's per-token
    recurrence (RWKV-v7_simple.py:20-32 as oracle.wkv7_oracle.wkv7_naive states it) -- y to bf16 rounding, the state to fp32."""
    from oracle.wkv7_oracle import make_inputs, wkv7_naive
    B, T, H = 2, 5, 3
    w, q, k, v, z, a, _ = make_inputs(B, T, H, seed=9)
    state = torch.zeros(B, H, 64, 64)
    ys = []
    for t in range(T):
        y = torch.zeros(B, H, 64, dtype=torch.bfloat16)
        args = [x[:, t].contiguous() for x in (w, q, k, v, z, a)]
        call(emu_lib, "vrwkv_wkv7_step_bf16", [I, I] + [VP] * 9, B, H, *[P(x) for x in args], P(state), P(y), None)
        ys.append(y)
    y_ref, s_ref = wkv7_naive(*[x.double() for x in (w, q, k, v, z, a)])
    assert rel(torch.stack(ys, 1), y_ref) < 3e-3
    assert rel(state, s_ref) < 1e-5


@pytest.mark.parametrize("B", [1, 3])
def test_decode_gemv_jobs(emu_lib, B):
    """csrc/gemv_decode.hip: several y = act(x W^T) (+ residual) products of one decode step in one launch (short and long rows, rows
    split over waves, every activation) on the emulator against fp32 torch -- the CPU twin of tests/test_stateful_gpu.py::
    test_gemv_multi_matches_torch."""
    g = torch.Generator().manual_seed(B)
    rn = lambda *s: (torch.randn(*s, generator=g) * 0.3).bfloat16()
    jobs = [(rn(256, 512), rn(B, 512), None, 0), (rn(96, 512), rn(B, 512), None, 1), (rn(512, 96), rn(B, 96), rn(B, 512), 0),
            (rn(70, 64), rn(B, 64), None, 2), (rn(120, 2048), rn(B, 2048), None, 3), (rn(33, 8), rn(B, 8), None, 0),
            (rn(37, 8192), rn(B, 8192), rn(B, 37), 0)]
    n = len(jobs)
    ys = [torch.zeros(B, W.shape[0], dtype=torch.bfloat16) for W, _, _, _ in jobs]
    arr = lambda ts: (VP * n)(*[t.data_ptr() if t is not None else 0 for t in ts])
    Ns = (I * n)(*[W.shape[0] for W, _, _, _ in jobs])
    Ks = (I * n)(*[W.shape[1] for W, _, _, _ in jobs])
    acts = (I * n)(*[a for _, _, _, a in jobs])
    call(emu_lib, "vrwkv_gemv_multi_bf16", [I, I] + [VP] * 8, n, B, arr([j[0] for j in jobs]), arr([j[1] for j in jobs]),
         arr([j[2] for j in jobs]), arr(ys), Ns, Ks, acts, None)
    for (W, x, res, act), y in zip(jobs, ys):
        ref = x.float() @ W.float().t()
        ref = [ref, torch.tanh(ref), torch.sigmoid(ref), torch.relu(ref) ** 2][act]
        if res is not None:
            ref = ref + res.float()
        assert rel(y, ref) < 5e-3, (tuple(W.shape), act)
