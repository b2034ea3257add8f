"""Stateful generation on the MI355X: the single-token WKV7 step kernel (C-ABI vrwkv_wkv7_step_bf16) against the
oracle recurrence, its consistency with the training forward kernel, and the stateful model path."""
from types import SimpleNamespace

import pytest
import torch

from oracle.wkv7_oracle import make_inputs, rel_rms, wkv7_naive

pytestmark = pytest.mark.gpu


def _step_inputs(B, H, seed):
    w, q, k, v, z, a, _ = make_inputs(B, 16, H, seed=seed)
    return [x[:, 7].contiguous() for x in (w, q, k, v, z, a)]


@pytest.mark.parametrize("B,H", [(1, 1), (3, 5), (8, 32)])
def test_step_kernel_matches_oracle(B, H):
    from visualrwkv_amd import wkv7
    ins = _step_inputs(B, H, seed=B * 100 + H)
    g = torch.Generator().manual_seed(5)
    s0 = torch.randn(B, H, 64, 64, generator=g) * 0.3
    y_ref, s_ref = wkv7_naive(*[x.float().unsqueeze(1) for x in ins], state0=s0.clone())
    s = s0.cuda()
    y = wkv7.wkv7_step(*[x.cuda() for x in ins], s)
    torch.cuda.synchronize()
    assert rel_rms(s.cpu(), s_ref) < 2e-6                     # fp32 state; v_exp_f32 vs libm in the decay
    err = (y.float().cpu() - y_ref[:, 0]).abs()
    assert bool((err <= y_ref[:, 0].abs() * 2 ** -8 + 1e-6).all())   # one bf16 rounding of the fp32 result


def test_step_rejects_bad_arguments():
    from visualrwkv_amd import wkv7
    ins = [x.cuda() for x in _step_inputs(2, 2, 0)]
    s = torch.zeros(2, 2, 64, 64, device="cuda")
    with pytest.raises(ValueError):
        wkv7.wkv7_step(*ins, s.half())
    with pytest.raises(ValueError):
        wkv7.wkv7_step(ins[0].float(), *ins[1:], s)
    with pytest.raises(ValueError):
        wkv7.wkv7_step(*ins, s[:, :1])
    with pytest.raises(NotImplementedError):
        wkv7.wkv7_step(*[x.cpu() for x in ins], s.cpu())


def test_steps_reproduce_training_forward():
    """T single-token steps from S = 0 == the training forward kernel (y per token and the chunk-end states)."""
    from visualrwkv_amd import wkv7
    B, T, H = 2, 48, 4
    ins = [x.cuda() for x in make_inputs(B, T, H, seed=11)[:6]]
    y_full, s_end = wkv7.wkv7_prefill(*ins)
    s = torch.zeros(B, H, 64, 64, device="cuda")
    ys = [wkv7.wkv7_step(*[x[:, t].contiguous() for x in ins], s) for t in range(T)]
    y_steps = torch.stack(ys, dim=1)
    assert rel_rms(y_steps.float(), y_full.float()) < 4e-3    # both round y to bf16; chunked kernel is bf16x3
    assert rel_rms(s, s_end) < 1e-3
    # prefill on the first 32 tokens + 16 steps ends in the same state
    _, s32 = wkv7.wkv7_prefill(*[x[:, :32].contiguous() for x in ins])
    for t in range(32, T):
        wkv7.wkv7_step(*[x[:, t].contiguous() for x in ins], s32)
    assert rel_rms(s32, s_end) < 1e-3


def _lm(fused, C=256):
    from visualrwkv_amd.rwkv7 import RWKV
    args = SimpleNamespace(n_embd=C, n_layer=3, dim_att=C, head_size_a=64, head_size_divisor=8, vocab_size=1000,
                           dropout=0, grad_cp=0, ctx_len=128, load_model="", fused=fused)
    torch.manual_seed(2)
    m = RWKV(args)
    with torch.no_grad():
        for b in m.blocks:
            b.att.output.weight.normal_(0, 0.03)
            b.ffn.value.weight.normal_(0, 0.03)
    return m.bfloat16().cuda().eval()


@pytest.mark.parametrize("splits", [[64], [37, 1, 1, 1, 24], [3, 61]])
def test_model_stateful_equals_full_forward(splits):
    m = _lm(fused=True)
    x = torch.randn(2, 64, 256, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        full = m(x).float()
    state, outs = None, []
    for n in splits:
        p0 = sum(o.size(1) for o in outs)
        o, state = m.forward_stateful(x[:, p0:p0 + n], state)
        outs.append(o)
    got = torch.cat(outs, dim=1).float()
    assert rel_rms(got, full) < 2e-2                          # bf16 activations, different kernel for the glue
    agree = (got.argmax(-1) == full.argmax(-1)).float().mean().item()
    assert agree > 0.9


def test_generate_stateful_first_token_matches_generate():
    from visualrwkv_amd.visual import VisualRWKV
    from visualrwkv_amd.rwkv7 import IMAGE_TOKEN_INDEX
    args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=65536,
                           dropout=0, grad_cp=0, ctx_len=128, num_token_per_image=16, vision_towers=("dino",),
                           vision_image_size=56, load_model="", proj_type="mlp", fused=True,
                           vision_tower_kwargs={"dino": dict(depth=2, dim=64, heads=1)})
    torch.manual_seed(0)
    m = VisualRWKV(args)
    with torch.no_grad():
        for b in m.rwkv.blocks:
            b.att.output.weight.normal_(0, 0.05)
            b.ffn.value.weight.normal_(0, 0.05)
    m = m.bfloat16().cuda().eval()
    ids = torch.randint(0, 256, (1, 27), device="cuda")
    ids[0, 3:19] = IMAGE_TOKEN_INDEX
    images = {"dino": torch.randn(1, 3, 56, 56, device="cuda", dtype=torch.bfloat16)}
    ref = m.generate(ids, images, False, 1.0, 1.0, 1, stop_token_idx=-7)
    got = m.generate_stateful(ids, images, False, 1.0, 1.0, 8, stop_token_idx=-7)
    assert len(got[0]) == 8 and all(0 <= t < 65536 for t in got[0])
    assert got[0][0] == ref[0][0]
    assert got[1][0] == pytest.approx(ref[1][0], rel=3e-2, abs=3e-2)


def test_forward_from_state_matches_oracle():
    from visualrwkv_amd import wkv7
    B, T, H = 2, 48, 3
    ins = make_inputs(B, T, H, seed=31)[:6]
    g = torch.Generator().manual_seed(6)
    s0 = torch.randn(B, H, 64, 64, generator=g) * 0.3
    y_ref, s_ref = wkv7_naive(*[x.double() for x in ins], state0=s0.double())
    y, s = wkv7.wkv7_forward_state(*[x.cuda() for x in ins], s0.cuda())
    assert rel_rms(y.double().cpu(), y_ref) < 4e-3 and rel_rms(s.double().cpu(), s_ref) < 2e-5
    y0, s_0 = wkv7.wkv7_forward_state(*[x.cuda() for x in ins])
    y_ref0, s_ref0 = wkv7_naive(*[x.double() for x in ins])
    assert rel_rms(y0.double().cpu(), y_ref0) < 4e-3 and rel_rms(s_0.double().cpu(), s_ref0) < 2e-5


@pytest.mark.parametrize("with_state", [False, True])
def test_tparallel_forward_equals_sequential(with_state):
    """Sequence-parallel forward (three launches over T-segments) == the sequential kernel."""
    from visualrwkv_amd import wkv7
    B, T, H, P = 1, 512, 4, 4
    ins = [x.cuda() for x in make_inputs(B, T, H, seed=41)[:6]]
    s0 = (torch.randn(B, H, 64, 64, generator=torch.Generator().manual_seed(8)) * 0.3).cuda() if with_state else None
    y_seq, s_seq = wkv7.wkv7_forward_state(*ins, s0)
    y_par, s_par = wkv7.wkv7_forward_tparallel(*ins, s0, segments=P)
    assert rel_rms(y_par.float(), y_seq.float()) < 4e-3          # both rounded to bf16
    assert rel_rms(s_par, s_seq) < 1e-4
    assert wkv7.tparallel_segments(1, 32, 2624) >= 3 and wkv7.tparallel_segments(16, 32, 2624) == 1
    with pytest.raises(ValueError):
        wkv7.wkv7_forward_tparallel(*ins, None, segments=5)


def test_stateful_chunk_continuation_uses_state():
    """A second multi-chunk call continues from the carried state (previously only a fresh context used the chunked kernel)."""
    m = _lm(fused=True)
    x = torch.randn(1, 96, 256, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        full = m(x).float()
    o1, st = m.forward_stateful(x[:, :32], None)
    o2, st = m.forward_stateful(x[:, 32:], st)
    got = torch.cat((o1, o2), dim=1).float()
    assert rel_rms(got, full) < 2e-2


def test_graph_decoder_matches_eager_steps():
    """The HIP-graph-captured decode step advances the same state and returns the same logits as the eager step."""
    m = _lm(fused=True)
    x = torch.randn(1, 40, 256, device="cuda", dtype=torch.bfloat16)
    _, st_a = m.forward_stateful(x[:, :32], None)
    _, st_b = m.forward_stateful(x[:, :32], None)
    dec = m.make_decoder(st_b)
    for t in range(32, 40):
        la, st_a = m.forward_stateful(x[:, t:t + 1], st_a, last_only=True)
        lb = dec(x[:, t:t + 1])
        assert rel_rms(lb.float(), la.float()) < 1e-3
    for a, b in zip(st_a.S + st_a.att_x, st_b.S + st_b.att_x):
        assert rel_rms(b.float(), a.float()) < 1e-3
    assert st_b.n_tokens == st_a.n_tokens == 40


@pytest.mark.parametrize("B", [1, 3])
def test_gemv_multi_matches_torch(B):
    from visualrwkv_amd import decode
    g = torch.Generator(device="cuda").manual_seed(B)
    rn = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.3).bfloat16()
    jobs = [(rn(2048, 512), rn(B, 512), None, decode.ACT_NONE), (rn(96, 512), rn(B, 512), None, decode.ACT_TANH),
            (rn(512, 96), rn(B, 96), rn(B, 512), decode.ACT_NONE), (rn(70, 64), rn(B, 64), None, decode.ACT_SIGMOID),
            (rn(1000, 2048), rn(B, 2048), None, decode.ACT_RELUSQ), (rn(33, 8), rn(B, 8), None, decode.ACT_NONE),
            (rn(301, 8192), rn(B, 8192), rn(B, 301), decode.ACT_NONE)]             # rows split over the waves
    ys = decode.gemv_multi(jobs, B, torch.device("cuda"))
    for (W, x, res, act), y in zip(jobs, ys):
        ref = x.float() @ W.float().t()
        ref = [ref, torch.tanh(ref), torch.sigmoid(ref), torch.relu(ref) ** 2][act]
        if res is not None:
            ref = ref + res.float()
        assert rel_rms(y.float(), ref) < 5e-3, (W.shape, act)


@pytest.mark.parametrize("C", [256, 512])                  # 512: LayerNorm folded into the GEMV launches
def test_decode_step_path_matches_module_path(C):
    """The batched-GEMV decode step (decode.py) against the module-level stateful step on the same state."""
    m = _lm(fused=True, C=C)
    x = torch.randn(2, 40, C, device="cuda", dtype=torch.bfloat16)
    _, st_a = m.forward_stateful(x[:, :32], None)
    _, st_b = m.forward_stateful(x[:, :32], None)
    m.args.fused = False                                   # module-level path (plain torch glue, wkv7_step)
    ref = []
    for t in range(32, 40):
        lg, st_a = m.forward_stateful(x[:, t:t + 1], st_a, last_only=True)
        ref.append(lg)
    m.args.fused = True
    for i, t in enumerate(range(32, 40)):
        lg, st_b = m.forward_stateful(x[:, t:t + 1], st_b, last_only=True)
        assert rel_rms(lg.float(), ref[i].float()) < 2e-2
    for a, b in zip(st_a.S, st_b.S):
        assert rel_rms(b, a) < 2e-2
    for a, b in zip(st_a.att_x + st_a.ffn_x, st_b.att_x + st_b.ffn_x):      # the carried token-shift rows
        assert rel_rms(b.float(), a.float()) < 2e-2


@pytest.mark.parametrize("B,C,M", [(1, 256, 6), (3, 2048, 1), (2, 4096, 6)])
def test_decode_ln_mix_matches_torch(B, C, M):
    """vrwkv_decode_ln_mix_bf16 against LayerNorm + shift + lerps in fp32 (src/model.py:169-173,250)."""
    from visualrwkv_amd import decode
    g = torch.Generator(device="cuda").manual_seed(C + M)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, prev = (rn(B, C) * 2 + 0.3).bfloat16(), rn(B, C).bfloat16()
    ln = torch.nn.LayerNorm(C).cuda().bfloat16()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * rn(C)); ln.bias.copy_(0.1 * rn(C))
    mus = [torch.rand(1, 1, C, device="cuda", generator=g).bfloat16() for _ in range(M)]
    h = torch.nn.functional.layer_norm(x.float(), (C,), ln.weight.float(), ln.bias.float(), ln.eps).bfloat16().float()
    want = [h + (prev.float() - h) * m.view(C).float() for m in mus]
    carried = prev.clone()
    got = decode.ln_mix(x, ln, carried, mus)
    assert rel_rms(carried.float(), h) < 3e-3
    for a, b in zip(got, want):
        assert rel_rms(a.float(), b) < 5e-3


@pytest.mark.parametrize("B,layer", [(1, 0), (3, 1)])
def test_decode_tmix_head_matches_unfused_chain(B, layer):
    """vrwkv_decode_tmix_head_bf16 against the kernels it fuses (LoRA second stage in torch, decay / kva / wkv7_step /
    post through their own C-ABI entries) on the same inputs and state."""
    from visualrwkv_amd import decode, fused, wkv7
    m = _lm(fused=True).blocks[layer].att
    C, H = 256, 4
    g = torch.Generator(device="cuda").manual_seed(10 + B)
    rn = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.5).bfloat16()
    with torch.no_grad():                      # the default init leaves several of these at zero
        W2 = [m.w2, m.a2, m.g2] + ([m.v2] if layer > 0 else [])
        for p in W2:
            p.copy_(rn(*p.shape) * 0.2)
    r, k, v, vf = rn(B, C), rn(B, C), rn(B, C), rn(B, C)
    hidden = [torch.tanh(rn(B, m.w2.shape[0]).float()).bfloat16(), rn(B, m.a2.shape[0]),
              torch.sigmoid(rn(B, m.g2.shape[0]).float()).bfloat16()] + ([rn(B, m.v2.shape[0])] if layer > 0 else [])
    S0 = torch.randn(B, H, 64, 64, device="cuda", generator=g) * 0.1
    sh = (B, 1, C)
    lo = [(h @ W).view(sh) for h, W in zip(hidden, W2)]
    w = fused.decay(lo[0], m.w0)
    if layer == 0:
        k2, z, b = fused.kva(k.view(sh), None, None, None, lo[1], m.k_k, m.k_a, m.a0, None)
        v2 = v.view(sh)
    else:
        k2, v2, z, b = fused.kva(k.view(sh), v.view(sh), vf.view(sh), lo[3], lo[1], m.k_k, m.k_a, m.a0, m.v0)
    S_ref = S0.clone()
    y = wkv7.wkv7_step(*[t.reshape(B, H, 64).contiguous() for t in (w, r, k2, v2, z, b)], S_ref)
    want = fused.post(y.view(sh), r.view(sh), k2, v2, lo[2], m.ln_x.weight, m.ln_x.bias, m.r_k, m.ln_x.eps).view(B, C)
    S = S0.clone()
    got = decode.tmix_head(m, r, k, v, vf if layer > 0 else None, hidden, S)
    assert rel_rms(S, S_ref) < 2e-3
    assert rel_rms(got.float(), want.float()) < 1e-2


@pytest.mark.parametrize("B,K", [(1, 2048), (3, 512), (2, 4096)])
def test_gemv_ln_multi_matches_torch(B, K):
    """vrwkv_gemv_ln_multi_bf16: LayerNorm + shift + lerp folded into the GEMV input, against fp32 torch."""
    from visualrwkv_amd import decode
    g = torch.Generator(device="cuda").manual_seed(K + B)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, prev = (rn(B, K) * 2 + 3.0).bfloat16(), rn(B, K).bfloat16()        # a large row mean: the shifted-sum statistics
    ln = torch.nn.LayerNorm(K).cuda().bfloat16()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * rn(K)); ln.bias.copy_(0.1 * rn(K))
    jobs = [((rn(n, K) * 0.05).bfloat16(), torch.rand(1, 1, K, device="cuda", generator=g).bfloat16(), act)
            for n, act in ((300, decode.ACT_NONE), (64, decode.ACT_TANH), (1030, decode.ACT_RELUSQ), (8, decode.ACT_SIGMOID))]
    carried = prev.clone()
    ys, h = decode.gemv_ln_multi(jobs, x, ln, carried)
    assert torch.equal(carried, prev)                                        # only read by this launch
    h_ref = torch.nn.functional.layer_norm(x.float(), (K,), ln.weight.float(), ln.bias.float(), ln.eps).bfloat16().float()
    assert rel_rms(h.float(), h_ref) < 3e-3
    for (W, mu, act), y in zip(jobs, ys):
        xin = (h_ref + (prev.float() - h_ref) * mu.view(K).float()).bfloat16().float()
        ref = xin @ W.float().t()
        ref = [ref, torch.tanh(ref), torch.sigmoid(ref), torch.relu(ref) ** 2][act]
        assert rel_rms(y.float(), ref) < 8e-3, (W.shape, act)


def test_gemv_copy_side_job_and_head_carry():
    from visualrwkv_amd import decode
    g = torch.Generator(device="cuda").manual_seed(5)
    rn = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.3).bfloat16()
    W, x, src, dst = rn(100, 1024), rn(2, 1024), rn(2, 2048), rn(2, 2048)
    (y,) = decode.gemv_multi_copy([(W, x, None, decode.ACT_NONE)], 2, torch.device("cuda"), src, dst)
    assert torch.equal(dst, src)
    assert rel_rms(y.float(), x.float() @ W.float().t()) < 5e-3
    m = _lm(fused=True).blocks[1].att
    r, k, v, vf = rn(2, 256), rn(2, 256), rn(2, 256), rn(2, 256)
    hidden = [rn(2, m.w2.shape[0]), rn(2, m.a2.shape[0]), rn(2, m.g2.shape[0]), rn(2, m.v2.shape[0])]
    S1, S2 = torch.zeros(2, 4, 64, 64, device="cuda"), torch.zeros(2, 4, 64, 64, device="cuda")
    src, dst = rn(2, 256), rn(2, 256)
    a = decode.tmix_head(m, r, k, v, vf, hidden, S1)
    b = decode.tmix_head(m, r, k, v, vf, hidden, S2, carry=(src, dst))
    assert torch.equal(a, b) and torch.equal(S1, S2) and torch.equal(dst, src)


def test_decoder_for_reuses_the_captured_graph_across_prompts():
    m = _lm(fused=True)
    x = torch.randn(1, 48, 256, device="cuda", dtype=torch.bfloat16)
    _, st1 = m.forward_stateful(x[:, :16], None)
    d1 = m.decoder_for(st1)
    for t in range(16, 20):
        d1(x[:, t:t + 1])
    _, st2 = m.forward_stateful(x[:, 16:48].flip(1).contiguous()[:, :32], None)         # another prompt
    _, ref = m.forward_stateful(x[:, 16:48].flip(1).contiguous()[:, :32], None)
    d2 = m.decoder_for(st2)
    assert d2 is d1 and d2.state is st1 and d2.state.n_tokens == 32
    for t in range(3):
        tok = x[:, t:t + 1]
        want, ref = m.forward_stateful(tok, ref, last_only=True)
        got = d2(tok)
        assert rel_rms(got.float(), want.float()) < 1e-3
    with torch.no_grad():
        m.blocks[0].att.w0.add_(0.01)                      # parameters changed in place: a new capture
    assert m.decoder_for(st2) is not d1
