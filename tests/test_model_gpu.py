"""RWKV-7 modules on the GPU (bf16, HIP WKV7 kernel) against fixtures recorded from the reference's own
src/model.py running in bf16 on the CPU with the oracle as its WKV op (tests/golden/make_golden_model.py).

Both sides are bf16 eager pipelines whose GEMMs accumulate in different orders, so module outputs are compared
with a bf16-level tolerance (the reference fixture itself differs from its own fp32 evaluation by 2.4e-3
rel-RMS); the WKV7 kernel's own 1e-3 bar is tested in test_wkv7_gpu.py."""
import os
from types import SimpleNamespace

import pytest
import torch

from oracle.wkv7_oracle import rel_rms
from tests.parity import group_bias

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "model_ref.pt")
TOL = 1e-2


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


def _lm(gold, fused=False, grad_cp=0):
    from visualrwkv_amd.rwkv7 import RWKV
    args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=512,
                           dropout=0, grad_cp=grad_cp, ctx_len=64, load_model="", fused=fused)
    m = RWKV(args)
    m.load_state_dict(gold["lm_state_fp32"])
    return m.bfloat16().cuda()


@pytest.mark.parametrize("fused", [False, True])
def test_modules_match_reference(gold, fused):
    m = _lm(gold, fused)
    g = gold["mods"]
    x, vf = g["x"].cuda(), g["v_first"].cuda()
    with torch.no_grad():
        y0, vf0 = m.blocks[0].att(x, torch.empty_like(x))
        y1, _ = m.blocks[1].att(x, vf)
        c1 = m.blocks[1].ffn(x)
        b1, _ = m.blocks[1](x, vf)
    assert rel_rms(y0.float().cpu(), g["tmix0_y"].float()) < TOL
    assert rel_rms(vf0.float().cpu(), g["tmix0_vfirst"].float()) < TOL
    assert rel_rms(y1.float().cpu(), g["tmix1_y"].float()) < TOL
    assert rel_rms(c1.float().cpu(), g["cmix1_y"].float()) < TOL
    assert rel_rms(b1.float().cpu(), g["block1_y"].float()) < TOL


@pytest.mark.parametrize("fused,grad_cp", [(False, 0), (False, 1), (True, 0), (True, 1), (True, 2)])      # 1 = every block re-computed (the reference's --grad_cp 1), 2 = selective recompute (fused path; eager: same as 1)
def test_lm_forward_backward_with_padding(gold, fused, grad_cp):
    """RWKV.forward on T=37 (left-padded to 48 with emb(261), model.py:286-312) + backward."""
    m = _lm(gold, fused, grad_cp)
    g = gold["lm"]
    x = g["x"].cuda().requires_grad_(True)
    logits = m(x)
    assert logits.shape == g["logits"].shape
    logits.backward(g["gout"].cuda())
    assert rel_rms(logits.detach().float().cpu(), g["logits"].float()) < TOL
    assert rel_rms(x.grad.float().cpu(), g["dx"].float()) < 2 * TOL
    named = dict(m.named_parameters())
    for n, ref in g["grads"].items():
        assert rel_rms(named[n].grad.float().cpu(), ref.float()) < 3 * TOL, n


def test_selective_recompute_keeps_less_and_computes_the_same(gold):
    """grad_cp=2 of the fused path (fused.blocks_forward): the WKV7 chunk checkpoints / sa and relu(h)^2 are not kept for the backward
    -- less memory held between forward and backward -- and the gradients are those of grad_cp=0 (the recompute runs the same kernels;
    only y comes from the by-product-free forward entry, another instantiation of the same algorithm)."""
    g = gold["lm"]
    res = {}
    for mode in (0, 2):
        m = _lm(gold, True, mode)
        x = torch.cat([g["x"]] * 3, dim=1).cuda().requires_grad_(True)          # T = 111 -> 112: seven chunks
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        logits = m(x)
        held = torch.cuda.memory_allocated() - base - logits.numel() * logits.element_size()
        gout = torch.cat([g["gout"]] * 3, dim=1).cuda()
        logits.backward(gout)
        res[mode] = (held, logits.detach().float().cpu(), x.grad.float().cpu(), {n: p.grad.float().cpu() for n, p in m.named_parameters() if p.grad is not None})
    assert res[2][0] < 0.9 * res[0][0], (res[0][0], res[2][0])
    assert rel_rms(res[2][1], res[0][1]) < 2e-3 and rel_rms(res[2][2], res[0][2]) < 4e-3
    for n, gr in res[0][3].items():
        assert rel_rms(res[2][3][n], gr) < 4e-3, n


def test_full_visual_step_runs_and_learns():
    """Tiny VisualRWKV end to end on the GPU: ViT -> pool -> projector -> scatter -> LM -> loss -> ZeRO-1 step."""
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.visual import VisualRWKV
    args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=65536,
                           dropout=0, grad_cp=0, ctx_len=48, num_token_per_image=16, vision_towers=("dino", "siglip", "sam"),
                           vision_image_size=56, load_model="", proj_type="mlp", weight_decay=0.0, fused=True,
                           vision_tower_kwargs={"dino": dict(depth=3, dim=64, heads=1), "siglip": dict(depth=3, dim=64, heads=1, mlp_hidden=96),
                                                "sam": dict(img_size=128, dim=64, depth=3, heads=1, out_chans=16, window=3, global_attn_indexes=(2,))})
    torch.manual_seed(0)
    m = VisualRWKV(args)
    with torch.no_grad():
        for p in m.rwkv.parameters():
            if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    m = m.bfloat16().cuda()
    m.freeze_emb()
    eng = Zero1Engine(m, lr=3e-3, weight_decay=0.0, grad_clip=1.0, bucket_mb=1.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    ids = torch.randint(0, 1000, (2, 48), device="cuda", generator=g)
    ids[:, 2:18] = 65535
    labels = ids.clone(); labels[:, :20] = -100
    batch = {"input_ids": ids, "labels": labels, "sample_id": ["0", "1"],
             "images": {"dino": torch.randn(2, 3, 56, 56, device="cuda").bfloat16(), "siglip": torch.randn(2, 3, 56, 56, device="cuda").bfloat16(),
                        "sam": torch.randn(2, 3, 128, 128, device="cuda").bfloat16()}}
    losses = []
    for _ in range(8):
        eng.zero_grad()
        loss = m.training_step(batch)
        loss.backward()
        eng.step()
        losses.append(float(loss))
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]


def test_full_visual_step_matches_an_independent_fp32_cpu_evaluation(monkeypatch):
    """End-to-end parity, not a property: one training step of a tiny VisualRWKV (three towers -> pool -> projector -> ln_v +
    scatter -> 2 RWKV-7 blocks -> head -> loss with L2Wrap) through the whole MI355X product path (patch-embed, attention,
    rel-pos attention, pool / gate / ln-scatter kernels, fused glue, WKV7 kernels, T,N input gradients, fused loss) against
    the same model evaluated in fp32 on the CPU with the eager modules and the ORACLE's naive recurrence as the WKV7 operator
    (differentiated by autograd).  Loss, the gradient of every trainable parameter group, and the loss after one SGD-like
    ZeRO-1 AdamW step."""
    from oracle.wkv7_oracle import wkv7_naive
    from visualrwkv_amd import rwkv7
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.visual import VisualRWKV

    def mk(fused_flag):
        args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=65536,
                               dropout=0, grad_cp=0, ctx_len=48, num_token_per_image=16, vision_towers=("dino", "siglip", "sam"),
                               vision_image_size=56, load_model="", proj_type="mlp", weight_decay=0.0, fused=fused_flag,
                               check_image_tokens=not fused_flag,
                               vision_tower_kwargs={"dino": dict(depth=3, dim=64, heads=1), "siglip": dict(depth=3, dim=64, heads=1, mlp_hidden=96),
                                                    "sam": dict(img_size=128, dim=64, depth=3, heads=1, out_chans=16, window=3, global_attn_indexes=(2,))})
        torch.manual_seed(0)
        m = VisualRWKV(args)
        with torch.no_grad():
            for p in m.rwkv.parameters():
                if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                    p.normal_(0, 0.02)
        m.freeze_emb()
        return m

    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 1000, (2, 48), generator=g)
    ids[:, 2:18] = 65535
    labels = ids.clone(); labels[:, :20] = -100
    imgs = {"dino": torch.randn(2, 3, 56, 56, generator=g).bfloat16(), "siglip": torch.randn(2, 3, 56, 56, generator=g).bfloat16(),
            "sam": torch.randn(2, 3, 128, 128, generator=g).bfloat16()}

    # ---- reference evaluation: CPU, fp32, eager modules, oracle recurrence behind RUN_CUDA_RWKV7g's signature
    def run_cpu(q, w, k, v, a, b):
        B, T, HC = q.shape
        ops = [i.view(B, T, HC // 64, 64) for i in (w, q, k, v, a, b)]
        return wkv7_naive(*ops)[0].reshape(B, T, HC).to(q.dtype)
    ref = mk(False).float()
    with monkeypatch.context() as mp:
        mp.setattr(rwkv7, "RUN_CUDA_RWKV7g", run_cpu)
        loss_ref = ref.training_step({"input_ids": ids, "labels": labels, "sample_id": ["0", "1"], "images": {k: v.float() for k, v in imgs.items()}})
        loss_ref.backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}

    # ---- product path
    m = mk(True).bfloat16().cuda()
    batch = {"input_ids": ids.cuda(), "labels": labels.cuda(), "sample_id": ["0", "1"], "images": {k: v.cuda() for k, v in imgs.items()}}
    eng = Zero1Engine(m, lr=1e-3, weight_decay=0.0, grad_clip=1.0, bucket_mb=1.0)
    eng.zero_grad()
    loss = m.training_step(batch)
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-2 * abs(float(loss_ref)), (float(loss), float(loss_ref))    # observed 11.3125 vs 11.3385
    named = dict(m.named_parameters())
    checked, errs, bias = 0, {}, {}
    # observed on MI355X (VRWKV_TEST_NOTES=1): scale - 1 = -0.4e-3 .. -2.7e-3, the same sign in every group: d(loss)/d(logits) is stored
    # in bf16 as in the reference's bf16 pipeline, and its dominant entries -w (1 - p_label) have nearly the same value in every row
    # of this batch (27-28 valid tokens per sample), so their rounding error (up to 2^-9) does not average out; a wrong term or factor
    # in one group (1 % and more) still fails
    SCALE_ERR = 8e-3
    for n, gr in gref.items():
        if gr.abs().max() == 0 or gr.numel() < 64:
            continue
        got = named[n].grad
        assert got is not None, n
        errs[n] = rel_rms(got.float().cpu(), gr)
        if gr.numel() >= 1024:                 # unbiasedness of every larger gradient group: a systematic 1 % error would show here
            bias[n] = group_bias(got.float(), gr, n, max_scale_err=SCALE_ERR)[0]
        checked += 1
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    if os.environ.get("VRWKV_TEST_NOTES") == "1":
        print("[parity] e2e worst parameter-gradient groups:", [(n, round(e, 4)) for n, e in worst])
        print("[parity] e2e largest |scale - 1| of a gradient group:", sorted(((abs(b), n) for n, b in bias.items()), reverse=True)[:5])
    assert all(e < 2.6e-2 for e in errs.values()), worst     # observed: worst 2.1e-2 (a token-shift mix parameter); bf16 path vs fp32
    assert checked >= 30
    # the common-mode part of that bias (the bf16 d(loss)/d(logits) every gradient descends from) is the same factor in EVERY group, so a
    # group is also held against the others: its scale may differ from the median scale of all groups by 2.5e-3 (observed on MI355X: groups
    # between -0.4e-3 and -3.1e-3 around a median of -2.0e-3, i.e. deviations up to 1.6e-3) -- a 0.5 % error in ONE group fails here
    # although it passes the absolute bound above
    med = sorted(bias.values())[len(bias) // 2]
    off = {n: b - med for n, b in bias.items() if abs(b - med) >= 2.5e-3}
    assert not off, (med, off)
    # ---- one optimizer step on both sides (CPU: torch AdamW + clip_grad_norm_; GPU: ZeRO-1 engine, HIP AdamW with the clip
    # factor formed on the device), then the loss again
    train = [p for p in ref.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(train, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0)
    torch.nn.utils.clip_grad_norm_(train, 1.0)
    opt.step()
    eng.step(1e-3)
    with monkeypatch.context() as mp, torch.no_grad():
        mp.setattr(rwkv7, "RUN_CUDA_RWKV7g", run_cpu)
        loss_ref2 = ref.training_step({"input_ids": ids, "labels": labels, "sample_id": ["0", "1"], "images": {k: v.float() for k, v in imgs.items()}})
    with torch.no_grad():
        loss2 = m.training_step(batch)
    assert float(loss_ref2) < float(loss_ref) and float(loss2) < float(loss)
    assert abs(float(loss2) - float(loss_ref2)) < 1.5e-2 * abs(float(loss_ref2)), (float(loss2), float(loss_ref2))


def test_weight_gradients_written_into_the_flat_buffer(monkeypatch):
    """fused._LinearTN writes a Linear weight's gradient straight into its slot of the ZeRO-1 flat buffer (no bucket copy): the
    buffer must hold the same bits as with fresh gradient tensors copied in, for one and for two backward passes per step (the
    second accumulates in place), and `.grad` must be the flat view."""
    from visualrwkv_amd import fused
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.rwkv7 import RWKV
    args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=512, dropout=0,
                           grad_cp=0, ctx_len=32, fused=True, weight_decay=0.0)
    torch.manual_seed(0)
    m = RWKV(args)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    m = m.bfloat16().cuda()
    eng = Zero1Engine(m, lr=1e-3, weight_decay=0.0, grad_clip=1.0, bucket_mb=0.05)
    ids = torch.randint(0, 512, (2, 32), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

    def grads(flat, passes):
        monkeypatch.setattr(fused, "FLAT_WGRAD", flat)
        eng.zero_grad()
        if passes == 0:                 # two forward passes under ONE backward: every weight has two gradient producers in the graph
            (m(m.emb(ids)).float().square().mean() + m(m.emb(ids.flip(1))).float().square().mean()).backward()
        for _ in range(passes):
            m(m.emb(ids)).float().square().mean().backward()
        torch.cuda.synchronize()
        in_place = sum(p.grad is not None and p.grad.data_ptr() == eng._view(k).data_ptr() for k, p in enumerate(eng.params))
        for b in eng.buckets:           # whatever is still stashed goes into the buffer, as step() would do
            eng._flush(b)
        return eng.flat_grad.clone(), in_place

    for passes in (1, 2, 0):
        g1, n1 = grads(True, passes)
        g0, n0 = grads(False, passes)
        if passes == 0:     # the two gradients of a weight are added by autograd in either order and rounded once more: not bit-equal
            assert rel_rms(g1.float().cpu(), g0.float().cpu()) < 2e-3
        else:
            assert torch.equal(g1, g0)
        assert n1 > n0 if passes == 1 else True          # Linear weights sit in the buffer before any copy


def test_flat_buffer_alias_never_leaves_the_engine_step():
    """a weight-gradient GEMM may write into the ZeRO-1 flat buffer only between the engine's zero_grad() and step().
    torch.autograd.grad after a step (or on a model whose engine was closed) must get tensors of its own, not views of that buffer."""
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.rwkv7 import RWKV
    args = SimpleNamespace(n_embd=256, n_layer=1, dim_att=256, head_size_a=64, head_size_divisor=8, vocab_size=512, dropout=0,
                           grad_cp=0, ctx_len=32, fused=True, weight_decay=0.0)
    torch.manual_seed(0)
    m = RWKV(args).bfloat16().cuda()
    eng = Zero1Engine(m, lr=1e-3, weight_decay=0.0, grad_clip=1.0, bucket_mb=0.05)
    ids = torch.randint(0, 512, (2, 32), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    lo, hi = eng.flat_grad.data_ptr(), eng.flat_grad.data_ptr() + eng.flat_grad.numel() * 2
    w = m.blocks[0].ffn.key.weight
    eng.zero_grad()
    m(m.emb(ids)).float().square().mean().backward()
    assert lo <= w.grad.data_ptr() < hi                      # armed: the gradient sits in the buffer
    eng.step(1e-3)
    (g,) = torch.autograd.grad(m(m.emb(ids)).float().square().mean(), [w])
    assert not (lo <= g.data_ptr() < hi)                     # disarmed by step(): a tensor of the caller's own
    eng.close()
    assert not hasattr(w, "_vrwkv_flat_grad")
