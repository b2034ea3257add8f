"""The HIP optimizer kernels (csrc/fused_ops.hip: vrwkv_adamw_step_bf16, vrwkv_sqnorm_bf16) on the GPU against
torch.optim.AdamW + clip_grad_norm_ in fp32 -- DeepSpeed FusedAdam(adam_w_mode=True) semantics as configured by the
reference (VisualRWKV-v7/v7.00/src/model.py:390-410, train.py:92)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def test_sqnorm_kernel(hip_lib):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1 << 20, device="cuda", generator=g).bfloat16()
    out = torch.zeros(1, device="cuda")
    rc = hip_lib.vrwkv_sqnorm_bf16(x.numel(), x.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = x.double().pow(2).sum()
    assert abs(float(out) - float(ref)) / float(ref) < 1e-5


def test_adamw_kernel_matches_torch_adamw_over_5_steps():
    """Zero1Engine on the GPU drives the HIP kernels (bf16 params/grads, fp32 master + moments, two weight-decay groups,
    clip 1.0).  Reference: fp32 torch.optim.AdamW fed the same bf16 gradients; compare the fp32 master weights."""
    from visualrwkv_amd.dp import Zero1Engine
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(64, 256), nn.LayerNorm(256), nn.Tanh(), nn.Linear(256, 64)).cuda().bfloat16()
    ref = [p.detach().float().clone().requires_grad_(True) for p in m.parameters()]
    wd = [p for p in ref if len(p.squeeze().shape) >= 2]
    nowd = [p for p in ref if len(p.squeeze().shape) < 2]
    opt = torch.optim.AdamW([{"params": wd, "weight_decay": 0.1}, {"params": nowd, "weight_decay": 0.0}],
                            lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    eng = Zero1Engine(m, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1, grad_clip=1.0, bucket_mb=0.01)
    assert len(eng.buckets) > 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(32, 64, device="cuda", generator=g).bfloat16()
    y = torch.randn(32, 64, device="cuda", generator=g).bfloat16()
    for step in range(5):
        eng.zero_grad()
        (((m(x) - y).float() ** 2).mean() * 30).backward()          # gradient norm > 1: the clip is active
        grads = [p.grad.detach().float().clone() for p in m.parameters()]
        gn = eng.step()
        for r, gr in zip(ref, grads):
            r.grad = gr
        total = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        assert abs(gn - float(total)) / float(total) < 1e-3
        assert float(total) > 1.0 or step > 0
        opt.step()
        # published bf16 parameters = rounded fp32 master of the reference trajectory
        for p, r in zip(m.parameters(), ref):
            assert torch.allclose(p.detach().float(), r.detach().bfloat16().float(), rtol=0, atol=1e-2 * float(r.abs().max()))
    # fp32 masters: every bucket piece against the reference parameters laid out the same way
    flat_ref = torch.zeros(eng.numel, device="cuda")
    ref_by_id = {id(p): r for p, r in zip(m.parameters(), ref)}
    for p, o in zip(eng.params, eng.offsets):
        flat_ref[o:o + p.numel()] = ref_by_id[id(p)].detach().reshape(-1)
    for b in eng.buckets:
        s = b.start + eng.rank * b.piece
        assert torch.allclose(b.master, flat_ref[s:s + b.piece], rtol=2e-5, atol=2e-6)
