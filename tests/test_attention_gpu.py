"""MFMA flash-attention forward (ViT towers) on the GPU vs fp32 softmax attention on the same bf16 inputs.
bf16 P in the PV product (as every flash kernel) => rel-RMS tolerance 5e-3."""
import pytest
import torch
import torch.nn.functional as F

from oracle.wkv7_oracle import rel_rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L,H,D", [(2, 1024, 16, 72), (2, 1029, 16, 64), (3, 196, 12, 64), (1, 37, 2, 72), (1, 4096, 12, 64)])
def test_flash_forward_matches_fp32(B, L, H, D):
    from visualrwkv_amd import hip_attention
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 3, H, D, generator=g).bfloat16().cuda()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o = hip_attention.flash_forward(q, k, v)
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)).transpose(1, 2)
    assert o.shape == (B, L, H, D) and o.is_contiguous()
    assert rel_rms(o.float().cpu(), ref.cpu()) < 5e-3
    assert torch.isfinite(o.float()).all()


def test_attention_dispatch_uses_hip_kernel_and_towers_agree():
    """`attention()` routes bf16 CUDA inputs without bias to the HIP kernel; a tiny SigLIP/DINOv2-style tower gives
    the same features either way."""
    from visualrwkv_amd import attention as att
    from visualrwkv_amd.vit import TimmViT
    torch.manual_seed(0)
    m = TimmViT(img_size=112, patch=14, dim=144, depth=3, heads=2, mlp_hidden=288, class_token=True, reg_tokens=4, ls_init=0.5).bfloat16().cuda()
    x = torch.randn(2, 3, 112, 112, device="cuda").bfloat16()
    with torch.no_grad():
        att.set_hip_attention(True)
        a = m(x)
        att.set_hip_attention(False)
        b = m(x)
        att.set_hip_attention(True)
    assert rel_rms(a.float().cpu(), b.float().cpu()) < 1e-2
