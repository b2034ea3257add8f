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


def _sam_attention_module(g, device, dtype):
    from visualrwkv_amd.vit import _SamAttention
    m = _SamAttention(g["dim"], g["heads"], (g["side"], g["side"]))
    m.load_state_dict({k: v.float() for k, v in g["state"].items()}, strict=True)
    return m.to(device=device, dtype=dtype)


@pytest.mark.parametrize("case", ["win14", "glob64"])
def test_sam_relpos_attention_against_reference_fixture(case):
    """SAM attention with the decomposed relative-position bias computed inside the MFMA kernel (no (B,H,L,L) tensor)
    against outputs of the reference's own Attention module (tests/golden/make_golden_sam_attn.py; sam.py:245-305,392-426)
    at the two window sizes of the SAM ViT-B tower.  bf16 activations => 1e-2 rel-RMS."""
    import os
    from visualrwkv_amd import attention as att
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sam_attn_ref.pt"), weights_only=True)[case]
    m = _sam_attention_module(g, "cuda", torch.bfloat16)
    x = g["x"].cuda()
    calls = []
    from visualrwkv_amd import hip_attention
    orig = hip_attention.flash_forward_relpos
    hip_attention.flash_forward_relpos = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            y = m(x)
            for qt in (1, 2):                            # every query-tile count of the kernel
                assert hip_attention.hip_lib.load().vrwkv_attention_set_qtiles(qt) == 0
                y_qt = m(x)
                assert rel_rms(y_qt.float().cpu(), g["y"].float()) < 1e-2, qt
            hip_attention.hip_lib.load().vrwkv_attention_set_qtiles(0)
            att.set_hip_attention(False)
            y_eager = m(x)
            att.set_hip_attention(True)
    finally:
        hip_attention.flash_forward_relpos = orig
        hip_attention.hip_lib.load().vrwkv_attention_set_qtiles(0)
    assert len(calls) == 3                                # the HIP kernel is the path that ran
    assert rel_rms(y.float().cpu(), g["y"].float()) < 1e-2
    assert rel_rms(y_eager.float().cpu(), g["y"].float()) < 2e-2
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("qt", [1, 2])
@pytest.mark.parametrize("B,L,H,D", [(2, 1024, 16, 72), (1, 1029, 8, 64), (3, 196, 12, 64), (1, 37, 2, 72)])
def test_flash_forward_query_tile_variants(B, L, H, D, qt):
    from visualrwkv_amd import hip_attention
    lib = hip_attention.hip_lib.load()
    g = torch.Generator().manual_seed(L + qt)
    qkv = torch.randn(B, L, 3, H, D, generator=g).bfloat16().cuda()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert lib.vrwkv_attention_set_qtiles(qt) == 0
    try:
        o = hip_attention.flash_forward(q, k, v)
    finally:
        lib.vrwkv_attention_set_qtiles(0)
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)).transpose(1, 2)
    assert rel_rms(o.float().cpu(), ref.cpu()) < 5e-3
