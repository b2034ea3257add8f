"""Host-side mirror of the reference modules vs fixtures recorded from the reference's own
src/model.py / src/sam.py (tests/golden/make_golden_model.py).  CPU, fp32, no WKV op involved."""
import os
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

from oracle.wkv7_oracle import rel_rms

GOLD = os.path.join(os.path.dirname(__file__), "golden", "model_ref.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


def lm_args(**kw):
    d = dict(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=512, dropout=0,
             grad_cp=0, ctx_len=64, load_model="", num_token_per_image=16, proj_type="mlp")
    d.update(kw)
    return SimpleNamespace(**d)


def test_state_dict_keys_and_shapes_match_reference(gold):
    from visualrwkv_amd.rwkv7 import RWKV
    mine = RWKV(lm_args()).state_dict()
    ref = gold["lm_state_fp32"]
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert mine[k].shape == ref[k].shape, k


def test_initialisers_match_reference_formulas():
    """Deterministic initialisers (x_*, w0, k_k, ...) equal the reference's element loops (model.py:89-160)."""
    from visualrwkv_amd.rwkv7 import RWKV
    import math
    args = lm_args()
    att = RWKV(args).blocks[1].att
    C, L, lid = 128, 2, 1
    r01, r10 = lid / (L - 1), 1.0 - lid / L
    i = 37
    assert math.isclose(att.x_r[0, 0, i].item(), 1.0 - (i / C) ** (0.2 * r10), rel_tol=1e-6)
    assert math.isclose(att.x_k[0, 0, i].item(), 1.0 - ((i / C) ** (0.9 * r10) + 0.4 * r01), rel_tol=1e-6, abs_tol=1e-7)
    assert math.isclose(att.w0[0, 0, i].item(), -7 + 5 * (i / (C - 1)) ** (0.85 + 1.0 * r01 ** 0.5) + 0.5, rel_tol=1e-6)
    assert att.w1.shape == (C, 32) and att.g1.shape == (C, 32) and att.v1.shape == (C, 32)
    assert float(att.output.weight.abs().sum()) == 0.0 and float(att.k_k[0, 0, 0]) == pytest.approx(0.85)
    big = RWKV(lm_args(n_embd=2048, dim_att=2048, n_layer=1, vocab_size=8)).blocks[0].att
    assert big.w1.shape[1] == 96 and big.a1.shape[1] == 96 and big.g1.shape[1] == 256


def test_cmix_fp32(gold):
    from visualrwkv_amd.rwkv7 import RWKV
    m = RWKV(lm_args())
    m.load_state_dict(gold["lm_state_fp32"])
    with torch.no_grad():
        y = m.blocks[1].ffn(gold["mods"]["x"].float())
    assert rel_rms(y, gold["mods"]["cmix1_y_fp32"]) < 1e-6


def test_token_shift_is_bit_exact():
    from visualrwkv_amd.rwkv7 import time_shift
    x = torch.randn(3, 7, 5).bfloat16()
    ref = nn.ZeroPad2d((0, 0, 1, -1))(x)
    assert torch.equal(time_shift(x), ref)
    assert torch.equal(time_shift(x)[:, 0], torch.zeros(3, 5, dtype=torch.bfloat16))
    assert torch.equal(time_shift(x)[:, 1:], x[:, :-1])


def test_projector(gold):
    from visualrwkv_amd.visual import MLPWithContextGating
    p = MLPWithContextGating(48, 128)
    p.load_state_dict(gold["proj"]["state"])
    with torch.no_grad():
        assert rel_rms(p(gold["proj"]["x"]), gold["proj"]["y"]) < 1e-6


def test_loss_and_l2wrap(gold):
    from visualrwkv_amd.visual import VisualRWKV
    g = gold["loss"]
    lg = g["logits"].clone().requires_grad_(True)
    loss = VisualRWKV.loss_from_logits(lg, g["targets"])
    loss.backward()
    assert torch.allclose(loss, g["loss"], rtol=1e-6, atol=1e-7)
    assert rel_rms(lg.grad, g["dlogits"]) < 1e-6


def test_adaptive_pooling_and_scatter(gold):
    from visualrwkv_amd.visual import VisualRWKV
    g = gold["pool"]
    holder = SimpleNamespace(pool=nn.AdaptiveAvgPool2d(g["out_side"]))
    assert rel_rms(VisualRWKV.adaptive_pooling(holder, g["x"]), g["y"]) < 1e-6
    s = gold["scatter"]
    emb = nn.Embedding(65536, 8)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(65536, 8, generator=torch.Generator().manual_seed(s["emb_seed"])))
    holder = SimpleNamespace(rwkv=SimpleNamespace(emb=emb), encode_images=lambda images: s["img_feats"],
                             args=SimpleNamespace(check_image_tokens=True))
    with torch.no_grad():
        y, _ = VisualRWKV.preparing_embedding(holder, {"input_ids": s["ids"], "labels": s["ids"], "images": {}})
    assert torch.equal(y, s["y"])          # indexing is bit-exact


def test_sam_encoder(gold):
    from visualrwkv_amd.vit import SamImageEncoder
    g = gold["sam"]
    m = SamImageEncoder(img_size=128, patch=16, dim=64, depth=3, heads=2, out_chans=16, window=3, global_attn_indexes=(2,))
    missing = m.load_state_dict(g["state"], strict=True)
    with torch.no_grad():
        h = m.patch_embed(g["x"]) + m.pos_embed
        for blk in m.blocks:
            h = blk(h)
        neck = m.neck(h.permute(0, 3, 1, 2))
    assert rel_rms(h, g["tokens"]) < 1e-5
    assert rel_rms(neck, g["neck"]) < 1e-5
    # space-to-depth at 64x64 -> 32x32 is pure indexing: bit-exact
    x = g["ds_x"]
    B, C, H, W = x.shape
    y = x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, H // 2, W // 2, C * 4).permute(0, 3, 1, 2)
    assert torch.equal(y, g["ds_y"])


def test_timm_style_vits_against_transformers():
    """SigLIP / DINOv2 arithmetic lives in timm (absent, unpinned): pin the restatement against the
    architecturally equivalent `transformers` modules with shared random weights."""
    tr = pytest.importorskip("transformers")
    from visualrwkv_amd.vit import TimmViT
    torch.manual_seed(0)
    # --- SigLIP: no cls token, learned pos-embed, tanh-GELU in HF
    cfg = tr.SiglipVisionConfig(hidden_size=64, intermediate_size=96, num_hidden_layers=3, num_attention_heads=2,
                                image_size=56, patch_size=14, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    hf = tr.SiglipVisionModel(cfg).eval()
    mine = TimmViT(img_size=56, patch=14, dim=64, depth=3, heads=2, mlp_hidden=96, class_token=False, reg_tokens=0,
                   ls_init=None, act="gelu_tanh").eval()
    v = hf.vision_model if hasattr(hf, 'vision_model') else hf
    with torch.no_grad():
        mine.patch_embed.proj.weight.copy_(v.embeddings.patch_embedding.weight)
        mine.patch_embed.proj.bias.copy_(v.embeddings.patch_embedding.bias)
        mine.pos_embed.copy_(v.embeddings.position_embedding.weight[None])
        for b, hb in zip(mine.blocks, v.encoder.layers):
            b.norm1.load_state_dict(hb.layer_norm1.state_dict()); b.norm2.load_state_dict(hb.layer_norm2.state_dict())
            b.attn.qkv.weight.copy_(torch.cat([hb.self_attn.q_proj.weight, hb.self_attn.k_proj.weight, hb.self_attn.v_proj.weight]))
            b.attn.qkv.bias.copy_(torch.cat([hb.self_attn.q_proj.bias, hb.self_attn.k_proj.bias, hb.self_attn.v_proj.bias]))
            b.attn.proj.load_state_dict(hb.self_attn.out_proj.state_dict())
            b.mlp.fc1.load_state_dict(hb.mlp.fc1.state_dict()); b.mlp.fc2.load_state_dict(hb.mlp.fc2.state_dict())
        x = torch.randn(2, 3, 56, 56)
        ref = hf(pixel_values=x, output_hidden_states=True).hidden_states[2]      # after block index 1 = depth-2
        got = mine(x)
    assert rel_rms(got, ref) < 1e-5


def test_dinov2_reg_against_transformers():
    tr = pytest.importorskip("transformers")
    if not hasattr(tr, "Dinov2WithRegistersModel"):
        pytest.skip("transformers without Dinov2WithRegisters")
    from visualrwkv_amd.vit import TimmViT
    torch.manual_seed(1)
    cfg = tr.Dinov2WithRegistersConfig(hidden_size=64, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14,
                                       num_register_tokens=4, mlp_ratio=4, layerscale_value=0.3, layer_norm_eps=1e-6)
    hf = tr.Dinov2WithRegistersModel(cfg).eval()
    mine = TimmViT(img_size=56, patch=14, dim=64, depth=3, heads=2, mlp_hidden=256, class_token=True, reg_tokens=4,
                   ls_init=0.3, act="gelu").eval()
    e = hf.embeddings
    with torch.no_grad():
        e.cls_token.normal_(); e.register_tokens.normal_(); e.position_embeddings.normal_(std=0.1)
        e.position_embeddings[:, 0].zero_()                 # timm's no_embed_class layout has no cls position
        mine.patch_embed.proj.load_state_dict(e.patch_embeddings.projection.state_dict())
        mine.pos_embed.copy_(e.position_embeddings[:, 1:])
        mine.cls_token.copy_(e.cls_token); mine.reg_token.copy_(e.register_tokens)
        for b, hb in zip(mine.blocks, hf.encoder.layer):
            a = hb.attention.attention
            b.norm1.load_state_dict(hb.norm1.state_dict()); b.norm2.load_state_dict(hb.norm2.state_dict())
            b.attn.qkv.weight.copy_(torch.cat([a.query.weight, a.key.weight, a.value.weight]))
            b.attn.qkv.bias.copy_(torch.cat([a.query.bias, a.key.bias, a.value.bias]))
            b.attn.proj.load_state_dict(hb.attention.output.dense.state_dict())
            b.mlp.fc1.load_state_dict(hb.mlp.fc1.state_dict()); b.mlp.fc2.load_state_dict(hb.mlp.fc2.state_dict())
            b.ls1.gamma.copy_(hb.layer_scale1.lambda1); b.ls2.gamma.copy_(hb.layer_scale2.lambda1)
        x = torch.randn(2, 3, 56, 56)
        ref = hf(pixel_values=x, output_hidden_states=True).hidden_states[2][:, 5:]   # drop cls + 4 registers
        got = mine(x)
    assert got.shape == (2, 16, 64)
    assert rel_rms(got, ref) < 1e-5


def test_vision_tower_checkpoints_load_or_fail_loudly(tmp_path):
    """`vision_tower_path` (vision.py:58-70, sam.py:498-505): timm-format files with a different pretraining grid and the
    pooling-head keys, a SAM file with the `image_encoder.` prefix and foreign keys; a missing file or a foreign checkpoint
    is an error, never a silent random-weight tower."""
    from visualrwkv_amd.vit import SamDinoSigLIPViTBackbone, SamImageEncoder, TimmViT
    torch.manual_seed(0)
    kw = {"dino": dict(depth=2, dim=32, heads=2), "siglip": dict(depth=2, dim=32, heads=2, mlp_hidden=48),
          "sam": dict(img_size=64, dim=32, depth=2, heads=2, out_chans=8, window=2, global_attn_indexes=(1,))}
    src_dino = TimmViT(42, 14, 32, 2, 2, 128, class_token=True, reg_tokens=4, ls_init=1e-5)      # 3x3 grid "pretraining" size
    src_sig = TimmViT(70, 14, 32, 2, 2, 48, class_token=False, reg_tokens=0, ls_init=None)       # 5x5 grid
    src_sam = SamImageEncoder(**kw["sam"])
    with torch.no_grad():
        for m in (src_dino, src_sig, src_sam):
            for p in m.parameters():
                p.normal_(0, 0.1)
    sd_sig = dict(src_sig.state_dict())
    sd_sig["attn_pool.q.weight"] = torch.zeros(3, 3)
    sd_sig["fc_norm.weight"] = torch.zeros(3)
    torch.save(src_dino.state_dict(), tmp_path / "dino.pth")
    torch.save(sd_sig, tmp_path / "siglip.bin")
    sam_sd = {"image_encoder." + k: v for k, v in src_sam.state_dict().items()}
    sam_sd["mask_decoder.foo"] = torch.zeros(2)
    torch.save(sam_sd, tmp_path / "sam.pth")
    paths = {"dino": str(tmp_path / "dino.pth"), "siglip": str(tmp_path / "siglip.bin"), "sam": str(tmp_path / "sam.pth")}
    bb = SamDinoSigLIPViTBackbone(paths, default_image_size=56, tower_kwargs=kw)                 # 4x4 grid here
    assert torch.equal(bb.dino_featurizer.blocks[1].attn.qkv.weight, src_dino.blocks[1].attn.qkv.weight)
    assert torch.equal(bb.siglip_featurizer.patch_embed.proj.weight, src_sig.patch_embed.proj.weight)
    assert bb.dino_featurizer.pos_embed.shape == (1, 16, 32) and bb.siglip_featurizer.pos_embed.shape == (1, 16, 32)
    ref = torch.nn.functional.interpolate(src_sig.pos_embed.reshape(1, 5, 5, 32).permute(0, 3, 1, 2), size=(4, 4), mode="bicubic",
                                          antialias=True).permute(0, 2, 3, 1).reshape(1, 16, 32)
    assert torch.allclose(bb.siglip_featurizer.pos_embed, ref)
    assert torch.equal(bb.sam_featurizer.blocks[0].attn.rel_pos_h, src_sam.blocks[0].attn.rel_pos_h)
    with pytest.raises(FileNotFoundError):
        SamDinoSigLIPViTBackbone({**paths, "sam": str(tmp_path / "nope.pth")}, default_image_size=56, tower_kwargs=kw)
    with pytest.raises(RuntimeError):
        SamDinoSigLIPViTBackbone({**paths, "dino": paths["siglip"]}, default_image_size=56, tower_kwargs=kw)
    with pytest.raises(KeyError):
        SamDinoSigLIPViTBackbone({"dino": paths["dino"]}, default_image_size=56, tower_kwargs=kw)


@pytest.mark.parametrize("case", ["win14", "glob64"])
def test_sam_attention_restatement_against_reference_fixture(case):
    """The torch statement of SAM's attention + decomposed rel-pos bias (vit._SamAttention, attention.attention_relpos)
    against outputs of the reference's own module at the real window sizes (tests/golden/make_golden_sam_attn.py)."""
    import os
    from visualrwkv_amd.vit import _SamAttention
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sam_attn_ref.pt"), weights_only=True)[case]
    m = _SamAttention(g["dim"], g["heads"], (g["side"], g["side"]))
    m.load_state_dict({k: v.float() for k, v in g["state"].items()}, strict=True)
    with torch.no_grad():
        y = m(g["x"].float())
    tol = 2e-3 if case == "glob64" else 1e-5              # glob64 output is stored as fp16
    assert rel_rms(y, g["y"].float()) < tol
