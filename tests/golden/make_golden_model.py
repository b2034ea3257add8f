"""Generate tests/golden/model_*.pt by importing the reference's own Python (VisualRWKV-v7/v7.00/src/model.py
and src/sam.py) in this container and recording inputs, weights and outputs.

The reference cannot be imported as-is here: it needs pytorch_lightning, deepspeed, timm,
torchvision and JIT-builds a CUDA op at import.  This script pre-seeds sys.modules with inert stand-ins for
those *third-party* packages (LightningModule -> nn.Module etc.), makes `cpp_extension.load` a no-op, and
registers `wind_backstepping::{forward,backward}` for the CPU with the repo's oracle (the reference asserts
bf16 there, so the RWKV fixtures are bf16 runs).  Nothing of the reference's source is stored -- only tensors.

Run where /root/reference exists:   python tests/golden/make_golden_model.py
"""
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/VisualRWKV-v7/v7.00"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import wkv7_c  # noqa: E402


def install_stubs():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    pl.__version__ = "1.9.5"
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_info = lambda *a, **k: None
    plu.rank_zero_warn = lambda *a, **k: None
    pls = types.ModuleType("pytorch_lightning.strategies")
    pls.DeepSpeedStrategy = type("DeepSpeedStrategy", (), {})
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu, "pytorch_lightning.strategies": pls})
    vis = types.ModuleType("src.vision")
    vis.SamDinoSigLIPViTBackbone = type("SamDinoSigLIPViTBackbone", (nn.Module,), {})
    sys.modules["src.vision"] = vis
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: None
    os.environ["RWKV_JIT_ON"] = "0"
    os.environ["RWKV_HEAD_SIZE_A"] = "64"
    lib = torch.library.Library("wind_backstepping", "DEF")
    lib.define("forward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, Tensor(a!) y, Tensor(b!) s, Tensor(c!) sa) -> ()")
    lib.define("backward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, Tensor dy, Tensor s, Tensor sa, "
               "Tensor(a!) dw, Tensor(b!) dq, Tensor(c!) dk, Tensor(d!) dv, Tensor(e!) dz, Tensor(f!) da) -> ()")

    def fwd(w, q, k, v, z, a, y, s, sa):
        yy, ss, ssa = wkv7_c.forward(w, q, k, v, z, a)
        y.copy_(yy); s.copy_(ss); sa.copy_(ssa)

    def bwd(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da):
        outs = wkv7_c.backward(w, q, k, v, z, a, dy, s, sa)
        for dst, src in zip((dw, dq, dk, dv, dz, da), outs):
            dst.copy_(src)

    lib.impl("forward", fwd, "CPU")
    lib.impl("backward", bwd, "CPU")
    return lib


def randomize(module, gen, scale=0.02):
    """The reference zero-initialises several tensors (output/value projections, LoRA A matrices); give every
    all-zero tensor small random values so that no path of the block is dead in the fixture."""
    with torch.no_grad():
        for p in module.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=gen) * scale)


def main():
    _lib = install_stubs()
    from src import model as ref            # the reference module itself
    from src import sam as ref_sam
    g = torch.Generator().manual_seed(7)
    out = {}

    args = SimpleNamespace(n_embd=128, n_layer=2, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=512,
                           dropout=0, grad_cp=0, ctx_len=64, load_model="", num_token_per_image=16, proj_type="mlp")
    torch.manual_seed(1234)
    lm = ref.RWKV(args)
    randomize(lm, g)
    with torch.no_grad():
        lm.blocks[0].att.r_k.copy_(torch.randn(lm.blocks[0].att.r_k.shape, generator=g) * 0.1)
        lm.blocks[1].att.r_k.copy_(torch.randn(lm.blocks[1].att.r_k.shape, generator=g) * 0.1)
    out["lm_state_fp32"] = {k: v.clone() for k, v in lm.state_dict().items()}
    lm = lm.bfloat16()
    B, T = 2, 37                                           # 37 -> left-padded to 48 (exercises pad/unpad)
    x = (torch.randn(B, T, 128, generator=g) * 0.5).bfloat16().requires_grad_(True)
    logits = lm(x)
    gout = (torch.randn(logits.shape, generator=g) * 0.1).bfloat16()
    logits.backward(gout)
    out["lm"] = {"x": x.detach().clone(), "logits": logits.detach().clone(), "gout": gout, "dx": x.grad.clone(),
                 "grads": {n: p.grad.clone() for n, p in lm.named_parameters()
                           if n in ("blocks.1.att.w0", "blocks.1.att.r_k", "blocks.0.att.key.weight", "blocks.1.att.a1",
                                    "blocks.1.ffn.x_k", "blocks.0.ln0.weight", "blocks.1.att.v2", "blocks.0.att.k_k")}}

    # single modules (bf16, T multiple of 16): Tmix layer 0 / layer 1, CMix, Block
    x16 = (torch.randn(2, 32, 128, generator=g) * 0.5).bfloat16()
    vf = (torch.randn(2, 32, 128, generator=g) * 0.5).bfloat16()
    with torch.no_grad():
        y0, vf0 = lm.blocks[0].att(x16, torch.empty_like(x16))
        y1, vf1 = lm.blocks[1].att(x16, vf)
        c1 = lm.blocks[1].ffn(x16)
        b1, _ = lm.blocks[1](x16, vf)
    out["mods"] = {"x": x16, "v_first": vf, "tmix0_y": y0, "tmix0_vfirst": vf0, "tmix1_y": y1, "cmix1_y": c1, "block1_y": b1}

    # fp32 pieces that do not touch the WKV op
    lm32 = ref.RWKV(args)
    lm32.load_state_dict(out["lm_state_fp32"])
    with torch.no_grad():
        out["mods"]["cmix1_y_fp32"] = lm32.blocks[1].ffn(x16.float())
    proj = ref.MLPWithContextGating(48, 128)
    xin = torch.randn(3, 16, 48, generator=g)
    with torch.no_grad():
        out["proj"] = {"state": {k: v.clone() for k, v in proj.state_dict().items()}, "x": xin, "y": proj(xin)}

    # loss + L2Wrap (src/model.py:418-434,257-271) on fixed logits/targets
    lg = (torch.randn(2, 12, 50, generator=g)).requires_grad_(True)
    tg = torch.randint(0, 50, (2, 12), generator=g)
    tg[0, :7] = -100
    tg[1, :] = -100                                       # a sample without any valid label
    holder = type("H", (), {"__call__": lambda self, batch: (lg, tg)})()
    loss = ref.VisualRWKV.training_step(holder, None, 0)
    loss.backward()
    out["loss"] = {"logits": lg.detach().clone(), "targets": tg, "loss": loss.detach().clone(), "dlogits": lg.grad.clone()}

    # adaptive_pooling + preparing_embedding scatter (src/model.py:442-447,473-494)
    feats = torch.randn(2, 64, 24, generator=g)
    pool_holder = SimpleNamespace(pool=nn.AdaptiveAvgPool2d(3))
    pooled = ref.VisualRWKV.adaptive_pooling(pool_holder, feats)
    out["pool"] = {"x": feats, "y": pooled, "out_side": 3}
    emb = nn.Embedding(65536, 8)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(65536, 8, generator=torch.Generator().manual_seed(99)))
    ids = torch.randint(0, 65535, (2, 10), generator=g)
    ids[0, 2:5] = 65535
    ids[1, 0:3] = 65535
    img_feats = torch.randn(2, 3, 8, generator=g)
    pe_holder = SimpleNamespace(rwkv=SimpleNamespace(emb=emb), encode_images=lambda images: img_feats)
    with torch.no_grad():
        emb_out, _ = ref.VisualRWKV.preparing_embedding(pe_holder, {"input_ids": ids, "labels": ids, "images": {}, "sample_id": ["a", "b"]})
    out["scatter"] = {"emb_seed": 99, "ids": ids, "img_feats": img_feats, "y": emb_out}

    # SAM encoder (in-repo reference), tiny config, fp32
    torch.manual_seed(5)
    sam = ref_sam.ImageEncoderViT(img_size=128, patch_size=16, embed_dim=64, depth=3, num_heads=2, mlp_ratio=4, out_chans=16,
                                  qkv_bias=True, norm_layer=lambda d: nn.LayerNorm(d, eps=1e-6), use_rel_pos=True,
                                  window_size=3, global_attn_indexes=(2,))
    with torch.no_grad():
        for n, p in sam.named_parameters():
            if "rel_pos" in n or "pos_embed" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    # the reference's LosslessDownSampler hard-codes a 32x32 output grid (sam.py:73), so call the pieces
    img = torch.randn(2, 3, 128, 128, generator=g)
    with torch.no_grad():
        h = sam.patch_embed(img) + sam.pos_embed
        for blk in sam.blocks:
            h = blk(h)
        neck = sam.neck(h.permute(0, 3, 1, 2))
    out["sam"] = {"state": {k: v.clone() for k, v in sam.state_dict().items()}, "x": img, "tokens": h, "neck": neck}
    # space-to-depth at the real 64x64 -> 32x32 geometry
    xds = torch.randn(1, 4, 64, 64, generator=g)
    out["sam"]["ds_x"] = xds
    out["sam"]["ds_y"] = ref_sam.LosslessDownSampler(2)(xds)

    torch.save(out, os.path.join(HERE, "model_ref.pt"))
    print("wrote model_ref.pt", {k: type(v).__name__ for k, v in out.items()})


if __name__ == "__main__":
    main()
