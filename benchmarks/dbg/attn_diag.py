import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visualrwkv_amd import hip_attention, attention as att
from visualrwkv_amd.vit import TimmViT
def rr(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
for (B, L, H, D, scale) in [(2, 261, 2, 64, 1.0), (2, 261, 2, 64, 4.0), (2, 256, 2, 64, 1.0), (1, 69, 2, 64, 1.0), (2, 261, 4, 64, 1.0)]:
    g = torch.Generator().manual_seed(L)
    qkv = (scale * torch.randn(B, L, 3, H, D, generator=g)).bfloat16().cuda()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o = hip_attention.flash_forward(q, k, v)
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)).transpose(1, 2)
    nan = torch.isnan(o.float())
    print((B, L, H, D, scale), "rel", rr(o, ref), "nan", int(nan.sum()), "nan rows", sorted(set(nan.any(-1).nonzero()[:, 1].tolist()))[:20])
torch.manual_seed(0)
vit = TimmViT(img_size=224, patch=14, dim=128, depth=2, heads=2, mlp_hidden=256, class_token=True, reg_tokens=4, ls_init=0.5).bfloat16().cuda()
xv = torch.randn(2, 3, 224, 224, device="cuda").bfloat16()
with torch.no_grad():
    x = vit.patch_embed(xv) + vit.pos_embed
    x = torch.cat([vit.cls_token.expand(2, -1, -1), vit.reg_token.expand(2, -1, -1), x], 1)
    for i, blk in enumerate(vit.blocks):
        h = blk.norm1(x)
        qkv = blk.attn.qkv(h).view(2, 261, 3, 2, 64)
        o1 = hip_attention.flash_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
        att.set_hip_attention(False)
        o2 = att.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
        att.set_hip_attention(True)
        print("block", i, "qkv absmax", float(qkv.float().abs().max()), "rel", rr(o1, o2), "nan", int(torch.isnan(o1.float()).sum()), int(torch.isnan(o2.float()).sum()))
        x = blk(x)
        print("   x nan", int(torch.isnan(x.float()).sum()))
