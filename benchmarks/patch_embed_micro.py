"""Patch embedding: implicit-GEMM HIP kernel (bias + position embedding fused) vs the eager statement (6-D permute copy +
F.linear + add) at the tower shapes, 16 images.  One JSON line per case."""
import json, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd import fused

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, B, side, p, N in [("siglip", 16, 448, 14, 1152), ("dinov2", 16, 448, 14, 1024), ("sam", 16, 1024, 16, 768)]:
    x = torch.randn(B, 3, side, side, device="cuda").bfloat16()
    w = (0.03 * torch.randn(N, 3, p, p, device="cuda")).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    g = side // p
    pos = torch.randn(g * g, N, device="cuda").bfloat16()
    wp = fused.padded_patch_weight(w)
    def eager():
        u = x.view(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
        return F.linear(u, w.view(N, -1), bias) + pos
    t_h = bench(lambda: fused.patch_embed(x, w, bias, pos, None, padded_weight=wp))
    t_e = bench(eager)
    fl = 2.0 * B * g * g * 3 * p * p * N
    print(json.dumps({"case": name, "B": B, "side": side, "patch": p, "N": N, "hip_ms": round(t_h, 4), "hip_TFLOPs": round(fl / t_h / 1e9, 1),
                      "eager_ms": round(t_e, 4), "eager_TFLOPs": round(fl / t_e / 1e9, 1)}), flush=True)
