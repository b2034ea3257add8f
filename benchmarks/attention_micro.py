"""ViT attention forward: HIP MFMA kernel vs torch SDPA on the tower shapes (B images)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd import hip_attention
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, B, L, H, D in [("siglip", 8, 1024, 16, 72), ("dinov2", 8, 1029, 16, 64), ("sam-window", 200, 196, 12, 64), ("sam-global", 8, 4096, 12, 64)]:
    qkv = torch.randn(B, L, 3, H, D, device="cuda").bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    fl = 4.0 * B * H * L * L * D
    t_h = bench(lambda: hip_attention.flash_forward(q, k, v))
    t_s = bench(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).contiguous())
    print(f"{name:11s} B{B} L{L} H{H} D{D}: hip {t_h:.3f} ms ({fl/t_h/1e9:.0f} TF)   sdpa {t_s:.3f} ms ({fl/t_s/1e9:.0f} TF)")
