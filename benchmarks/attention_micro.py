"""ViT attention forward: HIP MFMA kernels (query-tile variants) vs torch SDPA on the tower shapes; the SAM shapes also
with the decomposed relative-position bias (in-kernel vs SDPA fed the materialised (B,H,L,L) bias).  One JSON line per
case: ms and dense TFLOP/s (4 B H L^2 D flops) against the 2.5 PFLOP/s bf16 MFMA peak."""
import json, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd import hip_attention, hip_lib
from visualrwkv_amd.attention import rel_table

def bench(fn, iters=None):
    iters = iters or ITERS
    for _ in range(1 if HIP_ONLY else 3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

lib = hip_lib.load()
HIP_ONLY = "--hip-only" in sys.argv
ITERS = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
for name, B, L, H, D, S in [("siglip", 16, 1024, 16, 72, 0), ("dinov2", 16, 1029, 16, 64, 0), ("sam-window", 400, 196, 12, 64, 0),
                            ("sam-global", 16, 4096, 12, 64, 0), ("sam-window+relpos", 400, 196, 12, 64, 14),
                            ("sam-global+relpos", 16, 4096, 12, 64, 64)]:
    qkv = torch.randn(B, L, 3, H, D, device="cuda").bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    fl = 4.0 * B * H * L * L * D
    row = {"case": name, "B": B, "L": L, "H": H, "D": D}
    if S:
        rh = (0.3 * torch.randn(2 * S - 1, D, device="cuda")).bfloat16()
        rw = (0.3 * torch.randn(2 * S - 1, D, device="cuda")).bfloat16()
        fn = lambda: hip_attention.flash_forward_relpos(q, k, v, rh, rw, S)
    else:
        fn = lambda: hip_attention.flash_forward(q, k, v)
    for qt in (() if HIP_ONLY else (1, 2)):
        lib.vrwkv_attention_set_qtiles(qt)
        t = bench(fn)
        row[f"hip_qt{qt}_ms"] = round(t, 4); row[f"hip_qt{qt}_TFLOPs"] = round(fl / t / 1e9, 1)
    lib.vrwkv_attention_set_qtiles(0)
    t = bench(fn)
    row["hip_ms"] = round(t, 4); row["hip_TFLOPs"] = round(fl / t / 1e9, 1); row["frac_of_mfma_peak"] = round(fl / t / 1e9 / 2500.0, 3)
    if not HIP_ONLY and (S == 0 or B * H * L * L * 2 < 8e9):
        def sdpa():
            bias = None
            if S:
                rq = q.reshape(B, S, S, H, D)
                bh = torch.einsum("bhwnc,hkc->bnhwk", rq, rel_table(S, rh))
                bw_ = torch.einsum("bhwnc,wkc->bnhwk", rq, rel_table(S, rw))
                bias = (bh[..., :, None] + bw_[..., None, :]).reshape(B, H, L, L)
            return F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias).transpose(1, 2).contiguous()
        t = bench(sdpa, iters=5)
        row["sdpa_ms"] = round(t, 4); row["sdpa_TFLOPs"] = round(fl / t / 1e9, 1)
    print(json.dumps(row), flush=True)
