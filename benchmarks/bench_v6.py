"""BASELINE config 4 as a parity/sanity run (not the bench.py line): VisualRWKV-6 7B (L32, C4096, WKV6 kernels) +
CLIP ViT-L/14-336 (random init from the transformers config), 576+1 image tokens + 2048 text tokens, bf16, full train
step with the ZeRO-1 engine on one MI355X.   python benchmarks/bench_v6.py [--micro-bsz 2] [--layers 32]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--micro-bsz", type=int, default=2)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--n-embd", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grad-cp", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1, help="fused RWKV-6 glue kernels (ddmix, gn_silu, add+LN, loss)")
    a = ap.parse_args()
    import transformers
    from visualrwkv_amd import build, wkv6
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
    from visualrwkv_amd.visual6 import IMAGE_TOKEN_INDEX, VisualRWKV6
    build.build()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    enable_tuned_gemms()
    C = a.n_embd
    args = SimpleNamespace(n_embd=C, dim_att=C, n_layer=a.layers, head_size_a=64, head_size_divisor=8,
                           dim_ffn=int((C * 3.5) // 32 * 32), vocab_size=65536, dropout=0, grad_cp=a.grad_cp, ctx_len=4096,
                           load_model="", grid_size=-1, fused=bool(a.fused))
    clip_cfg = transformers.CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                             num_attention_heads=16, image_size=336, patch_size=14)
    torch.manual_seed(42)
    with torch.device(dev):
        model = VisualRWKV6(args, transformers.CLIPVisionModel(clip_cfg), 1024)
    with torch.no_grad():
        for p in model.rwkv.parameters():
            if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.01)
    model = model.to(torch.bfloat16)
    model.freeze_emb()
    engine = Zero1Engine(model, lr=2e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, grad_clip=1.0, bucket_mb=200.0)
    B, T_text = a.micro_bsz, 2048
    g = torch.Generator(device=dev).manual_seed(7)
    ids = torch.randint(0, 65535, (B, T_text + 1), device=dev, generator=g)
    ids[:, 4] = IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, : T_text // 2] = -100
    labels[ids == IMAGE_TOKEN_INDEX] = -100
    batch = {"input_ids": ids, "labels": labels,
             "images": torch.randn(B, 1, 3, 336, 336, device=dev, generator=g).bfloat16()}

    def step():
        engine.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        engine.step(2e-5)
        return loss

    losses = []
    for _ in range(a.warmup):
        loss = step()
        losses.append(round(float(loss.detach()), 3))
    torch.cuda.synchronize()
    from visualrwkv_amd import wkv6
    wkv6.EVENT_LOG = []              # HIP events on the launch stream around every WKV6 launch of the timed steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    log, wkv6.EVENT_LOG = wkv6.EVENT_LOG, None
    losses = [float(x) if torch.is_tensor(x) else x for x in losses]
    dt = (time.perf_counter() - t0) / a.steps
    T = T_text + 577
    n_par = sum(p.numel() for p in model.rwkv.parameters())
    # in-step WKV6 kernels against the 8 TB/s roofline, bytes as benchmarks/wkv6_micro.py counts them (r,k,v bf16 + ew f32 + y, plus the
    # 16 B / element chunk-state checkpoint this implementation writes in the forward and reads in the backward)
    FWD_B, BWD_B = 2 * 3 + 4 + 2 + 16, 2 * 4 + 4 + 16 + 2 * 4
    wkv = {}
    for kind, bpe in (("fwd", FWD_B), ("bwd", BWD_B)):
        ms = [e0.elapsed_time(e1) for k_, e0, e1, _ in log if k_ == kind]
        if ms:
            elems = next(n for k_, _, _, n in log if k_ == kind)
            avg = sum(ms) / len(ms)
            wkv[kind] = {"avg_ms": round(avg, 4), "launches": len(ms), "bytes_per_elem": bpe, "achieved_GBps": round(elems * bpe / (avg * 1e-3) / 1e9, 1),
                         "frac_of_8TBps": round(elems * bpe / (avg * 1e-3) / 8e12, 4)}
    print(json.dumps({"config": "cfg4: VisualRWKV-6 %dL C%d + CLIP ViT-L/14-336" % (a.layers, C), "wkv6_in_step": wkv, "lm_params_B": round(n_par / 1e9, 2),
                      "micro_bsz": B, "seq_len": T, "tokens_per_s": round(B * T / dt), "ms_per_step": round(dt * 1e3, 1),
                      "losses": [round(x, 3) for x in losses], "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1), "grad_cp": a.grad_cp, "fused": bool(a.fused)}))


if __name__ == "__main__":
    main()
