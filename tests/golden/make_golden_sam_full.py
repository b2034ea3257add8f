"""Fixture for the SAM ViT-B tower at its REAL size (cfg 5: 1024 x 1024 input, 64 x 64 grid, 14 x 14 windows, four global
blocks over 4096 tokens, 12 blocks, the neck and the 2x2 space-to-depth), computed by the reference's own module
(VisualRWKV-v7/v7.00/src/sam.py: _build_sam's ImageEncoderViT arguments, :473-497) in fp32 on the CPU.

Neither the 90 M weights nor the (1, 1024, 32, 32) output are stored: weights and input are functions of their names /
a seed (tests/golden/det_weights.py, used by the GPU test as well), and the fixture keeps 16 384 sampled output values
with their flat indices plus summary statistics.   Run where /root/reference exists (about a minute on 8 cores):
    python tests/golden/make_golden_sam_full.py
"""
import os
import sys
from functools import partial

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/VisualRWKV-v7/v7.00"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

from det_weights import det_image, det_state  # noqa: E402


def main():
    from src import sam as ref_sam            # the reference file itself (pure torch: imports as it is)
    enc = ref_sam.ImageEncoderViT(depth=12, embed_dim=768, img_size=1024, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                                  num_heads=12, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=[2, 5, 8, 11],
                                  window_size=14, out_chans=256)
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(det_state(shapes), strict=True)
    enc.eval()
    x = det_image((1, 3, 1024, 1024))
    with torch.no_grad():
        y = enc(x)
    assert tuple(y.shape) == (1, 1024, 32, 32), y.shape
    flat = y.reshape(-1)
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(3))[:16384]
    out = {"provenance": "VisualRWKV-v7/v7.00/src/sam.py ImageEncoderViT with the arguments of _build_sam (:473-497) for ViT-B, fp32, CPU; "
                         "weights / input = tests/golden/det_weights.py (seed 2026 / 7)",
           "param_shapes": shapes, "out_shape": tuple(y.shape), "index": idx.clone(), "values": flat[idx].clone(),
           "out_rms": float(flat.double().pow(2).mean().sqrt()), "out_absmax": float(flat.abs().max())}
    torch.save(out, os.path.join(HERE, "sam_full_ref.pt"))
    print("wrote sam_full_ref.pt: out rms", out["out_rms"], "absmax", out["out_absmax"])


if __name__ == "__main__":
    main()
