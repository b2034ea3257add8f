"""Golden vectors of the SAM attention block with decomposed relative-position bias at the two window sizes the
SAM ViT-B tower runs (14x14 windowed blocks, 64x64 global blocks), produced by RUNNING the reference's own
`Attention` module (VisualRWKV-v7/v7.00/src/sam.py:245-305, add_decomposed_rel_pos 392-426) in this container.

    python tests/golden/make_golden_sam_attn.py        # needs /root/reference; writes tests/golden/sam_attn_ref.pt

Inputs and weights are rounded to bf16-representable values so that the bf16 GPU path sees exactly the same numbers.
The fixture holds data only (inputs, weights, outputs)."""
import os
import sys

import torch

REF = "/root/reference/VisualRWKV-v7/v7.00"


def bf16_exact(t):
    return t.bfloat16().float()


def main():
    sys.path.insert(0, REF)
    from src import sam as ref_sam
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, dim, heads, side, batch in [("win14", 128, 2, 14, 3), ("glob64", 64, 1, 64, 1)]:
        m = ref_sam.Attention(dim, num_heads=heads, qkv_bias=True, use_rel_pos=True, input_size=(side, side))
        with torch.no_grad():
            for n, p in m.named_parameters():
                scale = 0.3 if n.startswith("rel_pos") else (0.1 if p.dim() > 1 else 0.05)
                p.copy_(bf16_exact(torch.randn(p.shape, generator=g) * scale))
        x = bf16_exact(torch.randn(batch, side, side, dim, generator=g))
        with torch.no_grad():
            y = m(x)
        out[name] = {"dim": dim, "heads": heads, "side": side, "x": x.bfloat16(),
                     "state": {k: v.bfloat16() for k, v in m.state_dict().items()},
                     "y": y.half() if name == "glob64" else y}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sam_attn_ref.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
