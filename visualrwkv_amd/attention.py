"""Softmax attention entry point of the vision towers: o = softmax(q k^T / sqrt(D) + bias) v.

q, k, v: (B, L, H, D) (any strides on the last-but-one axes, as sliced from a fused qkv projection);
returns (B, L, H, D) contiguous.  On an MI355X with bf16/fp16 inputs and no bias the hand-written
MFMA flash kernel of libvisualrwkv_hip.so is used when it supports the head size; `attention_relpos` is the SAM
variant whose decomposed relative-position bias is computed inside the kernel.  Everything else (fp32 / CPU tests)
goes through torch's SDPA.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

_USE_HIP = True


def set_hip_attention(flag: bool) -> None:
    global _USE_HIP
    _USE_HIP = bool(flag)


def _hip_supported(q: torch.Tensor, bias) -> bool:
    if not (_USE_HIP and q.is_cuda and bias is None and q.dtype == torch.bfloat16):
        return False
    from . import hip_attention
    return hip_attention.supported(q.shape[-1])


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    if _hip_supported(q, bias):
        from . import hip_attention
        return hip_attention.flash_forward(q, k, v)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias)
    return o.transpose(1, 2).contiguous()


def _rel_resized(size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """(2*size-1, C) table, linearly interpolated when the checkpoint's differs (src/sam.py:371-381)."""
    want = 2 * size - 1
    if rel_pos.shape[0] != want:
        rel_pos = F.interpolate(rel_pos.t()[None].float(), size=want, mode="linear")[0].t().to(rel_pos.dtype)
    return rel_pos


def rel_table(size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """(size, size, C) table R[q, k] = rel_pos[q - k + size - 1]  (src/sam.py:359-389, equal q/k sizes)."""
    rel_pos = _rel_resized(size, rel_pos)
    idx = torch.arange(size, device=rel_pos.device)
    return rel_pos[(idx[:, None] - idx[None, :]) + (size - 1)]


def attention_relpos(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor,
                     size: tuple) -> torch.Tensor:
    """SAM attention: softmax(q k^T / sqrt(D) + rel_h[q, kh] + rel_w[q, kw]) v over a (Hh, Ww) window
    (src/sam.py:289-305 with add_decomposed_rel_pos, 392-426; the bias uses the unscaled q)."""
    Hh, Ww = size
    B, L, nh, hd = q.shape
    if _USE_HIP and q.is_cuda and q.dtype == torch.bfloat16:
        from . import hip_attention
        if hip_attention.relpos_supported(hd, Hh, Ww):
            return hip_attention.flash_forward_relpos(q, k, v, _rel_resized(Hh, rel_pos_h), _rel_resized(Ww, rel_pos_w), Hh)
    rq = q.reshape(B, Hh, Ww, nh, hd)
    rel_h = torch.einsum("bhwnc,hkc->bnhwk", rq, rel_table(Hh, rel_pos_h).to(q.dtype))
    rel_w = torch.einsum("bhwnc,wkc->bnhwk", rq, rel_table(Ww, rel_pos_w).to(q.dtype))
    bias = (rel_h[..., :, None] + rel_w[..., None, :]).reshape(B, nh, Hh * Ww, Hh * Ww)
    return attention(q, k, v, bias=bias)
