"""Softmax attention entry point of the vision towers: o = softmax(q k^T / sqrt(D) + bias) v.

q, k, v: (B, L, H, D) (any strides on the last-but-one axes, as sliced from a fused qkv projection);
returns (B, L, H, D) contiguous.  On an MI355X with bf16/fp16 inputs and no bias the hand-written
MFMA flash kernel of libvisualrwkv_hip.so is used when it supports the head size; everything else
(fp32 CPU tests, SAM's decomposed relative-position bias) goes through torch's SDPA.
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
