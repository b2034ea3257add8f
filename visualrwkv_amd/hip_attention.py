"""Binding of the MFMA flash-attention forward of libvisualrwkv_hip.so (csrc/attention_kernels.h)."""
from __future__ import annotations

import torch

from . import hip_lib

_HEAD_DIMS = (64, 72)


def supported(head_dim: int) -> bool:
    return head_dim in _HEAD_DIMS


def flash_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """q, k, v: (B, L, H, D) bf16 views with identical strides and a contiguous last dim (slices of one fused
    qkv projection); returns (B, L, H, D) contiguous."""
    B, L, H, D = q.shape
    if not (q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.is_cuda):
        raise ValueError("flash_forward needs bf16 CUDA tensors")
    if q.stride() != k.stride() or q.stride() != v.stride() or q.stride(-1) != 1:
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    o = torch.empty(B, L, H, D, dtype=q.dtype, device=q.device)
    sb, sl, sh, _ = q.stride()
    rc = hip_lib.load().vrwkv_attention_fwd_bf16(B, L, H, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), sb, sl, sh,
                                                 o.data_ptr(), hip_lib.launch_stream(q.device))
    hip_lib.check(rc, "vrwkv_attention_fwd_bf16")
    return o
