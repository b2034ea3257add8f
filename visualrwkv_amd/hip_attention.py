"""Binding of the MFMA flash-attention forwards of libvisualrwkv_hip.so (csrc/attention_kernels.h)."""
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


_RELPOS_SIDES = (14, 64)


def relpos_supported(head_dim: int, side_h: int, side_w: int) -> bool:
    return head_dim == 64 and side_h == side_w and side_h in _RELPOS_SIDES


def flash_forward_relpos(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel_h: torch.Tensor, rel_w: torch.Tensor,
                         side: int) -> torch.Tensor:
    """SAM attention over windows of side x side tokens with the decomposed relative-position bias computed inside the
    kernel (src/sam.py:289-305, 392-426).  q, k, v: (B, side*side, H, 64) bf16 views of one qkv projection;
    rel_h, rel_w: (2*side-1, 64) tables (any float dtype).  Returns (B, L, H, D) contiguous."""
    B, L, H, D = q.shape
    if L != side * side or not relpos_supported(D, side, side):
        raise ValueError(f"flash_forward_relpos: unsupported window {side} / head dim {D} / L {L}")
    if not (q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.is_cuda):
        raise ValueError("flash_forward_relpos needs bf16 CUDA tensors")
    if tuple(rel_h.shape) != (2 * side - 1, D) or tuple(rel_w.shape) != (2 * side - 1, D):
        raise ValueError("rel_h / rel_w must be (2*side-1, head_dim)")
    if q.stride() != k.stride() or q.stride() != v.stride() or q.stride(-1) != 1:
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    rel_h = rel_h.to(torch.bfloat16).contiguous()
    rel_w = rel_w.to(torch.bfloat16).contiguous()
    o = torch.empty(B, L, H, D, dtype=q.dtype, device=q.device)
    sb, sl, sh, _ = q.stride()
    rc = hip_lib.load().vrwkv_attention_relpos_fwd_bf16(B, side, H, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), sb, sl, sh,
                                                        rel_h.data_ptr(), rel_w.data_ptr(), o.data_ptr(),
                                                        hip_lib.launch_stream(q.device))
    hip_lib.check(rc, "vrwkv_attention_relpos_fwd_bf16")
    return o
