"""MFMA flash-attention forward (gfx950) binding.  Filled in by csrc/attention.hip."""
from __future__ import annotations

import torch


def supported(head_dim: int) -> bool:
    from . import hip_lib
    lib = hip_lib.load()
    return hasattr(lib, "vrwkv_attention_fwd_bf16") and head_dim in (64, 72)


def flash_forward(q, k, v):
    raise NotImplementedError
