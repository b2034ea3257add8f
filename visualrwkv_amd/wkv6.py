"""WKV6 operator surface of the reference (BASELINE config 4), backed by the HIP kernels of csrc/wkv6_chunked.h.

Mirrors VisualRWKV-v6/v6.0/src/model.py:38-88: the extension module's `forward` / `backward` entry points (here
`torch.ops.wkv6.forward / backward`, the schemas of cuda/wkv6_op.cpp:8-13,21-24), the autograd function `WKV_6`
(same asserts, same `ew = -exp(w.float())`, same `gu` reduction over the batch) and `RUN_CUDA_RWKV6`.
Differences a caller cannot see: the forward also writes chunk-start state checkpoints that `WKV_6` keeps for the
backward (the reference re-sweeps the sequence five times instead), and T has no compile-time bound (the reference
sizes a per-thread array with -D_T_=ctx_len).  CPU tensors raise: there is no CPU implementation, as in the reference.
"""
from __future__ import annotations

import os

import torch

from . import hip_lib

HEAD_SIZE = 64

_FWD_SCHEMA = ("forward(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, "
               "Tensor(a!) y) -> ()")
_BWD_SCHEMA = ("backward(int B, int T, int C, int H, Tensor r, Tensor k, Tensor v, Tensor w, Tensor u, Tensor gy, "
               "Tensor(a!) gr, Tensor(b!) gk, Tensor(c!) gv, Tensor(d!) gw, Tensor(e!) gu) -> ()")


def _chk(name, t, shape, dtype):
    if not t.is_cuda:
        raise NotImplementedError("wkv6 has no CPU implementation (neither does the reference: cuda/wkv6_op.cpp "
                                  "binds CUDA kernels only). Move the tensors to an MI355X device.")
    if t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
        raise ValueError(f"wkv6: {name} must be a contiguous {dtype} tensor of shape {tuple(shape)}, "
                         f"got {t.dtype} {tuple(t.shape)} contiguous={t.is_contiguous()}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def ckpt_tensor(B, T, H, device):
    n = hip_lib.load().vrwkv_wkv6_ckpt_floats(B, T, H)
    return torch.empty(n, dtype=torch.float32, device=device)


# Optional per-launch timing with HIP events on the launch stream (benchmarks/bench_v6.py's roofline leg), as wkv7.EVENT_LOG: when it is a
# list, every launch appends (kind, start_event, end_event, B*T*C).
EVENT_LOG = None


def _timed(kind, elems, dev, fn):
    if EVENT_LOG is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(dev)
    e0.record(st)
    rc = fn()
    e1.record(st)
    EVENT_LOG.append((kind, e0, e1, elems))
    return rc


def forward_hip(B, T, C, H, r, k, v, ew, u, y, ckpt=None):
    """ew is the f32 log decay the reference's kernels take (`w` of wkv6_op.cpp:8)."""
    bf = torch.bfloat16
    for n, t in (("r", r), ("k", k), ("v", v), ("y", y)):
        _chk(n, t, (B, T, C), bf)
    _chk("w", ew, (B, T, C), torch.float32)
    if u.numel() != C:
        raise ValueError(f"wkv6: u must have {C} elements")
    _chk("u", u, u.shape, bf)
    with torch.cuda.device(r.device):
        rc = _timed("fwd", B * T * C, r.device, lambda: hip_lib.load().vrwkv_wkv6_forward_bf16(
            B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), ew.data_ptr(), u.data_ptr(), y.data_ptr(), ckpt.data_ptr() if ckpt is not None else 0, _stream(r)))
    hip_lib.check(rc, "vrwkv_wkv6_forward_bf16")


def backward_hip(B, T, C, H, r, k, v, ew, u, gy, gr, gk, gv, gw, gu, ckpt=None):
    bf = torch.bfloat16
    for n, t in (("r", r), ("k", k), ("v", v), ("gy", gy), ("gr", gr), ("gk", gk), ("gv", gv), ("gw", gw)):
        _chk(n, t, (B, T, C), bf)
    _chk("w", ew, (B, T, C), torch.float32)
    _chk("u", u, u.shape, bf)
    _chk("gu", gu, (B, C), bf)
    if ckpt is None:            # reference calling convention: no saved state -> regenerate the checkpoints
        ckpt = ckpt_tensor(B, T, H, r.device)
        forward_hip(B, T, C, H, r, k, v, ew, u, torch.empty_like(r), ckpt)
    with torch.cuda.device(r.device):
        rc = _timed("bwd", B * T * C, r.device, lambda: hip_lib.load().vrwkv_wkv6_backward_bf16(
            B, T, C, H, r.data_ptr(), k.data_ptr(), v.data_ptr(), ew.data_ptr(), u.data_ptr(), gy.data_ptr(), ckpt.data_ptr(), gr.data_ptr(),
            gk.data_ptr(), gv.data_ptr(), gw.data_ptr(), gu.data_ptr(), _stream(r)))
    hip_lib.check(rc, "vrwkv_wkv6_backward_bf16")


def _apply_variant_env():
    """VRWKV_WKV6_BWD_VARIANT: same-box A/B of the backward kernel generations (benchmarks only)."""
    v = os.environ.get("VRWKV_WKV6_BWD_VARIANT")
    if v is not None:
        hip_lib.check(hip_lib.load().vrwkv_wkv6_set_backward_variant(int(v)), "vrwkv_wkv6_set_backward_variant")


def _register():
    lib = torch.library.Library("wkv6", "DEF")
    lib.define(_FWD_SCHEMA)
    lib.define(_BWD_SCHEMA)
    lib.impl("forward", lambda B, T, C, H, r, k, v, w, u, y: forward_hip(B, T, C, H, r, k, v, w, u, y), "CUDA")
    lib.impl("backward", lambda B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu:
             backward_hip(B, T, C, H, r, k, v, w, u, gy, gr, gk, gv, gw, gu), "CUDA")

    def no_cpu(*a):
        raise NotImplementedError("wkv6 has no CPU implementation; move the tensors to an MI355X device")
    lib.impl("forward", no_cpu, "CPU")
    lib.impl("backward", no_cpu, "CPU")
    return lib


_LIB = _register()
_apply_variant_env()


class WKV_6(torch.autograd.Function):
    """VisualRWKV-v6/v6.0/src/model.py:43-85."""

    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u):
        with torch.no_grad():
            assert all(x.dtype == torch.bfloat16 for x in (r, k, v, w, u))
            assert HEAD_SIZE == C // H
            assert all(x.is_contiguous() for x in (r, k, v, w, u))
            ctx.B, ctx.T, ctx.C, ctx.H = B, T, C, H
            ew = (-torch.exp(w.float())).contiguous()
            y = torch.empty((B, T, C), device=r.device, dtype=torch.bfloat16)
            ckpt = ckpt_tensor(B, T, H, r.device) if any(ctx.needs_input_grad) else None
            forward_hip(B, T, C, H, r, k, v, ew, u, y, ckpt)
            ctx.save_for_backward(r, k, v, ew, u, ckpt)
            return y

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            B, T, C, H = ctx.B, ctx.T, ctx.C, ctx.H
            gy = gy.contiguous()
            r, k, v, ew, u, ckpt = ctx.saved_tensors
            gr, gk, gv, gw = (torch.empty((B, T, C), device=gy.device, dtype=torch.bfloat16) for _ in range(4))
            gu = torch.empty((B, C), device=gy.device, dtype=torch.bfloat16)
            backward_hip(B, T, C, H, r, k, v, ew, u, gy, gr, gk, gv, gw, gu, ckpt)
            gu = torch.sum(gu, 0).view(H, C // H)
            return (None, None, None, None, gr, gk, gv, gw, gu)


def RUN_CUDA_RWKV6(B, T, C, H, r, k, v, w, u):
    return WKV_6.apply(B, T, C, H, r, k, v, w, u)
