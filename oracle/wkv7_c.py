"""ctypes wrapper of oracle/libwkv7_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libwkv7_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_DIR, "wkv7_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.run(["make", "-C", _DIR, "-B", "libwkv7_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def forward(w, q, k, v, z, a):
    """bf16 CPU tensors (B,T,H,64) -> y bf16, s f32 (B,H,T/16,64,64), sa f32."""
    B, T, H, N = w.shape
    assert N == 64 and all(x.dtype == torch.bfloat16 and x.is_contiguous() and x.device.type == "cpu" for x in (w, q, k, v, z, a))
    y = torch.empty_like(v)
    s = torch.empty(B, H, T // 16, N, N, dtype=torch.float32)
    sa = torch.empty(B, T, H, N, dtype=torch.float32)
    rc = load().wkv7_oracle_forward(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a), _p(y), _p(s), _p(sa))
    assert rc == 0, rc
    return y, s, sa


def backward(w, q, k, v, z, a, dy, s, sa):
    B, T, H, N = w.shape
    outs = [torch.empty_like(w) for _ in range(6)]
    rc = load().wkv7_oracle_backward(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a), _p(dy.contiguous()),
                                     _p(s), _p(sa), *[_p(o) for o in outs])
    assert rc == 0, rc
    return tuple(outs)   # dw, dq, dk, dv, dz, da
