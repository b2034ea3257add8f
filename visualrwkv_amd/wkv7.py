"""The reference's native-operator surface for RWKV-7, backed by the gfx950 library.

Mirrors VisualRWKV-v7/v7.00/src/model.py:38-70 and cuda/wkv7_op.cpp:21-29:

* ``torch.ops.wind_backstepping.forward(w,q,k,v,z,a, y,s,sa)`` and ``.backward(...)`` with the
  reference's schemas (mutable out-arguments), registered for the CUDA dispatch key (which is the
  key of HIP tensors on PyTorch-ROCm).  Unlike the reference binding (raw data_ptr casts, legacy
  default stream, no checks) the implementation validates dtype/shape/contiguity/device and
  launches on PyTorch's *current* stream of the tensors' device, which is what makes overlapping
  the WKV backward with RCCL traffic on a side stream legal.
* ``WindBackstepping`` (autograd.Function), ``RUN_CUDA_RWKV7g``, ``CHUNK_LEN``, ``HEAD_SIZE`` with
  the reference's names, argument order and assertions.

The reference's own ``src/model.py`` works unchanged on top of this module if its import-time
``load(name="wind_backstepping", sources=[...cu])`` line is replaced by
``import visualrwkv_amd.wkv7``.
"""
from __future__ import annotations

import os

import torch

from . import hip_lib

HEAD_SIZE = 64
CHUNK_LEN = 16

_FWD_SCHEMA = ("forward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, "
               "Tensor(a!) y, Tensor(b!) s, Tensor(c!) sa) -> ()")
_BWD_SCHEMA = ("backward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, Tensor dy, "
               "Tensor s, Tensor sa, Tensor(a!) dw, Tensor(b!) dq, Tensor(c!) dk, Tensor(d!) dv, "
               "Tensor(e!) dz, Tensor(f!) da) -> ()")


# Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  When
# `EVENT_LOG` is a list, every op call appends (kind, start_event, end_event, B*T*H*64).
EVENT_LOG = None
# Sequence-parallel training op for few heads (B*H workgroups on 256 CUs).  Timed on MI355X (profiles/r2_tpar_micro.jsonl,
# H = 32): backward 1.52 -> 0.61 ms at B=1, T=6400 (8 segments), 0.61 -> 0.45 ms at B=1, T=2624 (4), 1.53 -> 1.09 / 0.63 -> 0.49 ms
# at B=2; no gain from B*H = 128 on.  The forward (three passes + a checkpoint re-order) only wins for one sequence of
# T >= 4096 (0.77 -> 0.48 ms).  VRWKV_TPAR_BWD=0 switches both off, =1 forces them wherever segments() > 1.
_TPAR_ENV = os.environ.get("VRWKV_TPAR_BWD", "auto")
TPARALLEL_BWD = _TPAR_ENV != "0"


def _timed(kind, elems, stream_dev, fn):
    if EVENT_LOG is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(stream_dev)
    e0.record(st)
    rc = fn()
    e1.record(st)
    EVENT_LOG.append((kind, e0, e1, elems))
    return rc


def _check_act(name, t, B, T, H):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"wind_backstepping: {name} must be bfloat16, got {t.dtype}")
    if tuple(t.shape) != (B, T, H, HEAD_SIZE):
        raise ValueError(f"wind_backstepping: {name} has shape {tuple(t.shape)}, expected {(B, T, H, HEAD_SIZE)}")
    if not t.is_contiguous():
        raise ValueError(f"wind_backstepping: {name} must be contiguous")


def _check_state(s, sa, B, T, H, dev):
    if s.dtype != torch.float32 or sa.dtype != torch.float32:
        raise TypeError("wind_backstepping: s and sa must be float32")
    if tuple(s.shape) != (B, H, T // CHUNK_LEN, HEAD_SIZE, HEAD_SIZE) or not s.is_contiguous():
        raise ValueError(f"wind_backstepping: s must be contiguous (B,H,T/{CHUNK_LEN},64,64), got {tuple(s.shape)}")
    if tuple(sa.shape) != (B, T, H, HEAD_SIZE) or not sa.is_contiguous():
        raise ValueError(f"wind_backstepping: sa must be contiguous (B,T,H,64), got {tuple(sa.shape)}")
    if s.device != dev or sa.device != dev:
        raise ValueError("wind_backstepping: all tensors must live on the same device")


def _dims(w):
    if w.dim() != 4 or w.shape[3] != HEAD_SIZE:
        raise ValueError(f"wind_backstepping: expected (B,T,H,{HEAD_SIZE}) activations, got {tuple(w.shape)}")
    B, T, H, _ = w.shape
    if T % CHUNK_LEN != 0:
        raise ValueError(f"wind_backstepping: T={T} must be a multiple of {CHUNK_LEN}")
    return B, T, H


def _forward_hip(w, q, k, v, z, a, y, s, sa):
    B, T, H = _dims(w)
    for n, t in zip("wqkvzay", (w, q, k, v, z, a, y)):
        _check_act(n, t, B, T, H)
        if t.device != w.device:
            raise ValueError("wind_backstepping: all tensors must live on the same device")
    _check_state(s, sa, B, T, H, w.device)
    lib = hip_lib.load()
    with torch.cuda.device(w.device):
        stream = torch.cuda.current_stream(w.device).cuda_stream
        rc = _timed("fwd", B * T * H * HEAD_SIZE, w.device, lambda: lib.vrwkv_wkv7_forward_bf16(
            B, T, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(), a.data_ptr(),
            y.data_ptr(), s.data_ptr(), sa.data_ptr(), stream))
    hip_lib.check(rc, "vrwkv_wkv7_forward_bf16")


def _backward_hip(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da):
    B, T, H = _dims(w)
    names = ("w", "q", "k", "v", "z", "a", "dy", "dw", "dq", "dk", "dv", "dz", "da")
    for n, t in zip(names, (w, q, k, v, z, a, dy, dw, dq, dk, dv, dz, da)):
        _check_act(n, t, B, T, H)
        if t.device != w.device:
            raise ValueError("wind_backstepping: all tensors must live on the same device")
    _check_state(s, sa, B, T, H, w.device)
    lib = hip_lib.load()
    with torch.cuda.device(w.device):
        stream = torch.cuda.current_stream(w.device).cuda_stream
        rc = _timed("bwd", B * T * H * HEAD_SIZE, w.device, lambda: lib.vrwkv_wkv7_backward_bf16(
            B, T, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(), a.data_ptr(),
            dy.data_ptr(), s.data_ptr(), sa.data_ptr(), dw.data_ptr(), dq.data_ptr(), dk.data_ptr(),
            dv.data_ptr(), dz.data_ptr(), da.data_ptr(), stream))
    hip_lib.check(rc, "vrwkv_wkv7_backward_bf16")


HOST_THREADS = int(os.environ.get("VRWKV_HOST_THREADS", "0"))     # 0 = every hardware thread


def _host_dtype(w, tensors, names):
    """CPU key: bf16 (the op's contract) or float32 (BASELINE config 1, RWKV_FLOAT_MODE=fp32) activations."""
    if w.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"wind_backstepping (CPU): activations must be bfloat16 or float32, got {w.dtype}")
    B, T, H = _dims(w)
    for n, t in zip(names, tensors):
        if t.dtype != w.dtype:
            raise TypeError(f"wind_backstepping (CPU): {n} is {t.dtype}, expected {w.dtype}")
        if tuple(t.shape) != (B, T, H, HEAD_SIZE) or not t.is_contiguous():
            raise ValueError(f"wind_backstepping (CPU): {n} must be contiguous {(B, T, H, HEAD_SIZE)}, got {tuple(t.shape)}")
        if t.device.type != "cpu":
            raise ValueError("wind_backstepping: all tensors must live on the same device")
    return B, T, H, 0 if w.dtype == torch.bfloat16 else 1


def _check_host(rc, what):
    if rc != 0:                 # the host library has no strerror of its own: the codes are those of include/visualrwkv_hip.h
        raise RuntimeError(f"{what} failed with code {rc} (VRWKV_E*: include/visualrwkv_hip.h)")


def _forward_host(w, q, k, v, z, a, y, s, sa):
    """`CPU` dispatch key (the reference has none, cuda/wkv7_op.cpp:26): csrc/wkv7_host.hip on the host cores."""
    B, T, H, code = _host_dtype(w, (w, q, k, v, z, a, y), "wqkvzay")
    _check_state(s, sa, B, T, H, w.device)
    rc = hip_lib.load_host().vrwkv_wkv7_forward_host(B, T, H, code, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                z.data_ptr(), a.data_ptr(), y.data_ptr(), s.data_ptr(), sa.data_ptr(), HOST_THREADS)
    _check_host(rc, "vrwkv_wkv7_forward_host")


def _backward_host(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da):
    names = ("w", "q", "k", "v", "z", "a", "dy", "dw", "dq", "dk", "dv", "dz", "da")
    B, T, H, code = _host_dtype(w, (w, q, k, v, z, a, dy, dw, dq, dk, dv, dz, da), names)
    _check_state(s, sa, B, T, H, w.device)
    rc = hip_lib.load_host().vrwkv_wkv7_backward_host(B, T, H, code, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(),
                                                 a.data_ptr(), dy.data_ptr(), s.data_ptr(), sa.data_ptr(), dw.data_ptr(), dq.data_ptr(),
                                                 dk.data_ptr(), dv.data_ptr(), dz.data_ptr(), da.data_ptr(), HOST_THREADS)
    _check_host(rc, "vrwkv_wkv7_backward_host")


def _no_cpu(*args):
    raise NotImplementedError("this WKV7 entry point (stateful step / prefill) has no CPU implementation; move the tensors to an "
                              "MI355X device.  The training op torch.ops.wind_backstepping does run on CPU tensors.")


def _apply_variant_env():
    """VRWKV_BWD_VARIANT / VRWKV_FWD_VARIANT: same-box A/B of kernel generations inside the training step (benchmarks only)."""
    for env, fn in (("VRWKV_BWD_VARIANT", "vrwkv_wkv7_set_backward_variant"), ("VRWKV_FWD_VARIANT", "vrwkv_wkv7_set_forward_variant")):
        v = os.environ.get(env)
        if v is not None:
            hip_lib.check(getattr(hip_lib.load(), fn)(int(v)), fn)


def _register():
    try:
        lib = torch.library.Library("wind_backstepping", "DEF")
    except RuntimeError as e:  # namespace already defined by another loader
        raise RuntimeError("torch.ops.wind_backstepping is already defined in this process; "
                           "do not load the reference's CUDA extension next to visualrwkv_amd") from e
    lib.define(_FWD_SCHEMA)
    lib.define(_BWD_SCHEMA)
    lib.impl("forward", _forward_hip, "CUDA")
    lib.impl("backward", _backward_hip, "CUDA")
    lib.impl("forward", _forward_host, "CPU")
    lib.impl("backward", _backward_host, "CPU")
    return lib


_LIB = _register()
_apply_variant_env()


# Selective activation recompute (fused.blocks_forward's memory mode 2): a training forward called with recompute_state=True keeps neither
# the chunk checkpoints `s` (16 B / element) nor `sa` (4 B / element) -- 10 of the ~40 activation tensors a 1.5B layer keeps -- and the
# backward re-runs the forward kernel to get them back (0.57 ms per layer at micro-batch 16).  The forward itself then runs the entry
# without by-products (2 instead of 24 output bytes per element).  The switch is an ARGUMENT of the call (WindBackstepping.apply(..., True),
# RUN_CUDA_RWKV7g(..., recompute_state=True)), not module state: forward and backward run on different threads.


class WindBackstepping(torch.autograd.Function):
    """src/model.py:45-65 (same asserts, same saved tensors, same allocation pattern)."""

    @staticmethod
    def forward(ctx, w, q, k, v, z, b, *extra):
        # extra: () as the reference calls it, or (recompute_state,) -- see the note above
        ctx.n_extra = len(extra)
        B, T, H, C = w.shape
        assert T % CHUNK_LEN == 0
        # bf16 as in the reference; float32 only for the CPU key (BASELINE config 1, RWKV_FLOAT_MODE=fp32)
        assert all(i.dtype == torch.bfloat16 or (i.dtype == torch.float32 and not i.is_cuda) for i in [w, q, k, v, z, b])
        assert all(i.is_contiguous() for i in [w, q, k, v, z, b])
        P = tparallel_segments(B, H, T, forward=True) if TPARALLEL_BWD and w.is_cuda else 1
        hip_bf16 = w.is_cuda and w.dtype == torch.bfloat16 and P == 1
        if hip_bf16 and not any(ctx.needs_input_grad[:6]):
            # nobody will ask for a gradient (evaluate.py / generate() under no_grad, frozen inputs): the by-products `s` and `sa` -- 22 of the
            # forward's 24 output bytes per element -- have no consumer, so the entry without them runs (same kernel, same y)
            ctx.recompute = False
            return wkv7_forward_state(w, q, k, v, z, b, None, want_state=False)[0]
        ctx.recompute = bool(extra and extra[0] and hip_bf16)
        if ctx.recompute:                               # y only; the backward regenerates s and sa
            y, _ = wkv7_forward_state(w, q, k, v, z, b, None, want_state=False)
            ctx.save_for_backward(w, q, k, v, z, b)
            return y
        if P > 1:                                       # one long sequence: sequence-parallel forward
            y, _, s, sa = wkv7_forward_tparallel(w, q, k, v, z, b, segments=P, train=True)
        else:
            y = torch.empty_like(v)
            s = torch.empty(B, H, T // CHUNK_LEN, C, C, dtype=torch.float32, device=w.device)
            sa = torch.empty(B, T, H, C, dtype=torch.float32, device=w.device)
            torch.ops.wind_backstepping.forward(w, q, k, v, z, b, y, s, sa)
        ctx.save_for_backward(w, q, k, v, z, b, s, sa)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.recompute:
            w, q, k, v, z, b = ctx.saved_tensors
            B, T, H, C = w.shape
            s = torch.empty(B, H, T // CHUNK_LEN, C, C, dtype=torch.float32, device=w.device)
            sa = torch.empty(B, T, H, C, dtype=torch.float32, device=w.device)
            torch.ops.wind_backstepping.forward(w, q, k, v, z, b, torch.empty_like(v), s, sa)
        else:
            w, q, k, v, z, b, s, sa = ctx.saved_tensors
        assert all(i.dtype == w.dtype for i in [dy])
        assert all(i.is_contiguous() for i in [dy])
        if TPARALLEL_BWD and w.is_cuda:                 # few heads -> sequence-parallel backward
            B, T, H, _ = w.shape
            P = tparallel_segments(B, H, T)
            if P > 1:
                return (*wkv7_backward_tparallel(w, q, k, v, z, b, dy, s, sa, P), *([None] * ctx.n_extra))
        dw, dq, dk, dv, dz, db = [torch.empty_like(x) for x in [w, q, k, v, z, b]]
        torch.ops.wind_backstepping.backward(w, q, k, v, z, b, dy, s, sa, dw, dq, dk, dv, dz, db)
        return (dw, dq, dk, dv, dz, db, *([None] * ctx.n_extra))


def RUN_CUDA_RWKV7g(q, w, k, v, a, b, recompute_state=False):
    """src/model.py:67-70: (B,T,HC) views -> (B,T,H,64); note the (w,q,...) argument re-order.  recompute_state (not in the reference):
    keep only the six inputs for the backward and re-run the forward kernel there for the by-products `s` and `sa`."""
    B, T, HC = q.shape
    q, w, k, v, a, b = [i.view(B, T, HC // 64, 64) for i in [q, w, k, v, a, b]]
    if recompute_state:
        return WindBackstepping.apply(w, q, k, v, a, b, True).view(B, T, HC)
    return WindBackstepping.apply(w, q, k, v, a, b).view(B, T, HC)


# ---------------------------------------------------------------------------------------------------------------
# Stateful generation.  Not in the reference, whose generate() re-runs the full forward for
# every new token (src/model.py:513-529); same recurrence, state carried between calls.
# ---------------------------------------------------------------------------------------------------------------
def wkv7_forward_state(w, q, k, v, z, a, state0=None, want_state=True, s_ckpt=None, sa=None):
    """Forward from an explicit state, without the training by-products: (B,T,H,64) bf16 inputs, T % 16 == 0,
    state0 (B,H,64,64) fp32 in [value row][key column] order or None (= zeros).  Returns (y, final state or None).
    The kernel neither writes the chunk checkpoints nor `sa` (22 of the training forward's 24 output bytes per element)
    unless the caller passes buffers for them (s_ckpt (B,H,T/16,64,64), sa (B,T,H,64), fp32)."""
    B, T, H = _dims(w)
    for n, t in zip("wqkvza", (w, q, k, v, z, a)):
        _check_act(n, t, B, T, H)
    if state0 is not None and (state0.dtype != torch.float32 or not state0.is_contiguous()
                               or tuple(state0.shape) != (B, H, HEAD_SIZE, HEAD_SIZE) or state0.device != w.device):
        raise ValueError(f"wkv7: state0 must be a contiguous fp32 ({B},{H},{HEAD_SIZE},{HEAD_SIZE}) tensor on {w.device}")
    if (s_ckpt is None) != (sa is None):
        raise ValueError("wkv7: s_ckpt and sa go together")
    if s_ckpt is not None:
        _check_state(s_ckpt, sa, B, T, H, w.device)
    y = torch.empty_like(v)
    s_fin = torch.empty(B, H, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=w.device) if want_state else None
    lib = hip_lib.load()
    with torch.cuda.device(w.device):
        stream = torch.cuda.current_stream(w.device).cuda_stream
        rc = _timed("fwd_state", B * T * H * HEAD_SIZE, w.device, lambda: lib.vrwkv_wkv7_forward_state_bf16(
            B, T, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(),
            a.data_ptr(), y.data_ptr(), state0.data_ptr() if state0 is not None else 0,
            s_fin.data_ptr() if want_state else 0,
            s_ckpt.data_ptr() if s_ckpt is not None else 0, sa.data_ptr() if sa is not None else 0, stream))
    hip_lib.check(rc, "vrwkv_wkv7_forward_state_bf16")
    return y, s_fin


def wkv7_prefill(w, q, k, v, z, a, state0=None):
    """(y, state after the last token) for a whole number of 16-token chunks, continuing from `state0` (None: empty)."""
    return wkv7_forward_state(w, q, k, v, z, a, state0, want_state=True)


def tparallel_segments(B, H, T, n_cu=256, max_segments=16, forward=False):
    """How many T-segments to cut a sequence into so that B*H*segments workgroups fill the chip:
    1 when the heads alone fill it or the sequence is too short to amortise the extra passes.  With the measured
    defaults (VRWKV_TPAR_BWD unset) the backward is cut when B*H <= 64 and T >= 1024, the forward when B*H <= 32 and
    T >= 4096; VRWKV_TPAR_BWD=1 cuts whenever at least 3 segments fit."""
    if B * H >= n_cu or T < 256:
        return 1
    if _TPAR_ENV == "auto" and (B * H > (32 if forward else 64) or T < (4096 if forward else 1024)):
        return 1
    p = min(max_segments, max(1, n_cu // (B * H)), T // 64)
    while p > 1 and (T % p != 0 or (T // p) % CHUNK_LEN != 0):
        p -= 1
    return p if p >= 3 else 1                       # three passes: fewer than 3 segments cannot win


def wkv7_forward_tparallel(w, q, k, v, z, a, state0=None, segments=None, train=False):
    """Sequence-parallel forward for few heads (inference prefill: B*H = 32 workgroups on 256 CUs).  The recurrence
    is linear in the state, S_end = S_start M_p + B_p per segment p, with M_p (64x64, acting on the key index) and
    B_p independent of S_start.  Three launches over all B*P*H (segment, head) pairs:
      1. B_p  : every segment from S = 0                       2. M_p : every segment from S = I with v = 0
      (then P tiny 64x64 products per head chain the segment-start states)
      3. y    : every segment from its true start state.
    Returns (y, final state); with train=True also the training op's by-products: (y, final state, s, sa)."""
    B, T, H = _dims(w)
    P = segments if segments is not None else tparallel_segments(B, H, T)
    if P <= 1:
        if not train:
            return wkv7_forward_state(w, q, k, v, z, a, state0)
        s = torch.empty(B, H, T // CHUNK_LEN, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=w.device)
        sa = torch.empty(B, T, H, HEAD_SIZE, dtype=torch.float32, device=w.device)
        y, fin = wkv7_forward_state(w, q, k, v, z, a, state0, s_ckpt=s, sa=sa)
        return y, fin, s, sa
    if T % P != 0 or (T // P) % CHUNK_LEN != 0:
        raise ValueError(f"wkv7: T = {T} cannot be cut into {P} segments of whole {CHUNK_LEN}-token chunks")
    Ts = T // P
    seg = [t.view(B * P, Ts, H, HEAD_SIZE) for t in (w, q, k, v, z, a)]          # contiguous views, no copies
    _, b_p = wkv7_forward_state(*seg)                                              # (B*P,H,64,64)
    eye = torch.eye(HEAD_SIZE, dtype=torch.float32, device=w.device).expand(B * P, H, HEAD_SIZE, HEAD_SIZE).contiguous()
    _, m_p = wkv7_forward_state(seg[0], seg[1], seg[2], torch.zeros_like(seg[3]), seg[4], seg[5], eye)
    b_p = b_p.view(B, P, H, HEAD_SIZE, HEAD_SIZE)
    m_p = m_p.view(B, P, H, HEAD_SIZE, HEAD_SIZE)
    cur = state0 if state0 is not None else torch.zeros(B, H, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=w.device)
    starts = []
    for p in range(P):
        starts.append(cur)
        cur = torch.matmul(cur, m_p[:, p]) + b_p[:, p]
    s0 = torch.stack(starts, dim=1).view(B * P, H, HEAD_SIZE, HEAD_SIZE).contiguous()
    if not train:
        y, _ = wkv7_forward_state(*seg, s0, want_state=False)
        return y.view(B, T, H, HEAD_SIZE), cur
    # training by-products: `sa` is token-major, so the segment view writes it in place; the chunk checkpoints come out
    # as (B*P, H, Ts/16, ..) and are re-ordered to the op's (B, H, T/16, ..) with one copy
    nc = Ts // CHUNK_LEN
    s_seg = torch.empty(B * P, H, nc, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=w.device)
    sa = torch.empty(B, T, H, HEAD_SIZE, dtype=torch.float32, device=w.device)
    y, _ = wkv7_forward_state(*seg, s0, want_state=False, s_ckpt=s_seg, sa=sa.view(B * P, Ts, H, HEAD_SIZE))
    s = s_seg.view(B, P, H, nc, HEAD_SIZE, HEAD_SIZE).transpose(1, 2).reshape(B, H, P * nc, HEAD_SIZE, HEAD_SIZE)
    return y.view(B, T, H, HEAD_SIZE), cur, s, sa


def wkv7_backward_tparallel(w, q, k, v, z, a, dy, s, sa, segments):
    """Sequence-parallel backward for few heads (the training op's arguments plus a segment count).
    dL/dS obeys dS_start = dS_end M_p^T + C_p over segment p, with M_p the forward map of the segment
    (S_end = S_start M_p + ..).  Launches over all B*H*P (head, segment) pairs:
      1. M_p   : forward of every segment from S = I with v = 0 (as in wkv7_forward_tparallel)
      2. C_p   : backward of every segment from dS_end = 0 (its gradients are discarded)
      (then P - 1 products of 64x64 matrices per head chain the segment-end dS)
      3. grads : backward of every segment from its true dS_end.
    ~2.5x the work of the sequential backward on P times the workgroups.  Returns (dw, dq, dk, dv, dz, da)."""
    B, T, H = _dims(w)
    P = int(segments)
    if P < 1 or T % P != 0 or (T // P) % CHUNK_LEN != 0:
        raise ValueError(f"wkv7: T = {T} cannot be cut into {P} segments of whole {CHUNK_LEN}-token chunks")
    for n, t in zip(("w", "q", "k", "v", "z", "a", "dy"), (w, q, k, v, z, a, dy)):
        _check_act(n, t, B, T, H)
    _check_state(s, sa, B, T, H, w.device)
    grads = [torch.empty_like(w) for _ in range(6)]
    lib = hip_lib.load()

    def launch(ds_in, ds_out):
        with torch.cuda.device(w.device):
            rc = lib.vrwkv_wkv7_backward_segments_bf16(
                B, T, H, P, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(), a.data_ptr(), dy.data_ptr(),
                s.data_ptr(), sa.data_ptr(), ds_in.data_ptr() if ds_in is not None else 0,
                ds_out.data_ptr() if ds_out is not None else 0, *[g.data_ptr() for g in grads],
                torch.cuda.current_stream(w.device).cuda_stream)
        hip_lib.check(rc, "vrwkv_wkv7_backward_segments_bf16")

    if P == 1:
        launch(None, None)
        return tuple(grads)
    Ts = T // P
    seg = [t.view(B * P, Ts, H, HEAD_SIZE) for t in (w, q, k, v, z, a)]
    eye = torch.eye(HEAD_SIZE, dtype=torch.float32, device=w.device).expand(B * P, H, HEAD_SIZE, HEAD_SIZE).contiguous()
    _, m_p = wkv7_forward_state(seg[0], seg[1], seg[2], torch.zeros_like(seg[3]), seg[4], seg[5], eye)
    m_p = m_p.view(B, P, H, HEAD_SIZE, HEAD_SIZE)
    c_p = torch.empty(B, H, P, HEAD_SIZE, HEAD_SIZE, dtype=torch.float32, device=w.device)
    launch(None, c_p)
    ds_end = torch.zeros_like(c_p)
    for p in range(P - 1, 0, -1):
        ds_end[:, :, p - 1] = torch.matmul(ds_end[:, :, p], m_p[:, p].transpose(-1, -2)) + c_p[:, :, p]
    launch(ds_end, None)
    return tuple(grads)


def wkv7_step(w, q, k, v, z, a, state):
    """One token: (B,H,64) bf16 each, `state` (B,H,64,64) fp32 updated in place; returns y (B,H,64) bf16."""
    B, H, C = w.shape
    if not w.is_cuda:
        _no_cpu()
    if C != HEAD_SIZE:
        raise ValueError(f"head size {C} != {HEAD_SIZE}")
    for name, t in zip("wqkvza", (w, q, k, v, z, a)):
        if t.dtype != torch.bfloat16 or not t.is_contiguous() or tuple(t.shape) != (B, H, C) or t.device != w.device:
            raise ValueError(f"wkv7_step: {name} must be a contiguous bf16 ({B},{H},{C}) tensor on {w.device}")
    if (state.dtype != torch.float32 or not state.is_contiguous() or tuple(state.shape) != (B, H, C, C)
            or state.device != w.device):
        raise ValueError(f"wkv7_step: state must be a contiguous fp32 ({B},{H},{C},{C}) tensor on {w.device}")
    y = torch.empty_like(v)
    lib = hip_lib.load()
    with torch.cuda.device(w.device):
        rc = lib.vrwkv_wkv7_step_bf16(B, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(),
                                      a.data_ptr(), state.data_ptr(), y.data_ptr(),
                                      torch.cuda.current_stream(w.device).cuda_stream)
    hip_lib.check(rc, "vrwkv_wkv7_step_bf16")
    return y
