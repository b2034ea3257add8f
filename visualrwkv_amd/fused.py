"""Fused HIP implementations of the element-wise glue of RWKV_Tmix_x070 / RWKV_CMix_x070.

Enabled per model with `args.fused = True` (rwkv7.py dispatches here for CUDA tensors).  Each autograd
Function below is one forward kernel and one backward kernel of csrc/tmix_fused.hip; the dense projections
stay hipBLASLt GEMMs (plain `F.linear` / `@`).  There is no fallback: tensors must be bf16, contiguous, on an
MI355X -- anything else raises.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

from . import gemm_tuning, hip_lib
from .wkv7 import RUN_CUDA_RWKV7g


def _stream(t):
    return hip_lib.launch_stream(t.device)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
            raise ValueError("fused RWKV-7 kernels need contiguous bf16 tensors on the GPU "
                             f"(got {t.dtype}, cuda={t.is_cuda}, contiguous={t.is_contiguous()})")


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _p(t):
    return t.data_ptr() if t is not None else 0


def _ws(ntok, C, nvec, device):
    n = hip_lib.load().vrwkv_param_grad_ws_floats(ntok, C, nvec)
    return torch.empty(n, dtype=torch.float32, device=device)


def wgrad_skinny_supported(x2d, dy2d):
    K, N = x2d.shape[1], dy2d.shape[1]
    nw, d = max(K, N), min(K, N)
    return (x2d.is_cuda and x2d.dtype == torch.bfloat16 and dy2d.dtype == torch.bfloat16 and nw % 128 == 0
            and d in (32, 64, 96, 128, 160, 256) and nw > d)


def wgrad_skinny(x2d, dy2d):
    """x2d^T dy2d for (M,K), (M,N) with one of K, N a LoRA rank: csrc/lora_wgrad.h (the library's kernels for this
    shape -- reduction over ~42 k tokens, one operand 64-256 columns wide -- run 3-4x above the cost of reading the
    wide operand once).  Returns (K,N) bf16."""
    x2d, dy2d = x2d.contiguous(), dy2d.contiguous()
    M, K = x2d.shape
    N = dy2d.shape[1]
    wide, narrow, transposed = (x2d, dy2d, 0) if K >= N else (dy2d, x2d, 1)
    lib = hip_lib.load()
    nws = lib.vrwkv_wgrad_skinny_ws_floats(M, wide.shape[1], narrow.shape[1])
    if nws < 0:
        raise ValueError(f"wgrad_skinny: unsupported shape ({M},{K}) x ({M},{N})")
    ws = torch.empty(nws, dtype=torch.float32, device=x2d.device)
    out = torch.empty(K, N, dtype=torch.bfloat16, device=x2d.device)
    rc = lib.vrwkv_wgrad_skinny_bf16(M, wide.shape[1], narrow.shape[1], wide.data_ptr(), narrow.data_ptr(), out.data_ptr(),
                                     transposed, ws.data_ptr(), _stream(x2d))
    hip_lib.check(rc, "vrwkv_wgrad_skinny_bf16")
    return out


class _LoraMM(torch.autograd.Function):
    """x @ w for a LoRA factor w (C x rank or rank x C): library GEMMs for the output and the input gradient, the
    skinny weight-gradient kernel for dw (src/model.py:176,181-183 -- plain `@` in the reference)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dy @ w.t() if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            x2d, dy2d = x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1])
            dw = wgrad_skinny(x2d, dy2d) if wgrad_skinny_supported(x2d, dy2d) else x2d.t() @ dy2d
        return dx, dw


lora_mm = _LoraMM.apply
LORA_WGRAD = os.environ.get("VRWKV_LORA_WGRAD", "1") != "0"      # A/B switch for benchmarks: 0 = autograd's torch.mm
GRAD_ALIAS = os.environ.get("VRWKV_GRAD_ALIAS", "1") != "0"      # A/B switch: 0 = autograd sums the gradients of x_v, k2, v2
DGRAD_TN = os.environ.get("VRWKV_DGRAD_TN", "1") != "0"          # A/B switch: 0 = autograd's dy.mm(W) for the input gradient of Linear
VF_CHAIN = os.environ.get("VRWKV_VF_CHAIN", "1") != "0"          # A/B switch: 0 = every layer returns its own v_first gradient term, autograd adds them
FLAT_WGRAD = os.environ.get("VRWKV_FLAT_WGRAD", "1") != "0"      # A/B switch: 0 = weight gradients as fresh tensors, copied into the ZeRO-1 buffer


# dgrad (library GEMM) and wgrad (csrc/wgrad_big.h) of one Linear on two HIP streams, joined before the node returns: both are whole-chip kernels whose LAST
# round of workgroups is part-filled (C x C at 41 984 rows: 5.125 rounds of 256 x 256 tiles cost 6, profiles/r6h_gemm_tail_probe.jsonl), and a kernel of another
# stream takes the idle CUs.  The streams never leave the autograd node, so what autograd and the ZeRO-1 hooks see is unchanged.  0 = one after the other.
# (The same for the skinny LoRA products measured +0.7 % -- two 50 us memory-bound kernels gain less than the two stream joins cost -- and is not done.)
# SAFETY: the side stream carries only this package's own kernels (wgrad_big, relu^2), never a second LIBRARY GEMM: hipBLASLt's default pick for large
# shapes is a stream-K kernel whose workgroups spin on each other, and two of those on two streams deadlock the GPU (gemm_tuning's docstring).  Where both
# sides would be library GEMMs (the r/k/v forward, a weight gradient the library computes) the streams are used only for shapes listed as checked in the
# loaded tuning file's sidecar (gemm_tuning.concurrent_ok); everything else runs one after the other.
OVERLAP_WGRAD = os.environ.get("VRWKV_OVERLAP_WGRAD", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(dev, i=0):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), i)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


BIG_WGRAD = os.environ.get("VRWKV_BIG_WGRAD", "1") != "0"        # A/B switch: 0 = the library's "N,T" kernel for dW = dy^T x of the Linear layers


def wgrad_big_supported(dy2d, x2d):
    return (BIG_WGRAD and dy2d.is_cuda and dy2d.dtype == torch.bfloat16 and x2d.dtype == torch.bfloat16 and dy2d.is_contiguous()
            and x2d.is_contiguous() and dy2d.shape[0] % 32 == 0 and dy2d.shape[1] % 256 == 0 and x2d.shape[1] % 256 == 0
            and max(dy2d.shape[1], x2d.shape[1]) <= 16384)      # the head (65 536 x 2048): the library is 4 % faster (profiles/r4_wgrad_big_micro.jsonl)


def wgrad_big(dy2d, x2d, out=None):
    """dy2d^T x2d for (M,N), (M,K) -> (N,K) bf16 (the weight gradient of nn.Linear, src/model.py:150-153,214-215,281) with
    csrc/wgrad_big.h instead of the library's N,T-class kernel; `out`: a contiguous (N,K) bf16 view to write into."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    lib = hip_lib.load()
    nws = lib.vrwkv_wgrad_big_ws_floats(M, N, K)
    if nws < 0:
        raise ValueError(f"wgrad_big: unsupported shape ({M},{N}) x ({M},{K})")
    ws = torch.empty(nws, dtype=torch.float32, device=dy2d.device) if nws else None
    if out is None:
        out = torch.empty(N, K, dtype=torch.bfloat16, device=dy2d.device)
    rc = lib.vrwkv_wgrad_big_bf16(M, N, K, dy2d.data_ptr(), x2d.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else 0, _stream(dy2d))
    hip_lib.check(rc, "vrwkv_wgrad_big_bf16")
    return out


def _wgrad_beside_dgrad(wp, dy2, x2):
    """May dW = dy2^T x2 run on the side stream while the library computes the input gradient?  Yes when it is csrc/wgrad_big.h (the same conditions as
    in _weight_grad), or when this shape's pair of library kernels is listed as checked."""
    if not (OVERLAP_WGRAD and dy2.is_cuda):
        return False
    flat = (FLAT_WGRAD and wp is not None and wp.grad is None and getattr(wp, "_vrwkv_flat_armed", False)
            and not getattr(wp, "_vrwkv_wgrad_pending", False) and wp._vrwkv_flat_grad[0].dtype == dy2.dtype)
    if wgrad_big_supported(dy2, x2) and (not flat or wp._vrwkv_flat_grad[1] % 8 == 0):
        return True
    return gemm_tuning.concurrent_ok(f"dgrad+wgrad {dy2.shape[0]}x{dy2.shape[1]}x{x2.shape[1]}")


def _weight_grad(wp, dy2, x2):
    """dW = dy2^T x2 (N,K) for a Linear weight; `wp` = the Parameter when a ZeRO-1 engine (dp.Zero1Engine) armed by its zero_grad() owns
    its gradient slot, else None.  First gradient of an armed weight in the step: the GEMM writes into the weight's slot of the flat
    gradient buffer; autograd adopts the returned view as `.grad` and the engine finds it in place.  `pending` until the engine's hook has
    seen it: a second use of the same weight in one graph (two forward passes under one backward) must not write the slot again while
    autograd still holds the first gradient there."""
    if (FLAT_WGRAD and wp is not None and wp.grad is None and getattr(wp, "_vrwkv_flat_armed", False)
            and not getattr(wp, "_vrwkv_wgrad_pending", False) and wp._vrwkv_flat_grad[0].dtype == dy2.dtype):
        flat, o = wp._vrwkv_flat_grad
        dw = flat[o:o + wp.numel()].view(wp.shape)
        if wgrad_big_supported(dy2, x2) and o % 8 == 0:
            wgrad_big(dy2, x2, out=dw)
        else:
            torch.mm(dy2.t(), x2, out=dw)
        wp._vrwkv_wgrad_pending = True
        return dw
    if wgrad_big_supported(dy2, x2):
        return wgrad_big(dy2, x2)
    return dy2.t().mm(x2)


class _LinearTN(torch.autograd.Function):
    """F.linear(x, W) whose input gradient is issued in the layout of the forward GEMMs.  Autograd computes dx = dy.mm(W)
    with W (N_out, K_in) row-major: the contraction index is the strided one of W (hipBLASLt "N,N"), 8-15 % slower on
    MI355X than the "T,N" kernels the forward gets (both operands contraction-contiguous; measured 908 vs 1229 TFLOP/s at
    41 984 x 2048 x 2048, benchmarks/dgrad_layout_micro.py).  Here dx = F.linear(dy, W^T) on a transposed copy of the
    weight (31-77 us per weight, included in the measurement): the same T,N kernels as the forward.  dW: _weight_grad."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.wparam = w if hasattr(w, "_vrwkv_flat_grad") else None      # the Parameter itself (saved_tensors hands back a plain tensor)
        return F.linear(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and dy.is_cuda:
            dy = dy.contiguous()
        if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and _wgrad_beside_dgrad(ctx.wparam, dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]))):
            cur, side = torch.cuda.current_stream(dy.device), _side_stream(dy.device)
            side.wait_stream(cur)                                    # dy and x are ready
            with torch.cuda.stream(side):
                dw = _weight_grad(ctx.wparam, dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]))
            dx = F.linear(dy, transpose2d(w))
            cur.wait_stream(side)                                    # joined: nothing of the side stream outlives this node
            dw.record_stream(cur)
            return dx, dw
        if ctx.needs_input_grad[0]:
            dx = F.linear(dy, transpose2d(w))
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(ctx.wparam, dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]))
        return dx, dw


class _Linear3TN(torch.autograd.Function):
    """Three independent bias-free Linear layers on three inputs -- receptance / key / value of the time-mix (src/model.py:175-178) -- as ONE autograd node
    whose GEMMs run on three HIP streams (forward) and whose input-gradient and weight-gradient GEMMs run on two (backward), joined before the node
    returns (see OVERLAP_WGRAD): each of these C x C GEMMs alone leaves 7/8 of its last round of tiles idle."""

    @staticmethod
    def forward(ctx, x0, x1, x2, w0, w1, w2):
        ctx.save_for_backward(x0, x1, x2, w0, w1, w2)
        ctx.wparams = [w if hasattr(w, "_vrwkv_flat_grad") else None for w in (w0, w1, w2)]
        M = x0.numel() // x0.shape[-1]
        if not gemm_tuning.concurrent_ok(f"3x tn_{w0.shape[0]}_{M}_{w0.shape[1]}"):       # three library GEMMs at once: checked shapes only (see OVERLAP_WGRAD)
            return F.linear(x0, w0), F.linear(x1, w1), F.linear(x2, w2)
        cur = torch.cuda.current_stream(x0.device)
        sides = [_side_stream(x0.device, 0), _side_stream(x0.device, 1)]
        outs = [None, None, None]
        for st, i, x, w in ((sides[0], 1, x1, w1), (sides[1], 2, x2, w2)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs[i] = F.linear(x, w)
        outs[0] = F.linear(x0, w0)
        for st, i in ((sides[0], 1), (sides[1], 2)):
            cur.wait_stream(st)
            outs[i].record_stream(cur)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d0, d1, d2):
        x0, x1, x2, w0, w1, w2 = ctx.saved_tensors
        dys = [d.contiguous() for d in (d0, d1, d2)]
        jobs = [(wp, dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])) for wp, dy, x in zip(ctx.wparams, dys, (x0, x1, x2))]
        beside = all(_wgrad_beside_dgrad(*j) for j in jobs)
        cur = torch.cuda.current_stream(x0.device)
        side = _side_stream(x0.device) if beside else cur
        if beside:
            side.wait_stream(cur)
        with torch.cuda.stream(side):
            dws = [_weight_grad(*j) for j in jobs]
        dxs = [F.linear(dy, transpose2d(w)) for dy, w in zip(dys, (w0, w1, w2))]
        if beside:
            cur.wait_stream(side)
            for dw in dws:
                dw.record_stream(cur)
        return (*dxs, *dws)


def linear3(modules, xs):
    """(m(x) for m, x in zip(modules, xs)) for three bias-free nn.Linear of equal shape; one node with internal stream concurrency in training on the GPU."""
    if (OVERLAP_WGRAD and DGRAD_TN and all(m.bias is None for m in modules) and all(x.is_cuda and x.requires_grad for x in xs) and torch.is_grad_enabled()
            and all(m.weight.requires_grad for m in modules)):
        return _Linear3TN.apply(*xs, *[m.weight for m in modules])
    return tuple(linear(m, x) for m, x in zip(modules, xs))


def transpose2d(w):
    """w.t().contiguous() for a 2-D bf16 matrix: tiled HIP kernel when both sides are multiples of 64 (3-4x torch's copy)."""
    if w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous() and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
        out = torch.empty(w.shape[1], w.shape[0], dtype=w.dtype, device=w.device)
        hip_lib.check(hip_lib.load().vrwkv_transpose_bf16(w.shape[0], w.shape[1], w.data_ptr(), out.data_ptr(), _stream(w)),
                      "vrwkv_transpose_bf16")
        return out
    return w.t().contiguous()


def linear(module, x):
    """module(x) for a bias-free nn.Linear; in training on the GPU through _LinearTN."""
    if DGRAD_TN and module.bias is None and x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
        return _LinearTN.apply(x, module.weight)
    return module(x)


class _Mix(torch.autograd.Function):
    """token-shift + M lerps:  out_m = x + (shift(x) - x) * mu_m."""

    @staticmethod
    def forward(ctx, x, *mus):
        return _Mix._forward(ctx, False, x, *mus)

    @staticmethod
    def _forward(ctx, dup3, x, *mus):
        B, T, C = x.shape
        x = x.contiguous()
        mus_c = [m.reshape(C).contiguous() for m in mus]
        _chk(x, *mus_c)
        outs = [torch.empty_like(x) for _ in mus]
        rc = hip_lib.load().vrwkv_mix_fwd_bf16(B * T, T, C, len(mus), x.data_ptr(), _ptr_array(mus_c), _ptr_array(outs), _stream(x))
        hip_lib.check(rc, "vrwkv_mix_fwd_bf16")
        ctx.save_for_backward(x, *mus_c)
        ctx.mu_shapes = [m.shape for m in mus]
        if dup3:                 # a 7th output aliasing output 3 (x_v) for its second consumer
            outs.append(outs[3].view_as(outs[3]))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        x, *mus_c = ctx.saved_tensors
        B, T, C = x.shape
        douts = [d.contiguous() for d in douts]
        _chk(*douts)
        M = len(mus_c)
        second = douts[M] if len(douts) > M else None          # gradient of the alias of output 3
        dx = torch.empty_like(x)
        dmu = torch.empty(M, C, dtype=torch.float32, device=x.device)
        ws = _ws(B * T, C, M, x.device)
        rc = hip_lib.load().vrwkv_mix_bwd2_bf16(B * T, T, C, M, x.data_ptr(), _ptr_array(mus_c), _ptr_array(douts[:M]), _p(second),
                                                dx.data_ptr(), dmu.data_ptr(), ws.data_ptr(), _stream(x))
        hip_lib.check(rc, "vrwkv_mix_bwd2_bf16")
        dmu = dmu.to(x.dtype)
        return (dx, *[dmu[i].view(s) for i, s in enumerate(ctx.mu_shapes)])


class _MixDup3(_Mix):
    """`_Mix` for the time-mix of layers > 0: x_v is returned twice (the second an alias) so that the gradients of its two
    consumers reach mix_bwd as separate inputs and are summed there, not by an element-wise kernel of autograd."""

    @staticmethod
    def forward(ctx, x, *mus):
        return _Mix._forward(ctx, True, x, *mus)


class _Decay(torch.autograd.Function):
    """w = -softplus(-(w0 + h)) - 0.5"""

    @staticmethod
    def forward(ctx, h, w0):
        h = h.contiguous()
        C = h.shape[-1]
        w0c = w0.reshape(C).contiguous()
        _chk(h, w0c)
        w = torch.empty_like(h)
        rc = hip_lib.load().vrwkv_decay_fwd_bf16(h.numel() // C, C, h.data_ptr(), w0c.data_ptr(), w.data_ptr(), _stream(h))
        hip_lib.check(rc, "vrwkv_decay_fwd_bf16")
        ctx.save_for_backward(h, w0c)
        ctx.w0_shape = w0.shape
        return w

    @staticmethod
    def backward(ctx, dw):
        h, w0c = ctx.saved_tensors
        C = h.shape[-1]
        dw = dw.contiguous()
        _chk(dw)
        dh = torch.empty_like(h)
        dw0 = torch.empty(C, dtype=torch.float32, device=h.device)
        ws = _ws(h.numel() // C, C, 1, h.device)
        rc = hip_lib.load().vrwkv_decay_bwd_bf16(h.numel() // C, C, h.data_ptr(), w0c.data_ptr(), dw.data_ptr(), dh.data_ptr(),
                                                 dw0.data_ptr(), ws.data_ptr(), _stream(h))
        hip_lib.check(rc, "vrwkv_decay_bwd_bf16")
        return dh, dw0.to(h.dtype).view(ctx.w0_shape)


class _Kva(torch.autograd.Function):
    """(k, v, v_first, vl, al; k_k, k_a, a0, v0) -> (k2, v2, z, b); v/v_first/vl/v0 are None for layer 0.
    dup=True appends aliases of k2 (and v2) for a second consumer: their gradients reach the backward kernel as
    separate inputs and are summed there (autograd would run one 3 x 172 MB element-wise add per tensor and layer).
    chain=True (layers > 0) appends an alias of v_first, which the NEXT layer uses as its v_first: the gradient of v_first then
    travels down the layers as one running sum that each kva backward adds its term to, instead of 23 terms for autograd to add."""

    @staticmethod
    def forward(ctx, k, v, v_first, vl, al, k_k, k_a, a0, v0, *flags):
        dup = bool(flags[0]) if len(flags) > 0 else False
        has = v is not None
        chain = (bool(flags[1]) if len(flags) > 1 else False) and has
        ctx.nflags = len(flags)
        ctx.set_materialize_grads(False)         # the last layer's alias of v_first has no consumer: its gradient arrives as None
        k, al = k.contiguous(), al.contiguous()
        C = k.shape[-1]
        ntok = k.numel() // C
        if has:
            v, v_first, vl = v.contiguous(), v_first.contiguous(), vl.contiguous()
        pk, pa, p0 = k_k.reshape(C).contiguous(), k_a.reshape(C).contiguous(), a0.reshape(C).contiguous()
        pv = v0.reshape(C).contiguous() if has else None
        _chk(k, al, v, v_first, vl, pk, pa, p0, pv)
        k2, z, b = torch.empty_like(k), torch.empty_like(k), torch.empty_like(k)
        v2 = torch.empty_like(k) if has else None
        rc = hip_lib.load().vrwkv_kva_fwd_bf16(ntok, C, int(has), k.data_ptr(), _p(v), _p(v_first), _p(vl), al.data_ptr(),
                                               pk.data_ptr(), pa.data_ptr(), p0.data_ptr(), _p(pv),
                                               k2.data_ptr(), _p(v2), z.data_ptr(), b.data_ptr(), _stream(k))
        hip_lib.check(rc, "vrwkv_kva_fwd_bf16")
        ctx.has = has
        ctx.dup = bool(dup)
        ctx.chain = chain
        ctx.shapes = (k_k.shape, k_a.shape, a0.shape, v0.shape if has else None)
        ctx.save_for_backward(k, v, v_first, vl, al, pk, pa, p0, pv)
        outs = (k2, v2, z, b) if has else (k2, z, b)
        if dup:                 # aliases for the second consumer (`post`): autograd then delivers their gradients separately
            outs += (k2.view_as(k2), v2.view_as(v2)) if has else (k2.view_as(k2),)
        if chain:
            outs += (v_first.view_as(v_first),)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        k, v, v_first, vl, al, pk, pa, p0, pv = ctx.saved_tensors
        has = ctx.has
        n = 4 if has else 3
        dvf_in = None
        if ctx.chain:
            dvf_in, grads = grads[-1], grads[:-1]
            dvf_in = dvf_in.contiguous() if dvf_in is not None else None
        if any(g is None for g in grads):         # set_materialize_grads(False): an output nobody used
            grads = [g if g is not None else torch.zeros_like(k) for g in grads]
        main, extra = [g.contiguous() for g in grads[:n]], [g.contiguous() for g in grads[n:]]
        if has:
            dk2, dv2, dz, db = main
        else:
            dk2, dz, db = main
            dv2 = None
        dk2b = extra[0] if extra else None
        dv2b = extra[1] if len(extra) > 1 else None
        _chk(dk2, dv2, dz, db, dk2b, dv2b, dvf_in)
        C = k.shape[-1]
        ntok = k.numel() // C
        dk, dal = torch.empty_like(k), torch.empty_like(k)
        dv = torch.empty_like(k) if has else None
        dvf = torch.empty_like(k) if has else None
        dvl = torch.empty_like(k) if has else None
        pg = torch.empty(4, C, dtype=torch.float32, device=k.device)
        ws = _ws(ntok, C, 4, k.device)
        rc = hip_lib.load().vrwkv_kva_bwd3_bf16(ntok, C, int(has), k.data_ptr(), _p(v), _p(v_first), _p(vl), al.data_ptr(),
                                                pk.data_ptr(), pa.data_ptr(), p0.data_ptr(), _p(pv),
                                                dk2.data_ptr(), _p(dv2), dz.data_ptr(), db.data_ptr(), _p(dk2b), _p(dv2b), _p(dvf_in),
                                                dk.data_ptr(), _p(dv), _p(dvf), _p(dvl), dal.data_ptr(),
                                                pg.data_ptr(), ws.data_ptr(), _stream(k))
        hip_lib.check(rc, "vrwkv_kva_bwd3_bf16")
        pgb = pg.to(k.dtype)
        s = ctx.shapes
        res = (dk, dv, dvf, dvl, dal, pgb[0].view(s[0]), pgb[1].view(s[1]), pgb[2].view(s[2]), pgb[3].view(s[3]) if has else None)
        return res + (None,) * ctx.nflags          # dup, chain


class _Post(torch.autograd.Function):
    """out = (GroupNorm(y) + (sum_head r*k*r_k) * v) * g"""

    @staticmethod
    def forward(ctx, y, r, k, v, g, ln_w, ln_b, r_k, eps):
        y, r, k, v, g = [t.contiguous() for t in (y, r, k, v, g)]
        C = y.shape[-1]
        lw, lb, rk = ln_w.contiguous(), ln_b.contiguous(), r_k.reshape(C).contiguous()
        _chk(y, r, k, v, g, lw, lb, rk)
        out = torch.empty_like(y)
        rc = hip_lib.load().vrwkv_post_fwd_bf16(y.numel() // C, C, float(eps), y.data_ptr(), r.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                g.data_ptr(), lw.data_ptr(), lb.data_ptr(), rk.data_ptr(), out.data_ptr(), _stream(y))
        hip_lib.check(rc, "vrwkv_post_fwd_bf16")
        ctx.save_for_backward(y, r, k, v, g, lw, lb, rk)
        ctx.eps = float(eps)
        ctx.rk_shape = r_k.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        y, r, k, v, g, lw, lb, rk = ctx.saved_tensors
        dout = dout.contiguous()
        _chk(dout)
        C = y.shape[-1]
        dy, dr, dk, dv, dg = [torch.empty_like(y) for _ in range(5)]
        pg = torch.empty(3, C, dtype=torch.float32, device=y.device)
        ws = _ws(y.numel() // C, C, 3, y.device)
        rc = hip_lib.load().vrwkv_post_bwd_bf16(y.numel() // C, C, ctx.eps, y.data_ptr(), r.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                g.data_ptr(), lw.data_ptr(), lb.data_ptr(), rk.data_ptr(), dout.data_ptr(),
                                                dy.data_ptr(), dr.data_ptr(), dk.data_ptr(), dv.data_ptr(), dg.data_ptr(),
                                                pg.data_ptr(), ws.data_ptr(), _stream(y))
        hip_lib.check(rc, "vrwkv_post_bwd_bf16")
        pgb = pg.to(y.dtype)
        return dy, dr, dk, dv, dg, pgb[0], pgb[1], pgb[2].view(ctx.rk_shape), None


class _ReluSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        h = h.contiguous()
        _chk(h)
        y = torch.empty_like(h)
        rc = hip_lib.load().vrwkv_relusq_fwd_bf16(h.numel(), h.data_ptr(), y.data_ptr(), _stream(h))
        hip_lib.check(rc, "vrwkv_relusq_fwd_bf16")
        ctx.save_for_backward(h)
        return y

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        dy = dy.contiguous()
        _chk(dy)
        dh = torch.empty_like(h)
        rc = hip_lib.load().vrwkv_relusq_bwd_bf16(h.numel(), h.data_ptr(), dy.data_ptr(), dh.data_ptr(), _stream(h))
        hip_lib.check(rc, "vrwkv_relusq_bwd_bf16")
        return dh


class _ReluSqLinear(torch.autograd.Function):
    """value(relu(h)^2) of the channel-mix (src/model.py:225-226) WITHOUT keeping relu(h)^2 for the backward: 4 of the ~40 activation
    tensors a layer keeps (it is as wide as the FFN).  The backward forms it again from h (one streaming kernel, 0.21 ms per layer at
    micro-batch 16) for the weight gradient.  Selective recompute (`grad_cp=2`) only; gradients are those of relu_sq + _LinearTN."""

    @staticmethod
    def forward(ctx, h, w):
        h = h.contiguous()
        _chk(h)
        y = torch.empty_like(h)
        hip_lib.check(hip_lib.load().vrwkv_relusq_fwd_bf16(h.numel(), h.data_ptr(), y.data_ptr(), _stream(h)), "vrwkv_relusq_fwd_bf16")
        ctx.save_for_backward(h, w)
        ctx.wparam = w if hasattr(w, "_vrwkv_flat_grad") else None
        return F.linear(y, w)

    @staticmethod
    def backward(ctx, dy):
        h, w = ctx.saved_tensors
        lib = hip_lib.load()
        dy = dy.contiguous()
        dw = None
        both = (ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and _wgrad_beside_dgrad(ctx.wparam, dy.reshape(-1, dy.shape[-1]), h.reshape(-1, h.shape[-1])))
        cur = torch.cuda.current_stream(h.device)
        side = _side_stream(h.device) if both else cur
        if ctx.needs_input_grad[1]:                          # relu(h)^2 again, only for the weight gradient (on the side stream beside the input gradient: OVERLAP_WGRAD)
            if both:
                side.wait_stream(cur)
            with torch.cuda.stream(side):
                y = torch.empty_like(h)
                hip_lib.check(lib.vrwkv_relusq_fwd_bf16(h.numel(), h.data_ptr(), y.data_ptr(), _stream(h)), "vrwkv_relusq_fwd_bf16")
                dw = _weight_grad(ctx.wparam, dy.reshape(-1, dy.shape[-1]), y.reshape(-1, y.shape[-1]))
                del y
        dh = None
        if ctx.needs_input_grad[0]:
            dyy = F.linear(dy, transpose2d(w))               # gradient of relu(h)^2 (the allocator hands it the bytes just freed)
            dh = torch.empty_like(h)
            hip_lib.check(lib.vrwkv_relusq_bwd_bf16(h.numel(), h.data_ptr(), dyy.data_ptr(), dh.data_ptr(), _stream(h)), "vrwkv_relusq_bwd_bf16")
        if both:
            cur.wait_stream(side)
            dw.record_stream(cur)
        return dh, dw


mix = _Mix.apply
mix_dup3 = _MixDup3.apply
decay = _Decay.apply
kva = _Kva.apply
post = _Post.apply
relu_sq = _ReluSq.apply


class _AddLN(torch.autograd.Function):
    """(xn, y) = (x + delta, LayerNorm(x + delta));  with delta None: y = LayerNorm(x) only (csrc/ln_fused.hip)."""

    @staticmethod
    def forward(ctx, x, delta, w, b, eps):
        C = x.shape[-1]
        x = x.contiguous()
        delta = delta.contiguous() if delta is not None else None
        wc, bc = w.contiguous(), b.contiguous()
        _chk(x, delta, wc, bc)
        if delta is not None and delta.shape != x.shape:
            raise ValueError("add_ln: x and delta must have the same shape")
        ntok = x.numel() // C
        xn = torch.empty_like(x) if delta is not None else x
        y = torch.empty_like(x)
        mean = torch.empty(ntok, dtype=torch.float32, device=x.device)
        rstd = torch.empty(ntok, dtype=torch.float32, device=x.device)
        rc = hip_lib.load().vrwkv_add_ln_fwd_bf16(ntok, C, float(eps), x.data_ptr(), _p(delta), wc.data_ptr(), bc.data_ptr(),
                                                  xn.data_ptr() if delta is not None else 0, y.data_ptr(),
                                                  mean.data_ptr(), rstd.data_ptr(), _stream(x))
        hip_lib.check(rc, "vrwkv_add_ln_fwd_bf16")
        ctx.save_for_backward(xn, mean, rstd, wc)
        ctx.has_delta = delta is not None
        if delta is None:
            return y
        return xn, y

    @staticmethod
    def backward(ctx, *grads):
        xn, mean, rstd, wc = ctx.saved_tensors
        d_xn, dy = grads if ctx.has_delta else (None, grads[0])
        C = xn.shape[-1]
        ntok = xn.numel() // C
        dy = dy.contiguous()
        d_xn = d_xn.contiguous() if d_xn is not None else None
        _chk(dy, d_xn)
        dx = torch.empty_like(xn)
        dwb = torch.empty(2, C, dtype=torch.float32, device=xn.device)
        lib = hip_lib.load()
        ws = torch.empty(lib.vrwkv_add_ln_ws_floats(ntok, C), dtype=torch.float32, device=xn.device)
        rc = lib.vrwkv_add_ln_bwd_bf16(ntok, C, dy.data_ptr(), _p(d_xn), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       wc.data_ptr(), dx.data_ptr(), dwb.data_ptr(), ws.data_ptr(), _stream(xn))
        hip_lib.check(rc, "vrwkv_add_ln_bwd_bf16")
        dwb = dwb.to(wc.dtype)
        return dx, (dx if ctx.has_delta else None), dwb[0], dwb[1], None


def add_ln(x, delta, ln):
    """Residual add + nn.LayerNorm `ln` in one kernel: returns (x + delta, ln(x + delta)); delta may be None."""
    if delta is None:
        return x, _AddLN.apply(x, None, ln.weight, ln.bias, ln.eps)
    return _AddLN.apply(x, delta, ln.weight, ln.bias, ln.eps)


@torch.no_grad()
def add_ln_infer(x, delta, dscale, ln):
    """Inference form (frozen ViT towers): returns (x + delta * dscale, LayerNorm(x + delta * dscale)); delta / dscale may be
    None.  One pass instead of the eager LayerScale multiply, residual add and LayerNorm."""
    C = x.shape[-1]
    x = x.contiguous()
    delta = delta.contiguous() if delta is not None else None
    _chk(x, delta, ln.weight, ln.bias, dscale)
    xn = torch.empty_like(x) if delta is not None else x
    y = torch.empty_like(x)
    rc = hip_lib.load().vrwkv_add_ln_scaled_fwd_bf16(x.numel() // C, C, float(ln.eps), x.data_ptr(), _p(delta), _p(dscale), ln.weight.data_ptr(),
                                                     ln.bias.data_ptr(), xn.data_ptr() if delta is not None else 0, y.data_ptr(), _stream(x))
    hip_lib.check(rc, "vrwkv_add_ln_scaled_fwd_bf16")
    return xn, y


class _AddLnMix(torch.autograd.Function):
    """(xn, out_0 .. out_{M-1}) = (x + delta, lerps of the token-shifted LayerNorm(x + delta)): `_AddLN` followed by `_Mix` in one
    kernel each way (csrc/ln_fused.hip: ln_mix_*): the LayerNorm output is never written.  M = 1 (channel-mix) or 6 (time-mix);
    dup3: a 7th output aliasing output 3 (x_v) for its second consumer, as `_MixDup3`.  delta may be None (first block)."""

    @staticmethod
    def forward(ctx, x, delta, w, b, eps, dup3, *mus):
        B, T, C = x.shape
        M = len(mus)
        x = x.contiguous()
        delta = delta.contiguous() if delta is not None else None
        wc, bc = w.contiguous(), b.contiguous()
        mus_c = [m.reshape(C).contiguous() for m in mus]
        _chk(x, delta, wc, bc, *mus_c)
        ntok = B * T
        xn = torch.empty_like(x) if delta is not None else x
        outs = [torch.empty_like(x) for _ in mus]
        mean = torch.empty(ntok, dtype=torch.float32, device=x.device)
        rstd = torch.empty(ntok, dtype=torch.float32, device=x.device)
        rc = hip_lib.load().vrwkv_ln_mix_fwd_bf16(ntok, T, C, float(eps), M, x.data_ptr(), _p(delta), wc.data_ptr(), bc.data_ptr(),
                                                  _ptr_array(mus_c), xn.data_ptr() if delta is not None else 0, _ptr_array(outs),
                                                  mean.data_ptr(), rstd.data_ptr(), _stream(x))
        hip_lib.check(rc, "vrwkv_ln_mix_fwd_bf16")
        ctx.save_for_backward(xn, mean, rstd, wc, bc, *mus_c)
        ctx.has_delta = delta is not None
        ctx.mu_shapes = [m.shape for m in mus]
        if dup3:
            outs.append(outs[3].view_as(outs[3]))
        return (xn, *outs)

    @staticmethod
    def backward(ctx, d_xn, *douts):
        xn, mean, rstd, wc, bc, *mus_c = ctx.saved_tensors
        B, T, C = xn.shape
        M = len(mus_c)
        douts = [d.contiguous() for d in douts]
        d_xn = d_xn.contiguous() if d_xn is not None else None
        _chk(d_xn, *douts)
        second = douts[M] if len(douts) > M else None
        ntok = B * T
        dx = torch.empty_like(xn)
        dwb = torch.empty(2, C, dtype=torch.float32, device=xn.device)
        dmu = torch.empty(M, C, dtype=torch.float32, device=xn.device)
        lib = hip_lib.load()
        if M == 1:
            ws = torch.empty(lib.vrwkv_ln_mix_ws_floats(ntok, C, M), dtype=torch.float32, device=xn.device)
            rc = lib.vrwkv_ln_mix_bwd_bf16(ntok, T, C, M, xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                           _ptr_array(mus_c), _ptr_array(douts[:M]), _p(second), _p(d_xn), dx.data_ptr(), dwb.data_ptr(),
                                           dmu.data_ptr(), ws.data_ptr(), _stream(xn))
            hip_lib.check(rc, "vrwkv_ln_mix_bwd_bf16")
        else:       # six lerps: their backward with the LayerNorm output recomputed in place of a stored one, then the LayerNorm's
            dy = torch.empty_like(xn)
            ws = _ws(ntok, C, M, xn.device)
            rc = lib.vrwkv_mix_bwd_ln_bf16(ntok, T, C, M, xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                           _ptr_array(mus_c), _ptr_array(douts[:M]), _p(second), dy.data_ptr(), dmu.data_ptr(),
                                           ws.data_ptr(), _stream(xn))
            hip_lib.check(rc, "vrwkv_mix_bwd_ln_bf16")
            ws = torch.empty(lib.vrwkv_add_ln_ws_floats(ntok, C), dtype=torch.float32, device=xn.device)
            rc = lib.vrwkv_add_ln_bwd_bf16(ntok, C, dy.data_ptr(), _p(d_xn), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           wc.data_ptr(), dx.data_ptr(), dwb.data_ptr(), ws.data_ptr(), _stream(xn))
            hip_lib.check(rc, "vrwkv_add_ln_bwd_bf16")
        dwb = dwb.to(wc.dtype)
        dmu = dmu.to(xn.dtype)
        return (dx, (dx if ctx.has_delta else None), dwb[0], dwb[1], None, None, *[dmu[i].view(sh) for i, sh in enumerate(ctx.mu_shapes)])


def add_ln_mix(x, delta, ln, mus, dup3=False):
    """Returns (x + delta, [lerp outputs]) -- see `_AddLnMix`."""
    xn, *outs = _AddLnMix.apply(x, delta, ln.weight, ln.bias, ln.eps, dup3, *mus)
    return xn, outs


LN_MIX = os.environ.get("VRWKV_LN_MIX", "1") != "0"              # A/B switch: 0 = add_ln and mix as two kernels
LN_MIX_TMIX = os.environ.get("VRWKV_LN_MIX_TMIX", "1") != "0"    # A/B switch: 0 = ln1 and the six time-mix lerps as two kernels


def ln_mix_supported(x):
    return LN_MIX and add_ln_supported(x) and x.dim() == 3 and x.shape[-1] <= 4096


def add_ln_supported(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 64 == 0 and x.shape[-1] <= 8192


def _block_segment(block, x, delta, v_first, selective=False):
    """One Block on the (x, pending delta) residual stream: returns (x + delta, ffn output still to be added, v_first).
    selective: the selective-recompute mode of blocks_forward (WKV7 by-products and relu(h)^2 are re-formed in the backward)."""
    att, ffn = block.att, block.ffn
    fuse = ln_mix_supported(x) and getattr(att.args, "fused", False)
    if fuse and LN_MIX_TMIX:
        dup3 = torch.is_grad_enabled() and GRAD_ALIAS and att.layer_id > 0
        x, mixed = add_ln_mix(x, delta, block.ln1, (att.x_r, att.x_w, att.x_k, att.x_v, att.x_a, att.x_g), dup3)
        att_out, v_first = tmix_from_mixed(att, mixed, v_first, recompute_state=selective)
    else:
        x, h = add_ln(x, delta, block.ln1)
        att_out, v_first = tmix_forward(att, h, v_first, recompute_state=selective) if getattr(att.args, "fused", False) else att(h, v_first)
    if fuse:            # ln2 + the channel-mix lerp in one kernel: the LayerNorm output is never materialised
        x, (k,) = add_ln_mix(x, att_out, block.ln2, (ffn.x_k,))
        return x, cmix_from_mixed(ffn, k, recompute_relusq=selective), v_first
    x, h = add_ln(x, att_out, block.ln2)
    return x, ffn(h), v_first


def blocks_forward(rwkv, x, grad_cp=0):
    """All Blocks + ln_out with the residual adds fused into the LayerNorms (same math as Block.forward chained,
    src/model.py:247-254,313-318): the residual stream is carried as (x, pending delta).
    grad_cp (the reference's memory-saving switch, src/model.py:318-319: deepspeed.checkpointing.checkpoint per block):
      0  keep every activation (288 GB of HBM hold the 1.5B model at micro-batch 16: 192 GB);
      1  THE REFERENCE'S RECIPE, same memory behaviour: every Block re-computed in the backward (block inputs only are kept: 42 GB) -- through
         these same fused kernels, so the recompute and the backward use add_ln / the glue kernels / the WKV7 op, not the eager modules;
      2  (not in the reference) SELECTIVE recompute: keep what is expensive to recompute (every GEMM output), drop what is cheap to recompute and
         large -- the WKV7 chunk checkpoints `s` and `sa` (10 of the ~40 activation tensors of a layer: the backward re-runs the forward kernel)
         and relu(h)^2 of the channel-mix (4 of them: one streaming kernel) -- about a third of the activation memory for ~1 ms per layer.
    (Rounds 4-5 had 1 and 2 the other way round; a trainer configured for the reference's `--grad_cp 1` must not get the mode that needs 3x
    the memory.)"""
    grad_cp = int(grad_cp) if torch.is_grad_enabled() else 0
    if grad_cp not in (0, 1, 2):
        raise ValueError(f"grad_cp = {grad_cp}: 0 (keep everything), 1 (re-compute every block, the reference's recipe) or 2 (selective recompute)")
    x = rwkv.blocks[0].ln0(x)
    v_first = torch.empty_like(x)
    delta = None
    for block in rwkv.blocks:
        if grad_cp == 1:
            from torch.utils.checkpoint import checkpoint
            x, delta, v_first = checkpoint(_block_segment, block, x, delta, v_first, use_reentrant=False)
        else:
            x, delta, v_first = _block_segment(block, x, delta, v_first, grad_cp == 2)
    _, h = add_ln(x, delta, rwkv.ln_out)
    return h


class _FusedCE(torch.autograd.Function):
    """training_step's loss (shifted CE, per-sample sum / max(valid,1), batch mean) with L2Wrap's gradient term
    (src/model.py:418-434,257-271) -- csrc/loss_fused.hip."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        B, T, V = logits.shape
        logits = logits.contiguous()
        _chk(logits)
        labels = torch.full((B, T), ignore_index, dtype=torch.long, device=logits.device)
        labels[:, :-1] = targets[:, 1:]                       # row (b,t) predicts token t+1; the last row has no target
        labels = torch.where(labels == ignore_index, torch.full_like(labels, -100), labels)
        valid = (labels >= 0).sum(1).clamp(min=1)
        n = B * T
        dev = logits.device
        row_loss, row_max, row_lse = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3))
        row_arg = torch.empty(n, dtype=torch.int32, device=dev)
        rc = hip_lib.load().vrwkv_ce_fwd_bf16(n, V, logits.data_ptr(), labels.data_ptr(), row_loss.data_ptr(),
                                              row_max.data_ptr(), row_lse.data_ptr(), row_arg.data_ptr(), _stream(logits))
        hip_lib.check(rc, "vrwkv_ce_fwd_bf16")
        w = ((labels >= 0).float() / (valid.float().unsqueeze(1) * B)).view(n)      # d loss / d row_loss
        ctx.save_for_backward(logits, labels, w, row_max, row_lse, row_arg)
        return (row_loss * w).sum().to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        logits, labels, w, row_max, row_lse, row_arg = ctx.saved_tensors
        B, T, V = logits.shape
        dlogits = torch.empty_like(logits)
        row_w = (w * g.float()).contiguous()
        rc = hip_lib.load().vrwkv_ce_bwd_bf16(B * T, V, logits.data_ptr(), labels.data_ptr(), row_w.data_ptr(),
                                              row_max.data_ptr(), row_lse.data_ptr(), row_arg.data_ptr(),
                                              1e-4 / (B * T), dlogits.data_ptr(), _stream(logits))
        hip_lib.check(rc, "vrwkv_ce_bwd_bf16")
        return dlogits, None, None


def loss_from_logits(logits, targets, ignore_index=-100):
    return _FusedCE.apply(logits, targets, ignore_index)


def ce_supported(logits):
    return logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 3 and logits.shape[-1] % 8 == 0


# ---------------------------------------------------------------------------------------------------------------
# RWKV-6 glue (BASELINE config 4; VisualRWKV-v6/v6.0/src/model.py:146-194, 213-226)
# ---------------------------------------------------------------------------------------------------------------
class _DDMix(torch.autograd.Function):
    """RWKV-6's data-dependent token shift: out_j = x + (shift(x) - x) * (mu_j + mm_j), j < 5, mm_j (B,T,C) per token."""

    @staticmethod
    def forward(ctx, x, mm, *mus):
        B, T, C = x.shape
        x = x.contiguous()
        mm = mm.contiguous()                                # (5, B, T, C)
        mus_c = [m.reshape(C).contiguous() for m in mus]
        _chk(x, mm, *mus_c)
        assert len(mus) == 5 and tuple(mm.shape) == (5, B, T, C)
        outs = [torch.empty_like(x) for _ in range(5)]
        rc = hip_lib.load().vrwkv_ddmix_fwd_bf16(B * T, T, C, x.data_ptr(), _ptr_array(mus_c), _ptr_array(list(mm.unbind(0))),
                                                 _ptr_array(outs), _stream(x))
        hip_lib.check(rc, "vrwkv_ddmix_fwd_bf16")
        ctx.save_for_backward(x, mm, *mus_c)
        ctx.mu_shapes = [m.shape for m in mus]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        x, mm, *mus_c = ctx.saved_tensors
        B, T, C = x.shape
        douts = [d.contiguous() for d in douts]
        _chk(*douts)
        dx = torch.empty_like(x)
        dmm = torch.empty_like(mm)
        dmu = torch.empty(5, C, dtype=torch.float32, device=x.device)
        ws = _ws(B * T, C, 5, x.device)
        rc = hip_lib.load().vrwkv_ddmix_bwd_bf16(B * T, T, C, x.data_ptr(), _ptr_array(mus_c), _ptr_array(list(mm.unbind(0))),
                                                 _ptr_array(douts), dx.data_ptr(), _ptr_array(list(dmm.unbind(0))), dmu.data_ptr(),
                                                 ws.data_ptr(), _stream(x))
        hip_lib.check(rc, "vrwkv_ddmix_bwd_bf16")
        dmu = dmu.to(x.dtype)
        return (dx, dmm, *[dmu[i].view(s) for i, s in enumerate(ctx.mu_shapes)])


class _GnSilu(torch.autograd.Function):
    """out = GroupNorm(C/64 groups)(y) * silu(gg)   (RWKV_Tmix_x060.jit_func_2 with the silu of jit_func)"""

    @staticmethod
    def forward(ctx, y, gg, ln_w, ln_b, eps):
        y, gg = y.contiguous(), gg.contiguous()
        C = y.shape[-1]
        lw, lb = ln_w.contiguous(), ln_b.contiguous()
        _chk(y, gg, lw, lb)
        out = torch.empty_like(y)
        rc = hip_lib.load().vrwkv_gn_silu_fwd_bf16(y.numel() // C, C, float(eps), y.data_ptr(), gg.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                                   out.data_ptr(), _stream(y))
        hip_lib.check(rc, "vrwkv_gn_silu_fwd_bf16")
        ctx.save_for_backward(y, gg, lw, lb)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gg, lw, lb = ctx.saved_tensors
        dout = dout.contiguous()
        _chk(dout)
        C = y.shape[-1]
        dy, dgg = torch.empty_like(y), torch.empty_like(gg)
        pg = torch.empty(2, C, dtype=torch.float32, device=y.device)
        ws = _ws(y.numel() // C, C, 2, y.device)
        rc = hip_lib.load().vrwkv_gn_silu_bwd_bf16(y.numel() // C, C, ctx.eps, y.data_ptr(), gg.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                                   dout.data_ptr(), dy.data_ptr(), dgg.data_ptr(), pg.data_ptr(), ws.data_ptr(), _stream(y))
        hip_lib.check(rc, "vrwkv_gn_silu_bwd_bf16")
        pgb = pg.to(y.dtype)
        return dy, dgg, pgb[0], pgb[1], None


ddmix = _DDMix.apply
gn_silu = _GnSilu.apply


def supported6(x, m=None):
    """bf16 CUDA activations with C a multiple of 64 (the glue kernels' 8-channel lanes and 64-channel heads).  With the time-mix
    module `m`: its GroupNorm must be the 64-channel-per-head one over all C channels that gn_silu hard-codes (head_size 64,
    dim_att == n_embd); anything else takes the eager path."""
    ok = x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 64 == 0 and x.shape[-1] <= 8192
    if ok and m is not None and hasattr(m, "ln_x"):
        ok = getattr(m, "head_size", 64) == 64 and m.ln_x.num_groups * 64 == x.shape[-1] == m.ln_x.num_channels
    return ok


def tmix6_forward(m, x, wkv=None):
    """RWKV_Tmix_x060.forward (VisualRWKV-v6/v6.0/src/model.py:146-194) with the glue fused: one kernel for the first lerp
    (xxx), one for the five data-dependent lerps, the T,N input-gradient layout for the projections, GroupNorm * silu(gate)
    in one pass.  `m` is the module (parameter names as in the reference)."""
    from . import wkv6 as _wkv6
    B, T, C = x.shape
    mm_ = lora_mm if torch.is_grad_enabled() and LORA_WGRAD else torch.matmul
    (xxx,) = mix(x, m.time_maa_x)
    h = torch.tanh(mm_(xxx, m.time_maa_w1)).view(B * T, 5, -1).transpose(0, 1)
    mm5 = torch.bmm(h, m.time_maa_w2).view(5, B, T, C)
    xw, xk, xv, xr, xg = ddmix(x, mm5, m.time_maa_w, m.time_maa_k, m.time_maa_v, m.time_maa_r, m.time_maa_g)
    r = linear(m.receptance, xr)
    k = linear(m.key, xk)
    v = linear(m.value, xv)
    gg = linear(m.gate, xg)
    w = m.time_decay + mm_(torch.tanh(mm_(xw, m.time_decay_w1)), m.time_decay_w2)
    run = wkv if wkv is not None else _wkv6.RUN_CUDA_RWKV6
    y = run(B, T, C, m.n_head, r, k, v, w, m.time_faaaa)
    y = gn_silu(y.reshape(B * T, C), gg.reshape(B * T, C), m.ln_x.weight, m.ln_x.bias, m.ln_x.eps).view(B, T, C)
    return linear(m.output, y)


def cmix6_forward(m, x):
    """RWKV_CMix_x060.forward (model.py:213-226): two lerps in one pass, relu^2, sigmoid(receptance) * value in one pass."""
    xk, xr = mix(x, m.time_maa_k, m.time_maa_r)
    kv = linear(m.value, relu_sq(linear(m.key, xk)))
    return gate(kv, linear(m.receptance, xr))


def blocks6_forward(rwkv, x, wkv=None, grad_cp=False):
    """RWKV-6 Blocks + ln_out with the residual adds fused into the LayerNorms (model.py:233-258,300-325)."""
    x = rwkv.blocks[0].ln0(x)
    delta = None

    def seg(block, x, delta):
        x, h = add_ln(x, delta, block.ln1)
        x, h = add_ln(x, tmix6_forward(block.att, h, wkv), block.ln2)
        return x, cmix6_forward(block.ffn, h)

    for block in rwkv.blocks:
        if grad_cp:
            from torch.utils.checkpoint import checkpoint
            x, delta = checkpoint(seg, block, x, delta, use_reentrant=False)
        else:
            x, delta = seg(block, x, delta)
    _, h = add_ln(x, delta, rwkv.ln_out)
    return h


def tmix_forward(m, x, v_first, recompute_state=False):
    """RWKV_Tmix_x070.forward (src/model.py:163-195) with the glue fused; `m` is the module."""
    train = torch.is_grad_enabled()
    if train and GRAD_ALIAS and m.layer_id > 0:          # x_v, k2, v2 have two consumers each: aliases keep their gradients apart until the
        mixed = mix_dup3(x, m.x_r, m.x_w, m.x_k, m.x_v, m.x_a, m.x_g)      # backward kernels sum them
    else:
        mixed = mix(x, m.x_r, m.x_w, m.x_k, m.x_v, m.x_a, m.x_g)
    return tmix_from_mixed(m, mixed, v_first, recompute_state)


def tmix_from_mixed(m, mixed, v_first, recompute_state=False):
    """The time-mix after its token shift: `mixed` = (xr, xw, xk, xv, xa, xg[, alias of xv for its second consumer])."""
    xr, xw, xk, xv, xa, xg = mixed[:6]
    xv_b = mixed[6] if len(mixed) > 6 else xv
    mm = lora_mm if torch.is_grad_enabled() and LORA_WGRAD else torch.matmul     # training: skinny weight-gradient kernel in the backward
    r, k, v = linear3((m.receptance, m.key, m.value), (xr, xk, xv))
    w = decay(mm(torch.tanh(mm(xw, m.w1)), m.w2), m.w0)
    al = mm(mm(xa, m.a1), m.a2)
    g = mm(torch.sigmoid(mm(xg, m.g1)), m.g2)
    if m.layer_id == 0:
        v_first = v
        k2, z, b, k2_b = kva(k, None, None, None, al, m.k_k, m.k_a, m.a0, None, True)
        k2_b = k2_b if GRAD_ALIAS else k2
        v2 = v2_b = v
    else:
        vl = mm(mm(xv_b, m.v1), m.v2)
        if VF_CHAIN and torch.is_grad_enabled():
            k2, v2, z, b, k2_b, v2_b, v_first = kva(k, v, v_first, vl, al, m.k_k, m.k_a, m.a0, m.v0, True, True)    # v_first: alias for the next layer
        else:
            k2, v2, z, b, k2_b, v2_b = kva(k, v, v_first, vl, al, m.k_k, m.k_a, m.a0, m.v0, True)
        if not GRAD_ALIAS:
            k2_b, v2_b = k2, v2
    y = RUN_CUDA_RWKV7g(r, w, k2, v2, z, b, recompute_state=recompute_state)
    y = post(y, r, k2_b, v2_b, g, m.ln_x.weight, m.ln_x.bias, m.r_k, m.ln_x.eps)
    return linear(m.output, y), v_first


def mix_prev(x, x_prev, *mus):
    """Inference-only `mix` whose shift sees `x_prev` (B,C) before the first token instead of zeros (stateful decode)."""
    B, T, C = x.shape
    x = x.contiguous()
    x_prev = x_prev.contiguous()
    mus_c = [m.reshape(C).contiguous() for m in mus]
    _chk(x, x_prev, *mus_c)
    outs = [torch.empty_like(x) for _ in mus]
    rc = hip_lib.load().vrwkv_mix_fwd_prev_bf16(B * T, T, C, len(mus), x.data_ptr(), x_prev.data_ptr(), _ptr_array(mus_c),
                                                _ptr_array(outs), _stream(x))
    hip_lib.check(rc, "vrwkv_mix_fwd_prev_bf16")
    return tuple(outs)


@torch.no_grad()
def tmix_forward_stateful(m, x, v_first, state):
    """`tmix_forward` continuing from `state` (an RWKV7State): fused glue kernels, WKV through state.wkv."""
    lid = m.layer_id
    xr, xw, xk, xv, xa, xg = mix_prev(x, state.att_x[lid], m.x_r, m.x_w, m.x_k, m.x_v, m.x_a, m.x_g)
    state.att_x[lid].copy_(x[:, -1])
    r = m.receptance(xr)
    w = decay(torch.tanh(xw @ m.w1) @ m.w2, m.w0)
    k = m.key(xk)
    v = m.value(xv)
    al = (xa @ m.a1) @ m.a2
    g = torch.sigmoid(xg @ m.g1) @ m.g2
    if lid == 0:
        v_first = v
        k2, z, b = kva(k, None, None, None, al, m.k_k, m.k_a, m.a0, None)
        v2 = v
    else:
        vl = (xv @ m.v1) @ m.v2
        k2, v2, z, b = kva(k, v, v_first, vl, al, m.k_k, m.k_a, m.a0, m.v0)
    y = state.wkv(lid, r, w, k2, v2, z, b)
    y = post(y, r, k2, v2, g, m.ln_x.weight, m.ln_x.bias, m.r_k, m.ln_x.eps)
    return m.output(y), v_first


@torch.no_grad()
def cmix_forward_stateful(m, x, state):
    lid = m.layer_id
    (k,) = mix_prev(x, state.ffn_x[lid], m.x_k)
    state.ffn_x[lid].copy_(x[:, -1])
    return m.value(relu_sq(m.key(k)))


def cmix_forward(m, x):
    """RWKV_CMix_x070.forward (src/model.py:221-227)."""
    (k,) = mix(x, m.x_k)
    return cmix_from_mixed(m, k)


def cmix_from_mixed(m, k, recompute_relusq=False):
    h = linear(m.key, k)
    if (recompute_relusq and DGRAD_TN and m.value.bias is None and h.is_cuda and h.dtype == torch.bfloat16 and torch.is_grad_enabled()
            and h.requires_grad):
        return _ReluSqLinear.apply(h, m.value.weight)
    return linear(m.value, relu_sq(h))


# ---------------------------------------------------------------------------------------------------------------
# Image side (csrc/visual_ops.hip, vrwkv_ln_scatter_* in csrc/ln_fused.hip): pooling, context gate, ln_v + scatter
# (VisualRWKV-v7/v7.00/src/model.py:328-338,442-447,485-493)
# ---------------------------------------------------------------------------------------------------------------
def visual_supported(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0


def adaptive_pool(image_features, side_out):
    """nn.AdaptiveAvgPool2d(side_out) on token-major ViT features (B, L, D) -> (B, side_out^2, D) without the
    (B, D, H, W) permutes of VisualRWKV.adaptive_pooling; forward only (the towers are frozen and detached)."""
    x = image_features.detach().contiguous()
    _chk(x)
    B, Ln, D = x.shape
    side = int(round(Ln ** 0.5))
    if side * side != Ln:
        raise ValueError(f"{Ln} patch tokens are not a square grid")
    y = torch.empty(B, side_out * side_out, D, dtype=x.dtype, device=x.device)
    hip_lib.check(hip_lib.load().vrwkv_adaptive_pool_bf16(B, side, side_out, D, x.data_ptr(), y.data_ptr(), _stream(x)),
                  "vrwkv_adaptive_pool_bf16")
    return y


class _Gate(torch.autograd.Function):
    """x * sigmoid(g) in one pass (model.py:337); backward dg (and dx when x needs it) in one pass."""

    @staticmethod
    def forward(ctx, x, g):
        x, g = x.contiguous(), g.contiguous()
        _chk(x, g)
        out = torch.empty_like(x)
        hip_lib.check(hip_lib.load().vrwkv_gate_fwd_bf16(x.numel(), x.data_ptr(), g.data_ptr(), out.data_ptr(), _stream(x)), "vrwkv_gate_fwd_bf16")
        ctx.save_for_backward(x, g)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, g = ctx.saved_tensors
        dout = dout.contiguous()
        dg = torch.empty_like(g)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        hip_lib.check(hip_lib.load().vrwkv_gate_bwd_bf16(x.numel(), x.data_ptr(), g.data_ptr(), dout.data_ptr(), dg.data_ptr(), _p(dx), _stream(x)),
                      "vrwkv_gate_bwd_bf16")
        return dx, dg


def gate(x, g):
    return _Gate.apply(x, g)


def gelu_(x, tanh_approx=False):
    """nn.GELU of the frozen towers' MLPs, IN PLACE on a contiguous bf16 device tensor without autograd (timm Mlp through src/vision.py:123-134,
    src/sam.py MLPBlock): csrc/visual_ops.hip, 16 VALU operations per element where the eager erf kernel is VALU-bound at 2.3 TB/s."""
    _chk(x)
    if x.requires_grad or x.numel() % 8 != 0 or not x.is_contiguous():
        raise ValueError("gelu_: contiguous, no autograd, a multiple of 8 elements")
    hip_lib.check(hip_lib.load().vrwkv_gelu_bf16(x.numel(), x.data_ptr(), x.data_ptr(), 1 if tanh_approx else 0, _stream(x)), "vrwkv_gelu_bf16")
    return x


class _LnScatter(torch.autograd.Function):
    """embeds[row_index[n]] = LayerNorm(y[n]): ln_v of the projector fused with the masked scatter into the token
    embeddings (model.py:338 + :485-493).  `embeds` (rows, C) is modified in place and returned."""

    @staticmethod
    def forward(ctx, embeds, y, w, b, row_index, eps):
        y = y.contiguous()
        _chk(embeds, y, w, b)
        n, C = y.shape
        mean = torch.empty(n, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        hip_lib.check(hip_lib.load().vrwkv_ln_scatter_fwd_bf16(n, C, float(eps), y.data_ptr(), w.data_ptr(), b.data_ptr(), row_index.data_ptr(),
                                                               embeds.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream(y)), "vrwkv_ln_scatter_fwd_bf16")
        ctx.mark_dirty(embeds)
        ctx.save_for_backward(y, w, mean, rstd, row_index)
        return embeds

    @staticmethod
    def backward(ctx, dout):
        y, w, mean, rstd, row_index = ctx.saved_tensors
        dout = dout.contiguous()
        n, C = y.shape
        lib = hip_lib.load()
        dy = torch.empty_like(y)
        dwb = torch.empty(2, C, dtype=torch.float32, device=y.device)
        ws = torch.empty(lib.vrwkv_add_ln_ws_floats(n, C), dtype=torch.float32, device=y.device)
        hip_lib.check(lib.vrwkv_ln_gather_bwd_bf16(n, C, dout.data_ptr(), row_index.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   w.data_ptr(), dy.data_ptr(), dwb.data_ptr(), ws.data_ptr(), _stream(y)), "vrwkv_ln_gather_bwd_bf16")
        d_emb = None
        if ctx.needs_input_grad[0]:                    # rows that were overwritten do not reach the embedding
            R = dout.shape[0]                          # dropped features carry row -1: they zero a scratch row past the end
            d_emb = torch.empty(R + 1, C, dtype=dout.dtype, device=dout.device)
            d_emb[:R] = dout
            d_emb.index_fill_(0, torch.where(row_index < 0, R, row_index), 0)
            d_emb = d_emb[:R]
        return d_emb, dy, dwb[0].to(w.dtype), dwb[1].to(w.dtype), None, None


def ln_scatter(embeds2d, y2d, ln, row_index):
    return _LnScatter.apply(embeds2d, y2d, ln.weight, ln.bias, row_index, ln.eps)


# ------------------------------------------------------------------------------------------------
# Patch embedding of the frozen towers (csrc/patch_embed_kernels.h)
# ------------------------------------------------------------------------------------------------
_PATCH_SIZES = (14, 16)


def patch_embed_supported(x, patch, dim):
    if not (visual_supported(x) and x.dim() == 4 and x.shape[1] == 3 and patch in _PATCH_SIZES and dim % 32 == 0):
        return False
    H, W = x.shape[-2:]
    return H % patch == 0 and W % patch == 0 and ((H // patch) * (W // patch)) % 64 == 0


def padded_patch_weight(weight):
    """(N, 3, p, p) conv weight -> (N, KP) bf16 with zero columns up to the MFMA K multiple the kernel reads."""
    N = weight.shape[0]
    K = weight[0].numel()
    KP = hip_lib.load().vrwkv_patch_embed_kp(int(weight.shape[-1]))
    wp = torch.zeros(N, KP, dtype=torch.bfloat16, device=weight.device)
    wp[:, :K] = weight.detach().reshape(N, K)
    return wp


def cached_padded_patch_weight(module, weight):
    """The padded weight kept on the module that owns `weight` (frozen towers: built once); rebuilt when the parameter
    is modified in place, replaced, or an optimizer step bumps the process-wide parameter generation."""
    from . import param_state
    # the optimizer generation only matters for trainable weights (it is bumped every step; the frozen towers' weights are
    # not in the ZeRO flat buffer and data_ptr / _version / shape identify them)
    gen = param_state.generation() if weight.requires_grad else -1
    key = (weight.data_ptr(), weight._version, gen, tuple(weight.shape), weight.device)
    hit = getattr(module, "_padded_patch_weight", None)
    if hit is None or hit[0] != key:
        hit = (key, padded_patch_weight(weight))
        module._padded_patch_weight = hit
    return hit[1]


def patch_embed(x, weight, bias, pos=None, prefix=None, padded_weight=None):
    """out[:, P:, :] = conv2d(x, weight, bias, stride=patch).flatten(2).T + pos; out[:, :P] = prefix tokens.
    x (B,3,H,W) bf16; weight (N,3,p,p); bias (N) or None; pos (M,N) or None; prefix (P,N) or None (class / register
    tokens, copied as they are).  Forward only (the towers are frozen, src/model.py:349,368)."""
    B, _, H, W = x.shape
    N, patch = weight.shape[0], weight.shape[-1]
    M = (H // patch) * (W // patch)
    npre = 0 if prefix is None else prefix.shape[0]
    x = x.contiguous()
    out = torch.empty(B, npre + M, N, dtype=torch.bfloat16, device=x.device)
    if npre:
        out[:, :npre] = prefix.to(torch.bfloat16)
    wp = padded_weight if padded_weight is not None else padded_patch_weight(weight)
    bias = None if bias is None else bias.detach().to(torch.bfloat16).contiguous()
    pos = None if pos is None else pos.detach().to(torch.bfloat16).reshape(M, N).contiguous()
    rc = hip_lib.load().vrwkv_patch_embed_bf16(B, H, W, patch, N, _p(x), _p(wp), None if bias is None else _p(bias),
                                               None if pos is None else _p(pos), _p(out), npre + M, npre,
                                               hip_lib.launch_stream(x.device))
    hip_lib.check(rc, "vrwkv_patch_embed_bf16")
    return out
