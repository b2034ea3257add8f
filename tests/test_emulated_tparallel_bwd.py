"""Sequence-parallel WKV7 backward (bwd_kernel_v5<.., TPAR>) on the host emulator: two passes over the segments with a
64x64 scan in between reproduce the gradients of the plain chunk-sequential backward kernel."""
import ctypes

import pytest
import torch

from oracle.wkv7_oracle import make_inputs, rel_rms


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("T,H,nseg", [(96, 2, 3), (64, 1, 4), (80, 1, 2)])
def test_two_pass_segments_equal_sequential_backward(emu_lib, T, H, nseg):
    B = 1
    segments = emu_lib.emu_wkv7_backward_segments_v5
    seq_mode = 6
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=T + nseg)
    y = torch.zeros_like(v)
    nch = T // 16
    s = torch.zeros(B, H, nch, 64, 64)
    sa = torch.zeros(B, T, H, 64)
    emu_lib.emu_wkv7_forward(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y), P(s), P(sa), -1)
    ref = [torch.zeros_like(w) for _ in range(6)]
    emu_lib.emu_wkv7_backward_chunked(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa), *[P(t) for t in ref], seq_mode)

    # forward maps of the segments: S_end = S_start M + ..  (forward from S = I with v = 0)
    bounds = [nch * p // nseg * 16 for p in range(nseg + 1)]
    eye = torch.eye(64).expand(B, H, 64, 64).contiguous()
    M = []
    for p in range(nseg):
        sl = slice(bounds[p], bounds[p + 1])
        ops = [t[:, sl].contiguous() for t in (w, q, k, torch.zeros_like(v), z, a)]
        out, y_seg = torch.zeros(B, H, 64, 64), torch.zeros_like(ops[0])
        emu_lib.emu_wkv7_forward_state(B, bounds[p + 1] - bounds[p], H, *[P(t) for t in ops], P(y_seg), P(eye), P(out))
        M.append(out)

    def run(ds_in):
        ds_out = torch.zeros(B, H, nseg, 64, 64)
        g = [torch.zeros_like(w) for _ in range(6)]
        segments(B, T, H, nseg, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa), P(ds_in), P(ds_out), *[P(t) for t in g])
        return ds_out, g

    C, _ = run(None)                                             # pass 1: dS at the segment starts from dS_end = 0
    ds_end = torch.zeros(B, H, nseg, 64, 64)
    for p in range(nseg - 1, 0, -1):                             # dS_start(p) = dS_end(p) M_p^T + C_p  is  dS_end(p-1)
        ds_end[:, :, p - 1] = ds_end[:, :, p] @ M[p].transpose(-1, -2) + C[:, :, p]
    _, got = run(ds_end.contiguous())                            # pass 2: the gradients
    for name, a_, b_ in zip(("dw", "dq", "dk", "dv", "dz", "da"), got, ref):
        assert rel_rms(a_.float(), b_.float()) < 2e-3, name      # bf16 outputs; state products in bf16x3


def test_host_function_on_the_emulator(emu_lib, monkeypatch):
    """visualrwkv_amd.wkv7.wkv7_backward_tparallel itself (segment views, M_p, scan, two launches), with the two C-ABI
    entries it calls redirected to the emulated kernels -- the host logic cannot run on a GPU in this suite."""
    import contextlib
    from types import SimpleNamespace

    from visualrwkv_amd import hip_lib, wkv7

    V = lambda p: ctypes.c_void_p(p) if p else None

    class Shim:
        @staticmethod
        def vrwkv_wkv7_forward_state_bf16(B, T, H, w, q, k, v, z, a, y, s0, s_fin, s_ckpt, sa, stream):
            return emu_lib.emu_wkv7_forward_state_train(B, T, H, V(w), V(q), V(k), V(v), V(z), V(a), V(y), V(s0), V(s_fin),
                                                        V(s_ckpt), V(sa))

        @staticmethod
        def vrwkv_wkv7_backward_segments_bf16(B, T, H, P_, w, q, k, v, z, a, dy, s, sa, ds_in, ds_out, dw, dq, dk, dv, dz, da, stream):
            fn = emu_lib.emu_wkv7_backward_segments_v5
            return fn(B, T, H, P_, V(w), V(q), V(k), V(v), V(z), V(a), V(dy), V(s), V(sa), V(ds_in), V(ds_out), V(dw), V(dq), V(dk), V(dv),
                      V(dz), V(da))

        @staticmethod
        def vrwkv_strerror(code):
            return b"emulated"

    monkeypatch.setattr(hip_lib, "load", lambda: Shim)
    monkeypatch.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: SimpleNamespace(cuda_stream=0))
    B, T, H, nseg = 2, 64, 1, 2
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=5)
    y = torch.zeros_like(v)
    s = torch.zeros(B, H, T // 16, 64, 64)
    sa = torch.zeros(B, T, H, 64)
    emu_lib.emu_wkv7_forward(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y), P(s), P(sa), -1)
    ref = [torch.zeros_like(w) for _ in range(6)]
    emu_lib.emu_wkv7_backward_chunked(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa), *[P(t) for t in ref], 6)
    got = wkv7.wkv7_backward_tparallel(w, q, k, v, z, a, dy, s, sa, nseg)
    for name, a_, b_ in zip(("dw", "dq", "dk", "dv", "dz", "da"), got, ref):
        assert rel_rms(a_.float(), b_.float()) < 2e-3, name
    one = wkv7.wkv7_backward_tparallel(w, q, k, v, z, a, dy, s, sa, 1)          # one segment = the sequential kernel
    for a_, b_ in zip(one, ref):
        assert torch.equal(a_, b_)

    # the sequence-parallel forward with the training by-products (segment views + re-ordered checkpoints)
    y_t, fin, s_t, sa_t = wkv7.wkv7_forward_tparallel(w, q, k, v, z, a, segments=nseg, train=True)
    assert rel_rms(y_t.float(), y.float()) < 3e-3
    assert s_t.shape == s.shape and rel_rms(s_t, s) < 1e-4 and rel_rms(sa_t, sa) < 1e-4
    assert rel_rms(fin, s[:, :, -1].transpose(-1, -2)) < 1e-4              # checkpoints hold S^T, the state tensors S
    y_1, _, s_1, sa_1 = wkv7.wkv7_forward_tparallel(w, q, k, v, z, a, segments=1, train=True)
    assert torch.equal(y_1, y) and torch.equal(s_1, s) and torch.equal(sa_1, sa)
