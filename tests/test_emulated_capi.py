"""The WKV7 C entry points themselves (csrc/wkv7_capi.hip compiled whole for the host emulator): what the GPU parity tests call
through the product library, called here on CPU tensors -- the launchers' kernel selection (few heads: two workgroups per head in the
forward; the backward's default and variant 5; batch slices above the 32-bit offset limit), the argument checks, and the kernels behind them against the C oracle."""
import ctypes

import pytest
import torch

from oracle import wkv7_c
from oracle.wkv7_oracle import make_inputs, rel_rms

I, VP = ctypes.c_int, ctypes.c_void_p
TOL = 1e-3          # north_star: 1e-3 against the oracle rounded to bf16 the same way


def P(t):
    return VP(t.data_ptr()) if t is not None else VP(0)


def _fwd(lib, w, q, k, v, z, a):
    B, T, H, N = w.shape
    y = torch.zeros_like(v)
    s = torch.zeros(B, H, T // 16, N, N)
    sa = torch.zeros(B, T, H, N)
    lib.vrwkv_wkv7_forward_bf16.argtypes = [I, I, I] + [VP] * 10
    rc = lib.vrwkv_wkv7_forward_bf16(B, T, H, *[P(x) for x in (w, q, k, v, z, a, y, s, sa)], None)
    assert rc == 0, rc
    return y, s, sa


def _bwd(lib, w, q, k, v, z, a, dy, s, sa):
    B, T, H, N = w.shape
    outs = [torch.zeros_like(w) for _ in range(6)]
    lib.vrwkv_wkv7_backward_bf16.argtypes = [I, I, I] + [VP] * 16
    rc = lib.vrwkv_wkv7_backward_bf16(B, T, H, *[P(x) for x in (w, q, k, v, z, a, dy, s, sa)], *[P(o) for o in outs], None)
    assert rc == 0, rc
    return outs


@pytest.mark.parametrize("B,T,H", [(1, 16, 1), (2, 48, 2)])
def test_forward_and_backward_through_the_c_entries(emu_lib, B, T, H):
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=B * 7 + T)
    yr, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
    y, s, sa = _fwd(emu_lib, w, q, k, v, z, a)                     # B * H <= 128: the launcher picks two workgroups per head
    assert rel_rms(y.float(), yr.float()) < TOL
    assert rel_rms(s, sr) < 2e-5 and rel_rms(sa, sar) < 2e-5
    ref = wkv7_c.backward(w, q, k, v, z, a, dy, sr, sar)
    for variant in (-1, 5):                                          # default = 8 at these sizes (wkv7_bwd_v8.h), 5 = wkv7_bwd_v5.h
        assert emu_lib.vrwkv_wkv7_set_backward_variant(variant) == 0
        try:
            outs = _bwd(emu_lib, w, q, k, v, z, a, dy, sr, sar)
        finally:
            emu_lib.vrwkv_wkv7_set_backward_variant(-1)
        for n, o, r in zip(("dw", "dq", "dk", "dv", "dz", "da"), outs, ref):
            assert rel_rms(o.float(), r.float()) < TOL, (variant, n)


def test_backward_batch_slices_on_the_host(emu_lib):
    """The launcher's batch slicing (tensors of 4 GiB and more in production; the limit lowered here): 3 samples as slices of 2 + 1 are
    bit-identical to one launch, a single sample above the limit goes to the 64-bit kernel (variant 5)."""
    B, T, H = 3, 32, 2
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=91)
    _, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
    ref = wkv7_c.backward(w, q, k, v, z, a, dy, sr, sar)
    per_sample = T * H * 64 * 4
    emu_lib.vrwkv_wkv7_set_backward_slice_limit.argtypes = [ctypes.c_ulonglong]
    try:
        base = _bwd(emu_lib, w, q, k, v, z, a, dy, sr, sar)
        assert emu_lib.vrwkv_wkv7_set_backward_slice_limit(2 * per_sample + 1) == 0
        sliced = _bwd(emu_lib, w, q, k, v, z, a, dy, sr, sar)
        assert emu_lib.vrwkv_wkv7_last_variant(1) == 8
        for o, r in zip(sliced, base):
            assert torch.equal(o, r)
        assert emu_lib.vrwkv_wkv7_set_backward_slice_limit(per_sample // 2) == 0
        wide = _bwd(emu_lib, w, q, k, v, z, a, dy, sr, sar)
        assert emu_lib.vrwkv_wkv7_last_variant(1) == 5
        for o, r in zip(wide, ref):
            assert rel_rms(o.float(), r.float()) < TOL
    finally:
        emu_lib.vrwkv_wkv7_set_backward_slice_limit(0)


def test_argument_checks_of_the_launchers(emu_lib):
    w, q, k, v, z, a, dy = make_inputs(1, 16, 1, seed=3)
    y, s, sa = torch.zeros_like(v), torch.zeros(1, 1, 1, 64, 64), torch.zeros(1, 16, 1, 64)
    f = emu_lib.vrwkv_wkv7_forward_bf16
    f.argtypes = [I, I, I] + [VP] * 10
    assert f(1, 15, 1, *[P(x) for x in (w, q, k, v, z, a, y, s, sa)], None) != 0            # T % 16
    assert f(1, 16, 1, P(w), None, *[P(x) for x in (k, v, z, a, y, s, sa)], None) != 0      # null pointer
    assert emu_lib.vrwkv_wkv7_set_backward_variant(4) != 0                                   # dropped generations
    for v in (6, 7, 10, 11, 61, 81):                                                            # A/B partners live in benchmarks/experiments, not in the product launcher
        assert emu_lib.vrwkv_wkv7_set_backward_variant(v) != 0
    assert emu_lib.vrwkv_wkv7_last_variant(0) in (0, 4, 6, 7) and emu_lib.vrwkv_wkv7_last_variant(1) in (0, 5, 8, 9)
    assert emu_lib.vrwkv_wkv7_set_backward_variant(-1) == 0
