"""The WKV6 HIP kernels (csrc/wkv6_chunked.h) run lane-exactly on the host emulator against the oracle."""
import ctypes

import pytest
import torch

from oracle.wkv6_oracle import make_inputs6, wkv6_autograd, wkv6_naive
from oracle.wkv7_oracle import rel_rms


def P(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,T,H", [(1, 16, 1), (2, 37, 2), (1, 64, 1)])
def test_forward(emu_lib, B, T, H):
    r, k, v, w, u, _ = make_inputs6(B, T, H, seed=T)
    ew = (-torch.exp(w.float())).contiguous()
    y = torch.zeros_like(r)
    nch = (T + 15) // 16
    s = torch.zeros(B * H, nch, 64, 64)
    emu_lib.emu_wkv6_forward(B, T, H, P(r), P(k), P(v), P(ew), P(u), P(y), P(s))
    y_ref, _ = wkv6_naive(r.double(), k.double(), v.double(), w.double(), u.double())
    assert rel_rms(y.double(), y_ref) < 3e-3                       # one bf16 rounding of the output
    # chunk-start checkpoints hold S^T of the naive recurrence
    for c in range(nch):
        _, S = wkv6_naive(r[:, :16 * c].double(), k[:, :16 * c].double(), v[:, :16 * c].double(), w[:, :16 * c].double(),
                          u.double()) if c else (None, torch.zeros(B, H, 64, 64, dtype=torch.float64))
        got = s.view(B, H, nch, 64, 64)[:, :, c].transpose(-1, -2).double()
        assert rel_rms(got, S) < 2e-5 if c else float(got.abs().max()) == 0.0


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("B,T,H", [(1, 16, 1), (2, 37, 2), (1, 80, 1)])
def test_backward(emu_lib, B, T, H, variant):
    r, k, v, w, u, gy = make_inputs6(B, T, H, seed=100 + T)
    ew = (-torch.exp(w.float())).contiguous()
    y = torch.zeros_like(r)
    nch = (T + 15) // 16
    s = torch.zeros(B * H, nch, 64, 64)
    emu_lib.emu_wkv6_forward(B, T, H, P(r), P(k), P(v), P(ew), P(u), P(y), P(s))
    outs = [torch.zeros_like(r) for _ in range(4)]
    gu = torch.zeros(B, H * 64, dtype=torch.bfloat16)
    entry = emu_lib.emu_wkv6_backward if variant == 1 else emu_lib.emu_wkv6_backward_v2      # four waves | three-role pipeline of twelve
    lds = entry(B, T, H, P(r), P(k), P(v), P(ew), P(u), P(gy), P(s), *[P(o) for o in outs], P(gu))
    assert 0 < lds <= (64 if variant == 1 else 160) * 1024
    _, (gr, gk, gv, gw, gu_ref) = wkv6_autograd(r, k, v, w, u, gy)
    for name, o, ref in zip(["gr", "gk", "gv", "gw"], outs, (gr, gk, gv, gw)):
        assert rel_rms(o.double(), ref) < 4e-3, name                  # one bf16 rounding
    assert rel_rms(gu.double().sum(0).view(H, 64), gu_ref) < 8e-3     # per-sample bf16 rows, summed as the reference does
