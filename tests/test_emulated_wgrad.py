"""The skinny weight-gradient kernel (csrc/lora_wgrad.h: transposing LDS reads feeding the MFMA, two register stages,
M-slices + reduction) run lane-exactly on the host emulator against an fp64 product."""
import ctypes

import pytest
import torch


def P(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("M,Nw,D,S,transposed", [(32, 128, 32, 1, 0), (75, 128, 96, 2, 0), (200, 256, 32, 3, 1), (161, 128, 96, 5, 1), (70, 256, 64, 2, 0)])
def test_wgrad_skinny(emu_lib, M, Nw, D, S, transposed):
    g = torch.Generator().manual_seed(M + D)
    wide = torch.randn(M, Nw, generator=g).bfloat16()
    narrow = torch.randn(M, D, generator=g).bfloat16()
    part = torch.full((S, Nw, D), float("nan"))
    out = torch.zeros((D, Nw) if transposed else (Nw, D), dtype=torch.bfloat16)
    emu_lib.emu_wgrad_skinny.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    assert emu_lib.emu_wgrad_skinny(M, Nw, D, S, P(wide), P(narrow), P(part), P(out), transposed) == 0
    ref = wide.double().t() @ narrow.double()
    assert torch.allclose(part.sum(0).double(), ref, rtol=1e-5, atol=1e-4)          # fp32 accumulation of exact bf16 products
    want = (ref.t() if transposed else ref).float().bfloat16()
    assert (out.float() - want.float()).abs().max() <= 2 * want.float().abs().max() * 2 ** -8


@pytest.mark.parametrize("M,N1,N2,S", [(32, 256, 256, 1), (96, 256, 512, 1), (160, 512, 256, 2), (224, 256, 256, 3)])
def test_big_weight_gradient_kernel(emu_lib, M, N1, N2, S):
    """csrc/wgrad_big.h on the host emulator: the LDS-DMA tile images (slots XOR-ed with 4 (row & 3) on the source side), the
    transposing operand reads of v_mfma_f32_32x32x16_bf16, the three-stage ring for 1 .. 7 stages, split-K partials + reduce --
    lane-exact index math against fp32 A^T B."""
    g = torch.Generator().manual_seed(M + N1 + S)
    A = (torch.randn(M, N1, generator=g) * 0.5).bfloat16()
    B = (torch.randn(M, N2, generator=g) * 0.5).bfloat16()
    out = torch.zeros(N1, N2, dtype=torch.bfloat16)
    part = torch.zeros(max(S, 1), N1, N2, dtype=torch.float32)
    emu_lib.emu_wgrad_big.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    lds = emu_lib.emu_wgrad_big(M, N1, N2, S, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(part.data_ptr()),
                                ctypes.c_void_p(out.data_ptr()))
    assert 0 < lds <= 160 * 1024
    ref = A.float().t() @ B.float()
    assert torch.equal(out.float(), ref.bfloat16().float()) or float((out.float() - ref).norm() / ref.norm()) < 2e-3

