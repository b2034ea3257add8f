"""Hardware check of the MFMA lane->element maps and cross-lane primitives that
visualrwkv_amd/csrc/gfx950_prims.h documents and tests/emu/gfx950_prims.h models."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _probe(lib, which, a, b, dshape):
    d = torch.zeros(dshape, device="cuda")
    rc = lib.vrwkv_debug_probe(which, a.data_ptr(), b.data_ptr() if b is not None else 0, d.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    return d


@pytest.mark.parametrize("which,M,N,K,bf", [(0, 16, 16, 4, False), (1, 32, 32, 2, False), (2, 16, 16, 32, True), (3, 32, 32, 16, True),
                                              (5, 16, 16, 16, True)])
def test_mfma_maps(hip_lib, which, M, N, K, bf):
    g = torch.Generator().manual_seed(which)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)          # asymmetric operands: a transposed map cannot pass
    if bf:
        a, b = a.bfloat16().float(), b.bfloat16().float()
    d = _probe(hip_lib, which, a.cuda(), b.cuda(), (M, N)).cpu()
    ref = a.double() @ b.double()
    assert torch.allclose(d.double(), ref, rtol=1e-5, atol=1e-5)


def test_lane_primitives(hip_lib):
    x = torch.arange(64, dtype=torch.float32) * 1.5 + 1
    d = _probe(hip_lib, 4, x.cuda(), None, (896,)).cpu().view(14, 64)
    lanes = torch.arange(64)
    assert torch.equal(d[0], x.view(4, 16).sum(1, keepdim=True).expand(4, 16).reshape(64))
    assert torch.equal(d[1], x[(lanes & ~7) | (7 - (lanes & 7))])
    assert torch.equal(d[2], x[(lanes & ~15) | (15 - (lanes & 15))])
    assert torch.equal(d[3], x[lanes ^ 16])
    assert torch.equal(d[4], x[lanes ^ 32])
    assert torch.allclose(d[5], x.sum().expand(64))
    assert torch.equal(d[6], x[lanes ^ 16]) and torch.equal(d[7], x[lanes ^ 32])          # permlane16/32_swap forms
    q = lanes & 3
    assert torch.equal(d[8], x[(lanes & ~3) | torch.tensor([0, 0, 1, 2])[q]])             # quad_perm [0,0,1,2]
    assert torch.equal(d[9], x[(lanes & ~3) | torch.tensor([1, 2, 3, 3])[q]])             # quad_perm [1,2,3,3]
    for k in range(4):   # quad_transpose: out[k] of lane m = in[m] of lane k of the quad
        assert torch.equal(d[10 + k], x[(lanes & ~3) | k] + 100.0 * q)


def test_mfma_issue_rate_report(hip_lib):
    """Not a pass/fail property: prints cycles per MFMA (1/4/8 independent chains) for the K=32 and K=16 bf16 forms."""
    d = torch.zeros(8, device="cuda")
    rc = hip_lib.vrwkv_debug_probe(6, d.data_ptr(), 0, d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    r = d.cpu().tolist()
    print("MFMA_RATE cycles/instr: x32 chains1 %.1f chains4 %.1f chains8 %.1f | x16 chains1 %.1f chains4 %.1f chains8 %.1f"
          % (r[0], r[1], r[5], r[2], r[3], r[4]))
    assert all(x > 0 for x in r[:6])


def test_lds_transpose_read(hip_lib):
    """ds_read_b64_tr_b16 as documented in gfx950_prims.h (and modelled by the emulator): lane l of a 16-lane group
    pointing at row l>>2, columns 4(l&3).. of a 4x16 block receives column l of that block."""
    img = torch.arange(64 * 80, dtype=torch.float32).view(64, 80)
    d = _probe(hip_lib, 7, img.cuda(), None, (256,)).cpu().view(64, 4)
    for l in range(64):
        g, i = l >> 4, l & 15
        expect = [float(img[4 * g + e, i]) for e in range(4)]
        assert d[l].tolist() == expect, (l, d[l].tolist(), expect)


def test_lds_dma_immediate_offset_moves_both_addresses(hip_lib):
    """global_load_lds_dwordx4 ... offset:IMM as lds_dma16_lean<IMM> documents (and the emulator models) it: lane l lands at
    M0 + IMM + 16 l with the bytes of base + lane_offset + IMM -- the immediate is added to the global AND the LDS address."""
    src = torch.arange(1024, dtype=torch.float32)
    d = _probe(hip_lib, 8, src.cuda(), None, (1024,)).cpu()
    expect = torch.full((1024,), -1.0)
    expect[256:512] = src[256:512]            # IMM = 1024 bytes = 256 floats on both sides, 64 lanes x 4 floats
    assert torch.equal(d, expect)
