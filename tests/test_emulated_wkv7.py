"""The device kernels' index math, run on the CPU under the lockstep wave64 emulator (tests/emu)
and compared with the oracle.  Catches lane-map / LDS / barrier mistakes without a GPU; the real
parity tests are the -m gpu ones."""
import ctypes

import pytest
import torch

from oracle import wkv7_c
from oracle.wkv7_oracle import make_inputs, rel_rms


def P(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("variant", [-1, 1, 2, 6, 7])   # default (no Ab / Kb images, tr16 reads); 1: round-2 instantiation; 2: no Ab / Kb only; 6: two workgroups per head; 7: wkv7_fwd_v4.h (rows by LDS-DMA in, output images out)
def test_forward_variants(emu_lib, variant):
    B, T, H, N = 2, 32, 2, 64
    w, q, k, v, z, a, _ = make_inputs(B, T, H, seed=variant + 1)
    yr, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
    y = torch.zeros_like(yr); s = torch.zeros_like(sr); sa = torch.zeros_like(sar)
    emu_lib.emu_wkv7_forward(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y), P(s), P(sa), variant)
    assert rel_rms(y.float(), yr.float()) < 1e-3
    assert (y != yr).float().mean() < 0.01
    assert rel_rms(s, sr) < 2e-5 and rel_rms(sa, sar) < 2e-5


@pytest.mark.parametrize("T", [16, 48, 96])
def test_forward_v4_chunk_counts(emu_lib, T):
    """wkv7_fwd_v4.h with 1, 3 and 6 chunks: the V ring (3 slots), the single-buffered staging / output images and their
    hand-off counters all wrap."""
    B, H = 1, 2
    w, q, k, v, z, a, _ = make_inputs(B, T, H, seed=100 + T)
    yr, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
    y = torch.zeros_like(yr); s = torch.zeros_like(sr); sa = torch.zeros_like(sar)
    emu_lib.emu_wkv7_forward(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y), P(s), P(sa), 7)
    assert rel_rms(y.float(), yr.float()) < 1e-3 and (y != yr).float().mean() < 0.01
    assert rel_rms(s, sr) < 2e-5 and rel_rms(sa, sar) < 2e-5


def test_forward_from_state(emu_lib):
    """Inference form of the producer/consumer forward: explicit initial state, final state out, no checkpoints / sa."""
    from oracle.wkv7_oracle import wkv7_naive
    B, T, H = 1, 32, 2
    w, q, k, v, z, a, _ = make_inputs(B, T, H, seed=21)
    g = torch.Generator().manual_seed(4)
    s0 = (torch.randn(B, H, 64, 64, generator=g) * 0.3).contiguous()
    y = torch.zeros_like(v)
    s_fin = torch.zeros(B, H, 64, 64)
    emu_lib.emu_wkv7_forward_state(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y), P(s0), P(s_fin))
    y_ref, s_ref = wkv7_naive(*[x.double() for x in (w, q, k, v, z, a)], state0=s0.double())
    assert rel_rms(y.double(), y_ref) < 4e-3 and rel_rms(s_fin.double(), s_ref) < 2e-5
    y2 = torch.zeros_like(v)
    emu_lib.emu_wkv7_forward_state(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(y2), None, None)
    y_ref0, _ = wkv7_naive(*[x.double() for x in (w, q, k, v, z, a)])
    assert rel_rms(y2.double(), y_ref0) < 4e-3


@pytest.mark.parametrize("mode,T", [(6, 48), (7, 16), (7, 32), (7, 96), (8, 16), (8, 32), (8, 48), (8, 64), (8, 80), (8, 160), (9, 16), (9, 32), (9, 48), (9, 64), (9, 80), (9, 160),
                                    (10, 16), (10, 32), (10, 48), (10, 64), (10, 80), (10, 160),
                                    (11, 16), (11, 32), (11, 48), (11, 64), (11, 80), (11, 160), (12, 16), (12, 32), (12, 48), (12, 64), (12, 80), (12, 160),
                                    (13, 16), (13, 48), (13, 80), (13, 160), (14, 16), (14, 48), (14, 80), (14, 160),
                                    (15, 16), (15, 32), (15, 48), (15, 64), (15, 80), (15, 160), (16, 16), (16, 32), (16, 48), (16, 80), (16, 160),
                                    (17, 16), (17, 32), (17, 48), (17, 64), (17, 80), (17, 160), (18, 16), (18, 32), (18, 48), (18, 64), (18, 80), (18, 160)])
def test_backward_chunked(emu_lib, mode, T):
    """Chunked MFMA backward kernels run lane-exactly on the host: 6 = the producer / consumer schedule (wkv7_bwd_v5.h), 7 = the three-stage wave pipeline (wkv7_bwd_v6.h;
    1, 2 and 6 chunks: pipeline shorter than, equal to and longer than its depth), 8 = the same pipeline with the full-row memory
    role (benchmarks/experiments/wkv7_bwd_v7.h, an A/B partner outside the library: rows by LDS-DMA, single-buffered staging and result images; 1 .. 5 chunks = only ragged steps, 10 chunks =
    five steady-state steps), 9 = wkv7_bwd_v8.h (one copy of dL/dS handed from the I to the J waves as an operand image, the decay-gradient
    term as an MFMA diagonal, the T chain on P wave 0, single-buffered S0), 10 = 9 with the score pieces a step ahead on the P waves, 11 / 12 = 9 / 10 with
    the element-wise tail and the gradient stores on the J waves (JTAIL), 13 / 14 = 10 / 9 with the round-6 LDS layout (OPT = 3: tile-pair reads dealt over
    both halves of their slots, dS image swizzled with row bit 3), 15 / 16 = 10 / 9 with S0 prefetched into the J waves' registers (OPT = 4), 17 = 10 with the five tail outputs stored as whole
    128-byte rows through the `res` slots (OPT = 8192), 18 = 10 with S0 handed back before the J waves' split and requested before the P waves' prepare -- all
    bit-identical to 10 / 9."""
    B, H = 1, 2
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=7 + mode)
    _, s, sa = wkv7_c.forward(w, q, k, v, z, a)
    ref = wkv7_c.backward(w, q, k, v, z, a, dy, s, sa)
    outs = [torch.zeros_like(w) for _ in range(6)]
    lds = emu_lib.emu_wkv7_backward_chunked(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa),
                                            *[P(o) for o in outs], mode)
    assert 0 < lds <= 160 * 1024
    for name, o, r in zip(["dw", "dq", "dk", "dv", "dz", "da"], outs, ref):
        assert rel_rms(o.float(), r.float()) < 1e-3, (name, mode)
    if mode in (13, 14, 15, 16, 17, 18):    # a pure re-addressing of LDS / S0 from memory into the J waves' registers instead of through an LDS image: the same values reach the same MFMAs
        base = [torch.zeros_like(w) for _ in range(6)]
        emu_lib.emu_wkv7_backward_chunked(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa), *[P(o) for o in base], {13: 10, 14: 9, 15: 10, 16: 9, 17: 10, 18: 10}[mode])
        for name, o, r in zip(["dw", "dq", "dk", "dv", "dz", "da"], outs, base):
            assert torch.equal(o, r), (name, mode)


def _sam_bias(q, rel_h, rel_w, S):
    """add_decomposed_rel_pos (src/sam.py:392-426) in fp32: (B, H, L, L) bias from the unscaled q."""
    B, L, H, D = q.shape
    idx = torch.arange(S)
    Rh = rel_h.float()[(idx[:, None] - idx[None, :]) + (S - 1)]
    Rw = rel_w.float()[(idx[:, None] - idx[None, :]) + (S - 1)]
    rq = q.float().reshape(B, S, S, H, D)
    bh = torch.einsum("bhwnc,hkc->bnhwk", rq, Rh)
    bw = torch.einsum("bhwnc,wkc->bnhwk", rq, Rw)
    return (bh[..., :, None] + bw[..., None, :]).reshape(B, H, L, L)


@pytest.mark.parametrize("D,L,qt,S", [(64, 70, 1, 0), (64, 200, 2, 0), (72, 48, 1, 0), (72, 150, 2, 0), (64, 196, 1, 14),
                                      (64, 196, 2, 14), (64, 4096, 2, 64)])
def test_attention_forward(emu_lib, D, L, qt, S):
    """ViT attention kernels (csrc/attention_kernels.h) vs fp32 softmax attention; q/k/v are strided slices of one
    fused qkv tensor as in the towers; L is not a multiple of the key tile (tail masking); S > 0: SAM window with the
    decomposed relative-position bias computed inside the kernel."""
    B, H = 1, (1 if S == 64 else 2)
    g = torch.Generator().manual_seed(D + L)
    qkv = (torch.randn(B, L, 3, H, D, generator=g)).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    o = torch.zeros(B, L, H, D, dtype=torch.bfloat16)
    sb, sl, sh, _ = q.stride()
    bias, rh, rw = None, None, None
    if S:
        rh = (0.3 * torch.randn(2 * S - 1, D, generator=g)).bfloat16()
        rw = (0.3 * torch.randn(2 * S - 1, D, generator=g)).bfloat16()
        bias = _sam_bias(q, rh, rw, S)
    rc = emu_lib.emu_attention_fwd(B, L, H, D, P(q), P(k), P(v), ctypes.c_long(sb), ctypes.c_long(sl), ctypes.c_long(sh), P(o),
                                   qt, S, P(rh) if S else None, P(rw) if S else None)
    assert rc == 0
    ref = torch.nn.functional.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2),
                                                           v.float().transpose(1, 2), attn_mask=bias).transpose(1, 2)
    assert rel_rms(o.float(), ref) < 1e-2


@pytest.mark.parametrize("patch,side,N,prefix", [(14, 112, 32, 5), (16, 128, 64, 0)])
def test_patch_embed(emu_lib, patch, side, N, prefix):
    """Implicit-GEMM patch embedding (csrc/patch_embed_kernels.h) vs conv2d + bias + position embedding in fp32."""
    B = 2
    g = torch.Generator().manual_seed(patch)
    x = torch.randn(B, 3, side, side, generator=g).bfloat16()
    w = (0.05 * torch.randn(N, 3, patch, patch, generator=g)).bfloat16()
    bias = (0.1 * torch.randn(N, generator=g)).bfloat16()
    M = (side // patch) ** 2
    pos = (0.1 * torch.randn(M, N, generator=g)).bfloat16()
    K, KP = 3 * patch * patch, (3 * patch * patch + 31) // 32 * 32
    wp = torch.zeros(N, KP, dtype=torch.bfloat16)
    wp[:, :K] = w.reshape(N, K)
    out = torch.full((B, prefix + M, N), 7.0, dtype=torch.bfloat16)
    rc = emu_lib.emu_patch_embed(B, side, side, patch, N, P(x), P(wp), P(bias), P(pos), P(out), prefix + M, prefix)
    assert rc == 0
    ref = torch.nn.functional.conv2d(x.float(), w.float(), bias.float(), stride=patch).flatten(2).transpose(1, 2) + pos.float()
    assert rel_rms(out[:, prefix:].float(), ref) < 5e-3
    assert torch.all(out[:, :prefix] == 7.0)             # prefix rows untouched
