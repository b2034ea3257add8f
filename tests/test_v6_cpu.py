"""RWKV-6 row (BASELINE config 4) on CPU: the WKV6 oracle against the reference's algorithm, and the host-side module
mirrors against fixtures recorded from the reference's own Python (tests/golden/make_golden_v6.py)."""
import os
from types import SimpleNamespace

import pytest
import torch

from oracle.wkv6_oracle import make_inputs6, wkv6_autograd, wkv6_backward_ref, wkv6_chunked, wkv6_naive
from oracle.wkv7_oracle import rel_rms

GOLD = os.path.join(os.path.dirname(__file__), "golden", "v6_ref.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


NAIVE_GOLD = os.path.join(os.path.dirname(__file__), "golden", "wkv6_naive_ref.pt")


def test_oracle_reproduces_the_references_own_recurrence():
    """PIN: tests/golden/wkv6_naive_ref.pt holds what the reference's pure-PyTorch `naive_recurrent_rwkv6_fla`
    (VisualRWKV-v6/v6.xx/test_kernel.py:175-215, cross-checked against app/modeling_rwkv.py:891-897) computes in fp64 on the
    reference test's own input distributions, with autograd gradients of the reference's LOSS.  The oracle must give the
    same numbers to fp64 rounding: output, final state (the reference keeps it as [key][value]), all five gradients,
    with and without an initial state."""
    g = torch.load(NAIVE_GOLD)
    assert "naive_recurrent_rwkv6_fla" in g["provenance"]
    B, T, H, N = g["B"], g["T"], g["H"], g["N"]
    f = lambda x: x.view(B, T, H, N)
    for tag in ("zero_state", "with_state"):
        c = g[tag]
        ins = [x.clone().requires_grad_(True) for x in (f(g["r"]), f(g["k"]), f(g["v"]), f(g["w"]), g["u"])]
        s0 = c["s0"].transpose(-1, -2).clone().requires_grad_(True) if c["s0"] is not None else None   # oracle state: [value][key]
        y, S = wkv6_naive(*ins, state0=s0)
        y.backward(f(c["gy"]))
        assert rel_rms(y.detach().reshape(B, T, H * N), c["y"]) < 1e-12
        assert rel_rms(S.detach().transpose(-1, -2), c["final_state"]) < 1e-12
        for x, n in zip(ins, ("gr", "gk", "gv", "gw", "gu")):
            assert rel_rms(x.grad.reshape(c[n].shape), c[n]) < 1e-12, (tag, n)
        if s0 is not None:
            # (the reference adds the initial state into an fp32 buffer, `h += initial_state`: its gradient is fp32-rounded)
            assert rel_rms(s0.grad.transpose(-1, -2), c["gs0"]) < 1e-7


def test_literal_backward_kernels_equal_autograd():
    """kernel_backward_111 / _222 (wkv6_cuda.cu:64-227) restated literally == autograd through the forward recurrence."""
    B, T, H, N = 2, 21, 2, 8
    r, k, v, w, u, gy = (x.double() for x in make_inputs6(B, T, H, N, dtype=torch.float64))
    _, (gr, gk, gv, gw, gu) = wkv6_autograd(r, k, v, w, u, gy)
    ew = -torch.exp(w)
    gu_sum = torch.zeros(H, N, dtype=torch.float64)
    for b in range(B):
        for h in range(H):
            lr, lk, lv, lw, lu = wkv6_backward_ref(r[b, :, h], k[b, :, h], v[b, :, h], ew[b, :, h], u[h], gy[b, :, h])
            for a, ref in ((lr, gr), (lk, gk), (lv, gv), (lw, gw)):
                assert rel_rms(a, ref[b, :, h]) < 1e-12
            gu_sum[h] += lu
            assert float(lw[0].abs().max()) == 0.0 and float(lw[-1].abs().max()) == 0.0     # :201,226
    assert rel_rms(gu_sum, gu) < 1e-12


@pytest.mark.parametrize("T", [16, 37, 64])
def test_chunked_formulation_equals_recurrence(T):
    B, H, N = 2, 2, 16
    r, k, v, w, u, gy = (x.double() for x in make_inputs6(B, T, H, N, seed=T, dtype=torch.float64))
    y, (gr, gk, gv, gw, gu) = wkv6_autograd(r, k, v, w, u, gy)
    ew = -torch.exp(w)
    yc, S, cr, ck, cv, cx, cu = wkv6_chunked(r, k, v, ew, u, gy)
    _, S_ref = wkv6_naive(r, k, v, w, u)
    for a, ref in ((yc, y), (cr, gr), (ck, gk), (cv, gv), (cx * ew, gw), (cu.sum(0), gu)):
        assert rel_rms(a, ref) < 1e-12
    if T % 16 == 0:
        assert rel_rms(S, S_ref) < 1e-12


def test_oracle_reproduces_op_fixture(gold):
    """The fixture was produced through the reference's WKV_6 / RUN_CUDA_RWKV6 wrappers (ew = -exp(w.float()), bf16
    outputs, gu summed over the batch): the oracle alone, fed the same inputs, must give the same tensors."""
    op = gold["op"]
    B, T, C = op["r"].shape
    H = op["u"].shape[0]
    f = lambda x: x.view(B, T, H, C // H)
    y, (gr, gk, gv, gw, gu) = wkv6_autograd(f(op["r"]), f(op["k"]), f(op["v"]), f(op["w"]), op["u"], f(op["gy"]))
    assert rel_rms(y.reshape(B, T, C), op["y"].double()) < 4e-3
    for a, n in ((gr, "gr"), (gk, "gk"), (gv, "gv"), (gw, "gw")):
        assert rel_rms(a.reshape(B, T, C), op[n].double()) < 4e-3, n
    assert rel_rms(gu, op["gu"].double()) < 8e-3


def _mine(gold):
    from visualrwkv_amd.rwkv6 import RWKV_CMix_x060, RWKV_Tmix_x060
    args = SimpleNamespace(**gold["args"])
    return RWKV_Tmix_x060(args, gold["layer_id"]), RWKV_CMix_x060(args, gold["layer_id"])


def test_state_dict_keys_shapes_and_deterministic_initialisers(gold):
    tmix, cmix = _mine(gold)
    for mine, ref in ((tmix.state_dict(), gold["tmix_state"]), (cmix.state_dict(), gold["cmix_state"])):
        assert list(mine.keys()) == list(ref.keys())
        for k in ref:
            assert mine[k].shape == ref[k].shape, k
    for k in ("time_maa_x", "time_maa_w", "time_maa_k", "time_maa_v", "time_maa_r", "time_maa_g", "time_decay", "time_faaaa"):
        assert torch.allclose(tmix.state_dict()[k], gold["tmix_state"][k], rtol=1e-6, atol=1e-7), k
    for k in ("time_maa_k", "time_maa_r"):
        assert torch.allclose(cmix.state_dict()[k], gold["cmix_state"][k], rtol=1e-6, atol=1e-7), k


def test_cmix_fp32(gold):
    _, cmix = _mine(gold)
    cmix.load_state_dict(gold["cmix_state"])
    with torch.no_grad():
        assert rel_rms(cmix(gold["x"]), gold["cmix_y_fp32"]) < 1e-6


def test_tmix_bf16_with_oracle_op(gold):
    """The mirrored time-mix around an oracle-backed op (tests may use the oracle; the product op is HIP only)."""
    tmix, _ = _mine(gold)
    tmix.load_state_dict(gold["tmix_state"])
    tmix = tmix.bfloat16()

    class Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, r, k, v, w, u):
            ctx.save_for_backward(r, k, v, w, u)
            B, T, C = r.shape
            H = u.shape[0]
            f = lambda x: x.float().view(B, T, H, C // H)
            return wkv6_naive(f(r), f(k), f(v), f(w), u.float())[0].reshape(B, T, C).to(r.dtype)

        @staticmethod
        def backward(ctx, gy):
            r, k, v, w, u = ctx.saved_tensors
            B, T, C = r.shape
            H = u.shape[0]
            f = lambda x: x.view(B, T, H, C // H)
            with torch.enable_grad():
                _, g = wkv6_autograd(f(r), f(k), f(v), f(w), u, f(gy))
            return tuple(x.reshape(B, T, C).to(r.dtype) for x in g[:4]) + (g[4].to(u.dtype),)

    x = gold["x"].bfloat16().requires_grad_(True)
    y = tmix(x, wkv=lambda B, T, C, H, r, k, v, w, u: Op.apply(r, k, v, w, u))
    y.backward(gold["tmix_gy"])
    assert rel_rms(y.float(), gold["tmix_y_bf16"].float()) < 1e-2
    assert rel_rms(x.grad.float(), gold["tmix_gx_bf16"].float()) < 2e-2
    named = dict(tmix.named_parameters())
    for k, g in gold["tmix_grads_bf16"].items():
        assert rel_rms(named[k].grad.float(), g.float()) < 3e-2, k


def test_wkv6_op_has_no_cpu_path():
    from visualrwkv_amd import wkv6
    r = torch.zeros(1, 16, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        wkv6.RUN_CUDA_RWKV6(1, 16, 64, 1, r, r, r, r, torch.zeros(1, 64, dtype=torch.bfloat16))


def _oracle_wkv():
    """An oracle-backed stand-in for RUN_CUDA_RWKV6 (tests may use the oracle; the product op is HIP only)."""
    class Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, r, k, v, w, u):
            ctx.save_for_backward(r, k, v, w, u)
            B, T, C = r.shape
            H = u.shape[0]
            f = lambda x: x.float().view(B, T, H, C // H)
            return wkv6_naive(f(r), f(k), f(v), f(w), u.float())[0].reshape(B, T, C).to(r.dtype)

        @staticmethod
        def backward(ctx, gy):
            r, k, v, w, u = ctx.saved_tensors
            B, T, C = r.shape
            H = u.shape[0]
            f = lambda x: x.view(B, T, H, C // H)
            with torch.enable_grad():
                _, g = wkv6_autograd(f(r), f(k), f(v), f(w), u, f(gy))
            return tuple(x.reshape(B, T, C).to(r.dtype) for x in g[:4]) + (g[4].to(u.dtype),)
    return lambda B, T, C, H, r, k, v, w, u: Op.apply(r, k, v, w, u)


def _visual(gold):
    import transformers
    from visualrwkv_amd import visual6
    vis = gold["visual"]
    args = SimpleNamespace(**vis["args"])
    clip = transformers.CLIPVisionModel(transformers.CLIPVisionConfig(**{k: v for k, v in vis["clip"].items()
                                                                         if k in ("hidden_size", "intermediate_size", "num_hidden_layers",
                                                                                  "num_attention_heads", "image_size", "patch_size")}))
    m = visual6.VisualRWKV6(args, clip, clip.config.hidden_size)
    assert vis["image_token_index"] == visual6.IMAGE_TOKEN_INDEX
    return m, vis


def test_visualrwkv6_state_dict_and_grid_pooling(gold):
    m, vis = _visual(gold)
    assert list(m.state_dict().keys()) == list(vis["state_fp32"].keys())
    m.load_state_dict(vis["state_fp32"])
    for gs, ref in vis["grid_pooling"].items():
        m.args.grid_size = gs
        assert torch.equal(m.grid_pooling(vis["clip_features"]), ref), gs


def test_visualrwkv6_embedding_assembly_and_training_step(gold):
    """preparing_embedding (left-padded first text part | image span | rest), the bidirectional pass and the loss,
    against the reference's bf16 run."""
    m, vis = _visual(gold)
    m.load_state_dict(vis["state_fp32"])
    m = m.bfloat16()
    samples = {"input_ids": vis["input_ids"], "labels": vis["labels"], "images": vis["images"].bfloat16()}
    x, tg, _ = m.preparing_embedding(samples)
    assert torch.equal(tg, vis["targets"]) and (m.img_start, m.img_end) == tuple(vis["img_span"])
    assert x.shape == vis["embeds_bf16"].shape and rel_rms(x.float(), vis["embeds_bf16"].float()) < 1e-2
    wkv = _oracle_wkv()
    loss = m.training_step(samples, 0, wkv)
    loss.backward()
    assert abs(float(loss) - vis["loss"]) < 2e-2 * abs(vis["loss"])
    with torch.no_grad():
        logits, _ = m(samples, wkv)
    assert rel_rms(logits.float(), vis["logits_bf16"].float()) < 2e-2
    assert rel_rms(m.proj.weight.grad.float(), vis["grad_proj"].float()) < 5e-2
    assert rel_rms(m.rwkv.head.weight.grad.float(), vis["grad_head"].float()) < 5e-2
