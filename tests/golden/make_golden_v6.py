"""Generate tests/golden/v6_ref.pt by importing the reference's own RWKV-6 Python
(VisualRWKV-v6/v6.0/src/model.py: RWKV_Tmix_x060, RWKV_CMix_x060, WKV_6, RUN_CUDA_RWKV6) in this container.

As for the v7 fixtures, third-party packages the image lacks get inert stand-ins and `cpp_extension.load` returns an
object whose forward / backward are the repo's WKV6 oracle (the reference has no CPU kernel; its wrapper asserts bf16,
so the time-mix fixture is a bf16 run).  Only tensors are stored, nothing of the reference's source.

Run where /root/reference exists:   python tests/golden/make_golden_v6.py
"""
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/VisualRWKV-v6/v6.0"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle.wkv6_oracle import wkv6_autograd, wkv6_naive  # noqa: E402


class OracleWkv6Module:
    """Stands in for the JIT-built extension module `wkv6_cuda` (model.py:40): same two entry points."""

    @staticmethod
    def forward(B, T, C, H, r, k, v, ew, u, y):
        w_raw = torch.log(-ew)                                   # the kernels receive ew = -exp(w)
        f = lambda x: x.float().view(B, T, H, C // H)
        out, _ = wkv6_naive(f(r), f(k), f(v), w_raw.view(B, T, H, C // H), u.float().view(H, C // H))
        y.copy_(out.reshape(B, T, C).to(y.dtype))

    @staticmethod
    def backward(B, T, C, H, r, k, v, ew, u, gy, gr, gk, gv, gw, gu):
        N = C // H
        w_raw = torch.log(-ew.double())
        f = lambda x: x.double().view(B, T, H, N)
        for b in range(B):                                       # gu is per sample in the reference (B,C)
          with torch.enable_grad():                              # the reference wrapper calls us under no_grad
            _, g = wkv6_autograd(f(r)[b:b + 1], f(k)[b:b + 1], f(v)[b:b + 1], w_raw.view(B, T, H, N)[b:b + 1],
                                 u.double().view(H, N), f(gy)[b:b + 1])
          for dst, src in zip((gr, gk, gv, gw), g[:4]):
              dst[b].copy_(src.reshape(T, C).to(dst.dtype))
          gu[b].copy_(g[4].reshape(C).to(gu.dtype))


def install_stubs():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    pl.__version__ = "1.9.5"
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_info = lambda *a, **k: None
    plu.rank_zero_only = lambda f: f
    pls = types.ModuleType("pytorch_lightning.strategies")
    pls.DeepSpeedStrategy = type("DeepSpeedStrategy", (), {})
    ds = types.ModuleType("src.dataset")
    ds.IGNORE_INDEX, ds.IMAGE_TOKEN_INDEX = -100, -200
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu, "pytorch_lightning.strategies": pls,
                        "src.dataset": ds})
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: OracleWkv6Module
    os.environ["RWKV_JIT_ON"] = "0"
    os.environ["RWKV_HEAD_SIZE_A"] = "64"
    os.environ["RWKV_CTXLEN"] = "64"


def main():
    install_stubs()
    from src import model as ref
    g = torch.Generator().manual_seed(11)
    args = SimpleNamespace(n_embd=128, dim_att=128, n_layer=4, head_size_a=64, head_size_divisor=8, dim_ffn=448)
    out = {"args": vars(args), "layer_id": 1}
    tmix = ref.RWKV_Tmix_x060(args, 1)
    cmix = ref.RWKV_CMix_x060(args, 1)
    with torch.no_grad():
        for m in (tmix, cmix):
            for p in m.parameters():
                if float(p.abs().sum()) == 0.0:                 # zero-initialised LoRA halves / projections
                    p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        for lin in (tmix.receptance, tmix.key, tmix.value, tmix.output, tmix.gate, cmix.key, cmix.receptance, cmix.value):
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.08)
    out["tmix_state"] = {k: v.detach().clone() for k, v in tmix.state_dict().items()}
    out["cmix_state"] = {k: v.detach().clone() for k, v in cmix.state_dict().items()}
    x = torch.randn(2, 24, 128, generator=g)
    out["x"] = x
    # channel-mix in fp32
    with torch.no_grad():
        out["cmix_y_fp32"] = cmix(x)
    # time-mix in bf16 (the wrapper asserts bf16), forward and backward through the reference's WKV_6
    tb = tmix.bfloat16()
    xb = x.bfloat16().requires_grad_(True)
    y = tb(xb)
    gy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(gy)
    out["tmix_y_bf16"] = y.detach()
    out["tmix_gy"] = gy
    out["tmix_gx_bf16"] = xb.grad.detach()
    out["tmix_grads_bf16"] = {k: p.grad.detach().clone() for k, p in tb.named_parameters()
                              if k in ("time_faaaa", "time_decay", "time_maa_w", "key.weight")}
    # the op itself as the reference wrapper drives it: RUN_CUDA_RWKV6 on random bf16 inputs
    B, T, C, H = 2, 24, 128, 2
    uni = lambda *s, lo=-1.0, hi=1.0: (torch.rand(*s, generator=g) * (hi - lo) + lo).bfloat16()
    r, k, v = (uni(B, T, C).requires_grad_(True) for _ in range(3))
    w = uni(B, T, C, lo=-8.0, hi=1.0).requires_grad_(True)
    u = uni(H, C // H).requires_grad_(True)
    yy = ref.RUN_CUDA_RWKV6(B, T, C, H, r, k, v, w, u)
    gyy = uni(B, T, C)
    yy.backward(gyy)
    # EXPECTED values of the op section: the reference's own pure-PyTorch recurrence (test_kernel.py:175-215,
    # run_naive_recurrent_fla, extracted and executed unmodified in fp64 as in make_golden_wkv6.py) on the same inputs,
    # differentiated by autograd, rounded once to bf16 -- NOT the repository oracle.  What the reference's WKV_6 wrapper
    # computed above with the oracle behind cpp_extension.load must agree with it (wrapper semantics: ew = -exp(w), per-sample
    # gu rows summed), which is asserted here at generation time.
    from make_golden_wkv6 import REF_TEST, _F64, extract_functions
    fns = extract_functions(REF_TEST, ["naive_recurrent_rwkv6_fla", "run_naive_recurrent_fla"])
    ns = {"torch": torch, "Optional": __import__("typing").Optional}
    for name in ("naive_recurrent_rwkv6_fla", "run_naive_recurrent_fla"):
        exec(compile(fns[name], REF_TEST + ":" + name, "exec"), ns)
    leaves = [x.detach().double().requires_grad_(True) for x in (r, k, v, w, u)]
    y64, _ = ns["run_naive_recurrent_fla"](B, T, C, H, *[_F64(x) for x in leaves], None)
    y64 = y64.as_subclass(torch.Tensor)
    (y64 * gyy.double()).sum().backward()
    exp = {"y": y64.detach(), "gr": leaves[0].grad, "gk": leaves[1].grad, "gv": leaves[2].grad, "gw": leaves[3].grad, "gu": leaves[4].grad}
    got = {"y": yy.detach(), "gr": r.grad, "gk": k.grad, "gv": v.grad, "gw": w.grad, "gu": u.grad}
    for n_ in exp:
        e_ = float((got[n_].double() - exp[n_]).norm() / exp[n_].norm())
        assert e_ < (2e-2 if n_ == "gu" else 6e-3), (n_, e_)      # the wrapper path rounds to bf16 (gu: bf16 rows summed in bf16)
    out["op"] = {"r": r.detach(), "k": k.detach(), "v": v.detach(), "w": w.detach(), "u": u.detach(), "gy": gyy,
                 "provenance": "expected values: VisualRWKV-v6/v6.xx/test_kernel.py:175-215 run_naive_recurrent_fla, fp64 autograd, "
                               "rounded once to bf16; the WKV_6 wrapper with the repository oracle behind it agrees (checked at generation)",
                 **{n_: exp[n_].float().bfloat16() for n_ in exp}}
    # ---- VisualRWKV (v6): CLIP tower stand-in of the same class (random tiny config), grid pooling, embedding assembly,
    # bidirectional pass, loss -- all through the reference's own code
    import transformers
    clip_cfg = transformers.CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                             image_size=28, patch_size=7)

    class TinyClip:
        @staticmethod
        def from_pretrained(name):
            torch.manual_seed(5)
            return transformers.CLIPVisionModel(clip_cfg)

    ref.CLIPVisionModel = TinyClip
    vargs = SimpleNamespace(n_embd=128, dim_att=128, n_layer=3, head_size_a=64, head_size_divisor=8, dim_ffn=448,
                            vocab_size=300, dropout=0, grad_cp=0, ctx_len=40, load_model="", vision_tower_name="tiny",
                            grid_size=2)
    torch.manual_seed(6)
    vm = ref.VisualRWKV(vargs)
    with torch.no_grad():
        for p in vm.rwkv.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    ids = torch.randint(0, 300, (3, 24), generator=g)
    ids[0, 5] = ref.IMAGE_TOKEN_INDEX
    ids[1, 2] = ref.IMAGE_TOKEN_INDEX                  # sample 2 has no image
    labels = ids.clone()
    labels[:, :8] = ref.IGNORE_INDEX
    labels[ids == ref.IMAGE_TOKEN_INDEX] = ref.IGNORE_INDEX
    images = torch.randn(3, 1, 3, 28, 28, generator=g)
    out["visual"] = {"args": vars(vargs), "clip": clip_cfg.to_dict(), "state_fp32": {k: v.detach().clone() for k, v in vm.state_dict().items()},
                     "input_ids": ids, "labels": labels, "images": images, "image_token_index": ref.IMAGE_TOKEN_INDEX}
    with torch.no_grad():
        feats = vm.vit(images.view(3, 3, 28, 28)).last_hidden_state
        pooled = {}
        for gs in (-1, 0, 1, 2, 4):
            vm.args.grid_size = gs
            pooled[gs] = vm.grid_pooling(feats).clone()
        vm.args.grid_size = 2
        out["visual"]["clip_features"] = feats
        out["visual"]["grid_pooling"] = pooled
    vb = vm.bfloat16()
    samples = {"input_ids": ids, "labels": labels, "images": images.bfloat16()}
    x, tg, imf = vb.preparing_embedding(samples)
    out["visual"]["embeds_bf16"] = x.detach().clone()
    out["visual"]["targets"] = tg.clone()
    out["visual"]["img_span"] = (int(vb.img_start), int(vb.img_end))
    loss = vb.training_step(samples, 0)
    loss.backward()
    with torch.no_grad():
        logits, _ = vb(samples)
    out["visual"]["logits_bf16"] = logits.detach()
    out["visual"]["loss"] = float(loss)
    out["visual"]["grad_proj"] = vb.proj.weight.grad.detach().clone()
    out["visual"]["grad_head"] = vb.rwkv.head.weight.grad.detach().clone()
    path = os.path.join(HERE, "v6_ref.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
