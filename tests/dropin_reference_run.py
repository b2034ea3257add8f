"""Helper of tests/test_dropin_reference_cpu.py (run as a subprocess, only where /root/reference exists): imports the REFERENCE's own
VisualRWKV-v7/v7.00/src/model.py with `import visualrwkv_amd.wkv7` in place of its JIT build of the CUDA operator -- exactly the patch
INTEGRATION.md section 1 describes -- and runs the reference's RWKV (its WindBackstepping / RUN_CUDA_RWKV7g / RWKV_Tmix_x070 / RWKV_CMix_x070 / Block /
RWKV classes, its code, unchanged) on top of torch.ops.wind_backstepping as this package registers it (CPU key: libvisualrwkv_host.so), next to this
package's mirror of the same classes with the same state dict.  Prints one JSON line.  Stand-ins only for third-party packages that are not installed
(pytorch_lightning, deepspeed, timm behind src.vision); nothing of the reference is stored anywhere."""
import json
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/VisualRWKV-v7/v7.00"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def main():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module
    pl.__version__ = "1.9.5"
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_info = lambda *a, **k: None
    plu.rank_zero_warn = lambda *a, **k: None
    pls = types.ModuleType("pytorch_lightning.strategies")
    pls.DeepSpeedStrategy = type("DeepSpeedStrategy", (), {})
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": plu, "pytorch_lightning.strategies": pls})
    vis = types.ModuleType("src.vision")
    vis.SamDinoSigLIPViTBackbone = type("SamDinoSigLIPViTBackbone", (nn.Module,), {})
    sys.modules["src.vision"] = vis
    os.environ["RWKV_JIT_ON"] = "0"
    os.environ["RWKV_HEAD_SIZE_A"] = "64"
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: None                    # the reference's import-time JIT build of cuda/wkv7_cuda.cu ...
    import visualrwkv_amd.wkv7 as vw                  # ... replaced by this import: registers torch.ops.wind_backstepping
    from src import model as ref                      # the reference module itself, unchanged
    from visualrwkv_amd import rwkv7 as mine

    assert ref.WindBackstepping is not vw.WindBackstepping and ref.RUN_CUDA_RWKV7g is not vw.RUN_CUDA_RWKV7g      # the reference's own autograd surface drives the op
    args = SimpleNamespace(n_embd=128, dim_att=128, n_layer=2, head_size_a=64, head_size_divisor=8, vocab_size=512, dropout=0, grad_cp=0,
                           ctx_len=64, load_model="", my_testing="x070", dim_ffn=512, fused=False)
    torch.manual_seed(3)
    r_model = ref.RWKV(args)
    with torch.no_grad():                             # the reference zero-initialises several projections: make every path live
        g = torch.Generator().manual_seed(5)
        for p in r_model.parameters():
            if float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m_model = mine.RWKV(args)
    missing = m_model.load_state_dict(r_model.state_dict(), strict=True)          # same parameter names and shapes
    r_model, m_model = r_model.bfloat16(), m_model.bfloat16()
    T = 37                                            # not a multiple of 16: the reference left-pads with emb(STOP_TOKEN_INDEX) (src/model.py:286-312)
    x = (torch.randn(2, T, 128, generator=torch.Generator().manual_seed(9)) * 0.5).bfloat16()
    gout = (torch.randn(2, T, 512, generator=torch.Generator().manual_seed(10)) * 0.1).bfloat16()
    res = {}
    for name, model in (("reference", r_model), ("mirror", m_model)):
        xi = x.clone().requires_grad_(True)
        logits = model(xi)
        logits.backward(gout)
        res[name] = (logits.detach().float(), xi.grad.float(), {n: p.grad.float() for n, p in model.named_parameters() if p.grad is not None})
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    lr, xr, gr = res["reference"]
    lm, xm, gm = res["mirror"]
    worst = max(((rel(gm[n], gr[n]), n) for n in gr), default=(0.0, ""))
    print(json.dumps({"ok": True, "logits_shape": list(lr.shape), "logits_rel": rel(lm, lr), "dx_rel": rel(xm, xr), "worst_param_grad_rel": worst[0],
                      "worst_param": worst[1], "n_param_grads": len(gr), "same_grad_keys": sorted(gr) == sorted(gm),
                      "finite": bool(torch.isfinite(lr).all() and torch.isfinite(xr).all())}))


if __name__ == "__main__":
    main()
