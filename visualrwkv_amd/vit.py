"""Frozen vision towers that feed the image-to-language projector (forward only).

Mirror of the reference's `SamDinoSigLIPViTBackbone` (VisualRWKV-v7/v7.00/src/vision.py:49-145):
    dino   : timm `vit_large_patch14_reg4_dinov2.lvd142m`  @448 -> 1024 patch tokens x 1024
    siglip : timm `vit_so400m_patch14_siglip_384`          @448 -> 1024 patch tokens x 1152
    sam    : SAM ViT-B image encoder (src/sam.py)          @1024 -> 64x64x256 -> 32x32x1024
each returning the output of its *second-to-last* block without the final norm and without prefix
tokens (`get_intermediate_layers(n={depth-2})`, vision.py:75-81), concatenated on the channel axis.

timm is a third-party dependency of the reference that is neither vendored nor pinned
(Dockerfile:13) and is not installed here, so the two timm towers are re-stated from timm's
published `VisionTransformer` (parameter names kept: patch_embed.proj, pos_embed, cls_token,
reg_token, blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}, norm) and
pinned in tests against `transformers`' Siglip/Dinov2 modules; the SAM tower is pinned against the
in-repo reference implementation (src/sam.py).  Softmax attention runs through
`visualrwkv_amd.attention.attention` (MFMA flash kernel on gfx950).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import gemm_tuning
from .attention import attention, attention_relpos

VIT_STREAMS = os.environ.get("VRWKV_VIT_STREAMS", "1") != "0"      # the towers of SamDinoSigLIPViTBackbone on one HIP stream each


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, act="gelu"):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)
        self.approx = "tanh" if act == "gelu_tanh" else "none"

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled():
            return _mlp_infer(self, x)
        return self.fc2(F.gelu(self.fc1(x), approximate=self.approx))


class _LayerScale(nn.Module):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, L, C = x.shape
        qkv = self.qkv(x).view(B, L, 3, self.num_heads, C // self.num_heads)
        o = attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])          # (B, L, H, D) in and out
        return self.proj(o.reshape(B, L, C))


class _Block(nn.Module):
    def __init__(self, dim, heads, hidden, ls_init, act, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim, heads)
        self.ls1 = _LayerScale(dim, ls_init) if ls_init is not None else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden, act)
        self.ls2 = _LayerScale(dim, ls_init) if ls_init is not None else nn.Identity()

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


def _tower_fused(x, dim):
    """The frozen towers' inference path on the GPU: LayerScale multiply + residual add + LayerNorm in one kernel
    (csrc/ln_fused.hip, vrwkv_add_ln_scaled_fwd_bf16).  bf16 on the device, no autograd, C a multiple of 64."""
    return x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and dim % 64 == 0 and dim <= 8192


def _mlp_infer(mlp, h):
    """fc2(gelu(fc1(h))) of a frozen tower on the device: the bias rides in the library GEMM's epilogue, the activation is one in-place streaming kernel
    (fused.gelu_).  (torch._addmm_activation(use_gelu=True) does not fuse on this ROCm build: it launched the eager GELU kernel -- 118 us per call,
    VALU-bound on erf / tanh -- 49 times per step, profiles/r6y_step_kernel_stats.csv.)"""
    from . import fused
    y = mlp.fc1(h)
    if y.numel() % 8 == 0 and y.is_contiguous():
        return mlp.fc2(fused.gelu_(y, mlp.approx == "tanh"))
    return mlp.fc2(F.gelu(y, approximate=mlp.approx))


def _timm_blocks_infer(blocks, x, last):
    """blocks 0..last of a timm pre-LN stack on the (x, pending delta * gamma) residual stream (same math as _Block.forward
    chained: x = x + ls1(attn(norm1(x))); x = x + ls2(mlp(norm2(x))))."""
    from . import fused
    delta, scale = None, None
    for i, blk in enumerate(blocks):
        g1 = blk.ls1.gamma if isinstance(blk.ls1, _LayerScale) else None
        g2 = blk.ls2.gamma if isinstance(blk.ls2, _LayerScale) else None
        x, h = fused.add_ln_infer(x, delta, scale, blk.norm1)
        x, h = fused.add_ln_infer(x, blk.attn(h), g1, blk.norm2)
        delta, scale = _mlp_infer(blk.mlp, h), g2
        if i == last:
            break
    return x + (delta * scale if scale is not None else delta)


class _PatchEmbed(nn.Module):
    """Non-overlapping patch embedding = one GEMM over unfolded patches (conv with stride = kernel)."""

    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        B, C, H, W = x.shape
        p = self.patch
        gh, gw = H // p, W // p
        x = x.view(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
        return F.linear(x, self.proj.weight.view(self.proj.weight.shape[0], -1), self.proj.bias)

    @property
    def num_patches_side(self):
        return None


class TimmViT(nn.Module):
    """timm VisionTransformer subset: patch embed, learned abs pos-embed, optional cls/register tokens
    (no_embed_class layout of DINOv2-reg), pre-LN blocks with optional LayerScale."""

    def __init__(self, img_size=448, patch=14, dim=1024, depth=24, heads=16, mlp_hidden=4096,
                 class_token=True, reg_tokens=0, ls_init: Optional[float] = None, act="gelu", eps=1e-6):
        super().__init__()
        self.embed_dim = dim
        self.grid = img_size // patch
        self.patch_embed = _PatchEmbed(patch, 3, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim)) if class_token else None
        self.reg_token = nn.Parameter(torch.zeros(1, reg_tokens, dim)) if reg_tokens else None
        self.num_prefix_tokens = (1 if class_token else 0) + reg_tokens
        self.pos_embed = nn.Parameter(torch.randn(1, self.grid * self.grid, dim) * 0.02)
        self.blocks = nn.ModuleList([_Block(dim, heads, mlp_hidden, ls_init, act, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=eps)   # present in checkpoints; not applied (norm=False)

    def get_intermediate_layers(self, x, n: Sequence[int]):
        """Patch tokens after block index max(n) (0-based), prefix tokens removed, no final norm."""
        last = max(n)
        from . import fused
        pe = self.patch_embed
        if (not torch.is_grad_enabled() or not pe.proj.weight.requires_grad) and pe.proj.weight.dtype == torch.bfloat16 \
                and fused.patch_embed_supported(x, pe.patch, self.embed_dim):
            # GPU: implicit GEMM over the NCHW pixels, bias + position embedding in the epilogue, prefix tokens in place
            pre = [t[0] for t in (self.cls_token, self.reg_token) if t is not None]
            x = fused.patch_embed(x, pe.proj.weight, pe.proj.bias, self.pos_embed[0], torch.cat(pre, dim=0) if pre else None,
                                  padded_weight=fused.cached_padded_patch_weight(pe, pe.proj.weight))
        else:
            x = pe(x) + self.pos_embed
            prefix = [t.expand(x.shape[0], -1, -1) for t in (self.cls_token, self.reg_token) if t is not None]
            if prefix:
                x = torch.cat(prefix + [x], dim=1)
        if _tower_fused(x, self.embed_dim) and self.blocks[0].norm1.weight.dtype == torch.bfloat16:
            return _timm_blocks_infer(self.blocks, x, last)[:, self.num_prefix_tokens:]
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i == last:
                break
        return x[:, self.num_prefix_tokens:]

    def forward(self, x):
        return self.get_intermediate_layers(x, n={len(self.blocks) - 2})


def dinov2_large_reg4(img_size=448, depth=24, dim=1024, heads=16):
    return TimmViT(img_size, 14, dim, depth, heads, dim * 4, class_token=True, reg_tokens=4, ls_init=1e-5)


def siglip_so400m(img_size=448, depth=27, dim=1152, heads=16, mlp_hidden=4304):
    return TimmViT(img_size, 14, dim, depth, heads, mlp_hidden, class_token=False, reg_tokens=0, ls_init=None)


# ------------------------------------------------------------------------------------------------
# SAM ViT-B image encoder (src/sam.py:77-181), parameter names kept.
# ------------------------------------------------------------------------------------------------
class _LayerNorm2d(nn.Module):
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = eps

    def forward(self, x):     # normalise over the channel axis of (B,C,H,W)
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class _SamMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.lin1 = nn.Linear(dim, hidden)
        self.lin2 = nn.Linear(hidden, dim)

    def forward(self, x):
        y = self.lin1(x)
        if _tower_fused(y, y.shape[-1]) and y.is_contiguous():      # frozen tower on the device: in-place streaming GELU (fused.gelu_)
            from . import fused
            return self.lin2(fused.gelu_(y))
        return self.lin2(F.gelu(y))


class _SamAttention(nn.Module):
    def __init__(self, dim, heads, input_size):
        super().__init__()
        self.num_heads = heads
        hd = dim // heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, hd))

    def forward(self, x):      # (B, Hh, Ww, C)
        B, Hh, Ww, C = x.shape
        nh, hd = self.num_heads, C // self.num_heads
        qkv = self.qkv(x).view(B, Hh * Ww, 3, nh, hd)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                 # (B, L, nh, hd)
        # decomposed relative position bias (src/sam.py:392-426), from the *unscaled* q: inside the MFMA kernel on the GPU
        o = attention_relpos(q, k, v, self.rel_pos_h, self.rel_pos_w, (Hh, Ww))
        return self.proj(o.reshape(B, Hh, Ww, C))


class _SamBlock(nn.Module):
    def __init__(self, dim, heads, window, grid):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _SamAttention(dim, heads, (window, window) if window > 0 else (grid, grid))
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _SamMlp(dim, dim * 4)
        self.window_size = window

    def forward(self, x):      # (B, H, W, C)
        short = x
        x = self.attn_branch(self.norm1(x))
        x = short + x
        return x + self.mlp(self.norm2(x))

    def attn_branch(self, x):  # norm1 output (B, H, W, C) -> attention output, windows partitioned / merged
        ws = self.window_size
        if ws > 0:
            B, H, W, C = x.shape
            ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
            x = F.pad(x, (0, 0, 0, pw, 0, ph))
            Hp, Wp = H + ph, W + pw
            x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
            x = self.attn(x)
            x = x.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
            x = x[:, :H, :W].contiguous()
        else:
            x = self.attn(x)
        return x


class _SamPatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):      # (B,3,H,W) -> (B, H/p, W/p, C)
        B, C, H, W = x.shape
        p = self.patch
        gh, gw = H // p, W // p
        x = x.view(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh, gw, C * p * p)
        return F.linear(x, self.proj.weight.view(self.proj.weight.shape[0], -1), self.proj.bias)


class SamImageEncoder(nn.Module):
    """SAM ViT-B: 12 blocks, window 14 except global attention at [2,5,8,11], conv neck to 256 channels,
    then the lossless 2x2 space-to-depth of the reference (src/sam.py:47-74) -> (B, 1024, 32, 32)."""

    def __init__(self, img_size=1024, patch=16, dim=768, depth=12, heads=12, out_chans=256, window=14,
                 global_attn_indexes=(2, 5, 8, 11)):
        super().__init__()
        g = img_size // patch
        self.patch_embed = _SamPatchEmbed(patch, dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, g, g, dim))
        self.blocks = nn.ModuleList([_SamBlock(dim, heads, 0 if i in global_attn_indexes else window, g)
                                     for i in range(depth)])
        self.neck = nn.Sequential(nn.Conv2d(dim, out_chans, 1, bias=False), _LayerNorm2d(out_chans),
                                  nn.Conv2d(out_chans, out_chans, 3, padding=1, bias=False), _LayerNorm2d(out_chans))
        self.output_dim = out_chans * 4

    def forward(self, x):
        from . import fused
        pe = self.patch_embed
        if (not torch.is_grad_enabled() or not pe.proj.weight.requires_grad) and pe.proj.weight.dtype == torch.bfloat16 \
                and fused.patch_embed_supported(x, pe.patch, pe.proj.weight.shape[0]):
            g = x.shape[-1] // pe.patch
            x = fused.patch_embed(x, pe.proj.weight, pe.proj.bias, self.pos_embed[0].reshape(g * g, -1),
                                  padded_weight=fused.cached_padded_patch_weight(pe, pe.proj.weight)).view(x.shape[0], g, g, -1)
        else:
            x = pe(x) + self.pos_embed
        if _tower_fused(x, x.shape[-1]) and self.blocks[0].norm1.weight.dtype == torch.bfloat16:
            from . import fused                     # residual adds fused into the LayerNorms (src/sam.py:231-247 chained)
            delta = None
            for blk in self.blocks:
                x, h = fused.add_ln_infer(x, delta, None, blk.norm1)
                x, h = fused.add_ln_infer(x, blk.attn_branch(h), None, blk.norm2)
                delta = blk.mlp(h)
            x = x + delta
        else:
            for blk in self.blocks:
                x = blk(x)
        x = self.neck_nhwc(x)                       # (B, H, W, C)
        B, H, W, C = x.shape                        # space-to-depth: each 2x2 block -> 4C channels, channel index (c, dh, dw)
        x = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, C * 4)
        return x.permute(0, 3, 1, 2)

    def neck_nhwc(self, x):
        """The neck of src/sam.py:149-165 (conv 1x1, LayerNorm2d, conv 3x3 pad 1, LayerNorm2d) on the channels-last activations the
        blocks produce, as two GEMMs: a 1x1 convolution is a Linear over the channel axis, the 3x3 one a Linear over the nine
        shifted copies of the zero-padded map (weight re-ordered to (out, kh, kw, in)).  Handing MIOpen the (B, C, H, W) VIEW of
        this tensor made it run its find step on the first call with `naive_conv_ab_nonpacked_fwd_nhwc_*_double_*` as the reference
        (25.6 ms per call, 16 calls in the warm-up: profiles/r4b_cfg5_kernel_stats_head.csv); the timed steps are unchanged (CK's
        grouped-conv kernel took 0.68 ms per call).  Same modules and state-dict keys; `self.neck(x_nchw)` stays valid."""
        c1, n1, c3, n2 = self.neck
        B, H, W, C = x.shape
        y = F.linear(x, c1.weight.view(c1.weight.shape[0], C))
        y = F.layer_norm(y, (y.shape[-1],), n1.weight, n1.bias, n1.eps)
        Co = y.shape[-1]
        yp = F.pad(y, (0, 0, 1, 1, 1, 1))           # zero border on W and H
        cols = torch.cat([yp[:, kh:kh + H, kw:kw + W, :] for kh in range(3) for kw in range(3)], dim=-1)      # (B, H, W, 9 Co), (kh, kw, in) order
        w3 = c3.weight.permute(0, 2, 3, 1).reshape(c3.weight.shape[0], 9 * Co)
        z = F.linear(cols, w3)
        return F.layer_norm(z, (z.shape[-1],), n2.weight, n2.bias, n2.eps)


# ------------------------------------------------------------------------------------------------
# Pretrained tower files (vision.py:58-70: timm `pretrained_cfg_overlay=dict(file=...)`; sam.py:498-505)
# ------------------------------------------------------------------------------------------------
_TIMM_IGNORED = ("attn_pool.", "fc_norm.", "head.", "mask_token")     # dropped by timm for num_classes=0 / never used by
                                                                      # get_intermediate_layers (SigLIP's pooling head)


def _read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    if not isinstance(path, (str, bytes)) or not path or not __import__("os").path.exists(path):
        raise FileNotFoundError(f"vision tower checkpoint not found: {path!r}")
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model"):                      # common wrappers
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    return sd


def _resample_pos_embed(pe: torch.Tensor, grid: int, num_prefix: int) -> torch.Tensor:
    """timm `resample_abs_pos_embed`: the grid part of a learned position embedding to `grid` x `grid` (bicubic,
    antialias), prefix tokens kept.  pe: (1, P + g*g, D)."""
    if pe.shape[1] == num_prefix + grid * grid:
        return pe
    prefix, body = pe[:, :num_prefix], pe[:, num_prefix:]
    g0 = int(round(body.shape[1] ** 0.5))
    if g0 * g0 != body.shape[1]:
        raise ValueError(f"position embedding with {body.shape[1]} grid entries is not square")
    body = body.reshape(1, g0, g0, -1).permute(0, 3, 1, 2).float()
    body = F.interpolate(body, size=(grid, grid), mode="bicubic", antialias=True, align_corners=False)
    body = body.permute(0, 2, 3, 1).reshape(1, grid * grid, -1).to(pe.dtype)
    return torch.cat([prefix, body], dim=1) if num_prefix else body


def load_timm_vit(model: "TimmViT", path: str) -> None:
    """Load a timm VisionTransformer checkpoint (DINOv2-reg4 / SigLIP-so400m) into the restated tower: pooling-head keys
    are ignored, the position embedding is resampled to this tower's grid (timm does the same when `img_size` differs from
    the pretraining size: 37x37 or 27x27 -> 32x32 at 448), everything else must match exactly."""
    sd = {k: v for k, v in _read_checkpoint(path).items() if not k.startswith(_TIMM_IGNORED)}
    if "pos_embed" in sd:
        pe = sd["pos_embed"]
        # DINOv2-reg (no_embed_class) stores grid entries only; plain class-token ViTs store 1 + g*g
        n_grid = model.grid * model.grid
        extra = pe.shape[1] - int(round((pe.shape[1]) ** 0.5)) ** 2
        num_prefix = extra if extra in (0, 1) and pe.shape[1] != n_grid else 0
        pe = _resample_pos_embed(pe, model.grid, num_prefix)
        sd["pos_embed"] = pe[:, num_prefix:] if num_prefix else pe
    own = model.state_dict()
    missing, unexpected = sorted(set(own) - set(sd)), sorted(set(sd) - set(own))
    if missing or unexpected:
        raise RuntimeError(f"{path}: not a checkpoint of this tower (missing {missing[:5]}, unexpected {unexpected[:5]})")
    for k, v in sd.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise RuntimeError(f"{path}: {k} has shape {tuple(v.shape)}, the tower expects {tuple(own[k].shape)}")
    model.load_state_dict(sd, strict=True)


def load_sam_encoder(model: "SamImageEncoder", path: str) -> None:
    """SAM checkpoint -> image encoder (src/sam.py:498-505: keys under `image_encoder.`; prompt encoder / mask decoder
    ignored).  Unlike the reference (strict=False + a printed message) a key or shape mismatch is an error."""
    sd = _read_checkpoint(path)
    if any(k.startswith("image_encoder.") for k in sd):
        sd = {k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}
    own = model.state_dict()
    missing = sorted(set(own) - set(sd))
    if missing:
        raise RuntimeError(f"{path}: SAM image-encoder keys missing: {missing[:5]}")
    model.load_state_dict({k: sd[k] for k in own}, strict=True)


class SamDinoSigLIPViTBackbone(nn.Module):
    """vision.py:49-145 without the PIL/timm transform plumbing (pixel tensors come in pre-processed).
    `towers` selects which encoders exist; the concatenation order is dino, siglip, sam (vision.py:134)."""

    def __init__(self, vision_tower_path: Optional[dict] = None, default_image_size: int = 448,
                 towers: Sequence[str] = ("dino", "siglip", "sam"), tower_kwargs: Optional[Dict[str, dict]] = None):
        super().__init__()
        self._streams = None
        tk = tower_kwargs or {}
        self.towers = tuple(towers)
        if "dino" in towers:
            self.dino_featurizer = dinov2_large_reg4(default_image_size, **tk.get("dino", {}))
        if "siglip" in towers:
            self.siglip_featurizer = siglip_so400m(default_image_size, **tk.get("siglip", {}))
        if "sam" in towers:
            self.sam_featurizer = SamImageEncoder(**tk.get("sam", {}))
        if vision_tower_path:           # the reference's pretrained files: all requested towers must load, or it is an error
            for name, loader in (("dino", load_timm_vit), ("siglip", load_timm_vit), ("sam", load_sam_encoder)):
                if name in towers:
                    if name not in vision_tower_path:
                        raise KeyError(f"vision_tower_path has no entry for the '{name}' tower")
                    loader(getattr(self, f"{name}_featurizer"), vision_tower_path[name])
        self.eval()

    @property
    def embed_dim(self) -> int:
        d = 0
        if "dino" in self.towers:
            d += self.dino_featurizer.embed_dim
        if "siglip" in self.towers:
            d += self.siglip_featurizer.embed_dim
        if "sam" in self.towers:
            d += self.sam_featurizer.output_dim
        return d

    def _tower(self, name, pixel_values):
        if name == "sam":
            s = self.sam_featurizer(pixel_values["sam"])
            B, C, H, W = s.shape
            return s.view(B, C, H * W).permute(0, 2, 1)
        return getattr(self, f"{name}_featurizer")(pixel_values[name])

    def forward(self, pixel_values: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Concatenated features of the towers (src/vision.py:123-134).  On the device and without autograd (the towers are frozen) the towers run
        CONCURRENTLY, one HIP stream each: a tower's GEMMs have 16 384 rows x 1 024-1 152 columns = 320 tiles of 256 x 256 for 256 CUs, i.e. a full
        round and a quarter-filled one, and the other tower's kernels take the idle CUs (VRWKV_VIT_STREAMS=0: one after the other).  Library GEMMs of
        two streams at once are only safe with kernels that were checked side by side (gemm_tuning's docstring: two stream-K kernels deadlock), so
        this needs the tower configuration's key in the loaded tuning file's sidecar; otherwise the towers run one after the other."""
        names = [n for n in ("dino", "siglip", "sam") if n in self.towers]
        x0 = pixel_values[names[0]]
        if (VIT_STREAMS and len(names) > 1 and x0.is_cuda and not torch.is_grad_enabled()
                and gemm_tuning.concurrent_ok("vit " + " ".join(f"{n}:{'x'.join(str(d) for d in pixel_values[n].shape)}" for n in names))):
            cur = torch.cuda.current_stream(x0.device)
            if self._streams is None or self._streams[0].device != x0.device:
                self._streams = [torch.cuda.Stream(device=x0.device) for _ in names[1:]]
            outs = [None] * len(names)
            for i, name in enumerate(names[1:], start=1):
                st = self._streams[i - 1]
                st.wait_stream(cur)                          # the pixels (and whatever wrote them) are ready
                with torch.cuda.stream(st):
                    outs[i] = self._tower(name, pixel_values)
            outs[0] = self._tower(names[0], pixel_values)    # the first tower on the caller's stream
            for i, st in enumerate(self._streams, start=1):
                cur.wait_stream(st)
                outs[i].record_stream(cur)                   # allocated on the side stream, consumed (and freed) on the caller's
            return torch.cat(outs, dim=2)
        return torch.cat([self._tower(n, pixel_values) for n in names], dim=2)
