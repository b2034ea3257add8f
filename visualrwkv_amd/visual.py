"""VisualRWKV-7: ViT features -> pool -> projector -> scatter into the token embeddings -> RWKV-7 LM.

Mirror of VisualRWKV-v7/v7.00/src/model.py:328-530 (`MLPWithContextGating`, `VisualRWKV`) without the
Lightning/DeepSpeed base classes: same constructor argument object, same sub-module names
(`rwkv`, `vit`, `proj`, `pool` => same state-dict keys), same batch-dict schema
(`input_ids`, `labels`, `images{dino,siglip,sam,...}`, `sample_id`; src/dataset.py:24-36), same loss.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn
from torch.nn import functional as F

from .rwkv7 import IGNORE_INDEX, IMAGE_TOKEN_INDEX, L2Wrap, RWKV
from .vit import SamDinoSigLIPViTBackbone


class MLPWithContextGating(nn.Module):
    """LayerNorm(o_proj(x * sigmoid(gate(x))))   (src/model.py:328-338)"""

    def __init__(self, in_dim, n_embd):
        super().__init__()
        self.gate = nn.Linear(in_dim, in_dim, bias=False)
        self.o_proj = nn.Linear(in_dim, n_embd, bias=False)
        self.ln_v = nn.LayerNorm(n_embd)

    def forward(self, x):
        return self.ln_v(self.pre_norm(x))

    def pre_norm(self, x):
        """o_proj(x * sigmoid(gate(x))): everything before ln_v.  On the GPU the gate is one streaming HIP kernel and
        ln_v is fused with the scatter into the token embeddings by the caller (VisualRWKV.preparing_embedding)."""
        from . import fused
        if fused.visual_supported(x) and self.gate.weight.dtype == torch.bfloat16:
            return self.o_proj(fused.gate(x, self.gate(x)))
        return self.o_proj(x * torch.sigmoid(self.gate(x)))


class VisualRWKV(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.rwkv = RWKV(args)
        if len(getattr(args, "load_model", "")) > 0:
            self.rwkv.load_state_dict(torch.load(args.load_model, map_location="cpu", weights_only=True))
        self.vit = SamDinoSigLIPViTBackbone(getattr(args, "vision_tower_path", None),
                                            towers=getattr(args, "vision_towers", ("dino", "siglip", "sam")),
                                            default_image_size=getattr(args, "vision_image_size", 448),
                                            tower_kwargs=getattr(args, "vision_tower_kwargs", None))
        self.freeze_vit()
        if getattr(args, "proj_type", "mlp") == "linear":
            self.proj = nn.Linear(self.vit.embed_dim, args.n_embd, bias=False)
        else:
            self.proj = MLPWithContextGating(self.vit.embed_dim, args.n_embd)
        self.pool = nn.AdaptiveAvgPool2d(int(args.num_token_per_image ** 0.5))

    # ---- freezing helpers (src/model.py:368-388)
    def freeze_vit(self):
        self.vit.requires_grad_(False)

    def freeze_rwkv(self, num_layers_to_freeze):
        if num_layers_to_freeze == self.args.n_layer:
            self.rwkv.requires_grad_(False)
        for i, block in enumerate(self.rwkv.blocks):
            block.requires_grad_(i >= num_layers_to_freeze)

    def freeze_emb(self):
        self.rwkv.emb.requires_grad_(False)

    def freeze_proj(self):
        self.proj.requires_grad_(False)

    def optimizer_groups(self):
        """Parameter groups of configure_optimizers (src/model.py:390-410): tensors that are at least 2-D
        after squeeze() get weight decay, everything else none."""
        no_wd = [p for p in self.parameters() if p.requires_grad and len(p.squeeze().shape) < 2]
        wd = [p for p in self.parameters() if p.requires_grad and len(p.squeeze().shape) >= 2]
        groups = []
        if no_wd:
            groups.append({"params": no_wd, "weight_decay": 0.0})
        if wd:
            groups.append({"params": wd, "weight_decay": float(getattr(self.args, "weight_decay", 0.0))})
        return groups

    # ---- forward path
    def adaptive_pooling(self, image_features):
        from . import fused
        if getattr(getattr(self, "args", None), "fused", False) and fused.visual_supported(image_features):
            osz = self.pool.output_size
            return fused.adaptive_pool(image_features, osz if isinstance(osz, int) else osz[0])
        B, Ln, D = image_features.shape
        side = int(Ln ** 0.5)
        x = image_features.view(B, side, side, D).permute(0, 3, 1, 2)
        return self.pool(x).view(B, D, -1).permute(0, 2, 1)

    def encode_images(self, images: dict, minibatch_size: int = None, normed: bool = True) -> torch.Tensor:
        """ViTs (frozen, no grad) in mini-batches of `minibatch_size` images -> pool -> projector
        (src/model.py:449-471; the reference's per-mini-batch torch.cuda.empty_cache() is a device sync
        plus an allocator flush and is deliberately not reproduced).  The reference encodes 4 images at a time to save
        memory; `args.vit_minibatch` raises that where the HBM allows (no activations are kept: the towers are frozen) --
        the tower GEMMs run at 0.6 PFLOP/s with 4 images (4 096 rows) and at 0.9 with 16."""
        if minibatch_size is None:
            minibatch_size = int(getattr(getattr(self, "args", None), "vit_minibatch", 4) or 4)
        keys = [k for k in ("dino", "siglip", "sam") if k in images]
        n = len(images[keys[0]])
        feats = []
        with torch.no_grad():
            for i in range(0, n, minibatch_size):
                feats.append(self.vit({k: images[k][i:i + minibatch_size] for k in keys}))
        image_features = feats[0] if len(feats) == 1 else torch.cat(feats, dim=0)
        pooled = self.adaptive_pooling(image_features.detach())
        if normed:
            return self.proj(pooled)
        return self.proj.pre_norm(pooled)               # ln_v is applied by the fused scatter

    def preparing_embedding(self, samples):
        if "images" not in samples:
            return self.rwkv.emb(samples["input_ids"]), samples["labels"]
        from . import fused
        ids = samples["input_ids"]
        B, Ln = ids.shape
        D = self.rwkv.emb.weight.shape[1]
        selected = ids.reshape(B * Ln) == IMAGE_TOKEN_INDEX
        if (getattr(self.args, "fused", False) and isinstance(getattr(self, "proj", None), MLPWithContextGating) and ids.is_cuda
                and self.rwkv.emb.weight.dtype == torch.bfloat16 and not getattr(self.args, "check_image_tokens", True)
                and self.proj.ln_v.weight.dtype == torch.bfloat16 and D % 64 == 0):
            # GPU path: ln_v of the projector writes straight into the placeholder rows (no masked_scatter, no host sync: the
            # row list comes from a stable sort of the mask).  The frozen towers run BEFORE the first trainable module is
            # touched, so the ZeRO-1 parameter all-gather of the previous step (dp.py) overlaps the ViT encode.
            feats = self.encode_images(samples["images"], normed=False)
            feats = feats.reshape(-1, feats.shape[-1])
            input_embeds = self.rwkv.emb(ids).view(B * Ln, D)
            n_feat = feats.shape[0]
            if n_feat > B * Ln:                              # more features than tokens at all: truncate like the reference
                feats, n_feat = feats[:B * Ln], B * Ln
            rows = torch.argsort(~selected, stable=True)[:n_feat]
            # fewer placeholders than features (a multi-image sample truncated at ctx_len): the reference keeps the first
            # n_sel features and warns (model.py:487-491).  Same result without a host synchronisation: the surplus
            # features get row -1, which the kernels drop in both directions -- they never overwrite text embeddings.
            rows = torch.where(selected[rows], rows, torch.full_like(rows, -1))
            input_embeds = fused.ln_scatter(input_embeds, feats, self.proj.ln_v, rows)
            return input_embeds.view(B, Ln, D), samples["labels"]
        input_embeds = self.rwkv.emb(ids)
        input_embeds = input_embeds.view(B * Ln, D)
        image_features = self.encode_images(samples["images"])
        image_features = image_features.view(-1, image_features.shape[-1])
        n_sel = int(selected.sum()) if getattr(self.args, "check_image_tokens", True) else image_features.shape[0]
        if n_sel != image_features.shape[0]:
            n_feat = image_features.shape[0]
            image_features = image_features[:n_sel]      # the reference truncates and warns (model.py:487-491)
            warnings.warn(f"image tokens: {n_sel}, but image features: {n_feat}")
        input_embeds = input_embeds.masked_scatter(selected[:, None], image_features.to(input_embeds.dtype))
        return input_embeds.view(B, Ln, D), samples["labels"]

    def forward(self, samples):
        x, targets = self.preparing_embedding(samples)
        return self.rwkv(x), targets

    @staticmethod
    def loss_from_logits(logits, targets):
        """Shifted CE, summed per sample over valid labels / max(valid,1), batch mean, wrapped in L2Wrap
        (src/model.py:418-434)."""
        shift_logits = logits[..., :-1, :].contiguous()
        shift_labels = targets[..., 1:].contiguous()
        valid = (shift_labels != IGNORE_INDEX).sum(1)
        valid = torch.max(valid, torch.ones_like(valid))
        loss = F.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1),
                               ignore_index=IGNORE_INDEX, reduction="none")
        loss = (loss.view(shift_labels.size()).sum(1) / valid).mean()
        return L2Wrap.apply(loss, logits)

    def training_step(self, batch, batch_idx=0):
        logits, targets = self(batch)
        if getattr(self.args, "fused", False):
            from . import fused
            if fused.ce_supported(logits):
                return fused.loss_from_logits(logits, targets, IGNORE_INDEX)
        return self.loss_from_logits(logits, targets)

    @torch.no_grad()
    def generate(self, input_ids, images, do_sample, temperature, top_p, max_new_tokens, stop_token_idx):
        """Greedy decoding with the reference's semantics (src/model.py:496-530): the full sequence is
        re-run for every new token (stateful decoding is SURVEY.md 8f rank 1)."""
        if do_sample:
            raise NotImplementedError
        samples = {"input_ids": input_ids, "images": images, "labels": torch.full_like(input_ids, IGNORE_INDEX)}
        x, _ = self.preparing_embedding(samples)
        toks, lgs, prs = [], [], []
        for _ in range(max_new_tokens):
            logits = self.rwkv(x)[:, -1, :]
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            toks.append(nxt.item())
            lgs.append(logits.gather(-1, nxt).item())
            prs.append(torch.softmax(logits, dim=-1).gather(-1, nxt).item())
            if toks[-1] == stop_token_idx:
                break
            x = torch.cat((x, self.rwkv.emb(nxt)), dim=-2)[:, -self.args.ctx_len:, :]
        return toks, lgs, prs

    @torch.no_grad()
    def generate_stateful(self, input_ids, images, do_sample, temperature, top_p, max_new_tokens, stop_token_idx,
                          use_graph=None):
        """`generate` with the recurrent state carried between tokens: one prefill over the prompt, then one
        single-token step per new token (O(1) per token instead of re-running the whole sequence).
        The prompt is left-padded once, like `RWKV.forward` pads it (src/model.py:301-307), so the first token
        is the one `generate` returns; later tokens are conditioned on that same fixed prefix, whereas the
        reference's per-step re-padding changes the number of pad tokens as the sequence grows."""
        if do_sample:
            raise NotImplementedError
        from .rwkv7 import CHUNK_LEN
        samples = {"input_ids": input_ids, "images": images, "labels": torch.full_like(input_ids, IGNORE_INDEX)}
        x, _ = self.preparing_embedding(samples)
        x = x[:, -self.args.ctx_len:, :]
        rem = x.size(1) % CHUNK_LEN
        x = self.rwkv.pad_left(x, CHUNK_LEN - rem if rem else 0)
        logits, state = self.rwkv.forward_stateful(x, None, last_only=True)
        if use_graph is None:                            # captured step where the batched-GEMV decode path applies
            use_graph = (x.is_cuda and bool(getattr(self.args, "fused", False)) and x.dtype == torch.bfloat16 and x.size(0) <= 4
                         and max_new_tokens >= 8)
        decoder = self.rwkv.decoder_for(state) if use_graph and x.is_cuda else None     # one capture per batch size, re-used
        toks, lgs, prs = [], [], []
        for _ in range(max_new_tokens):
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            toks.append(nxt.item())
            lgs.append(logits.gather(-1, nxt).item())
            prs.append(torch.softmax(logits, dim=-1).gather(-1, nxt).item())
            if toks[-1] == stop_token_idx or len(toks) == max_new_tokens:
                break
            if decoder is not None:
                logits = decoder(self.rwkv.emb(nxt))
            else:
                logits, state = self.rwkv.forward_stateful(self.rwkv.emb(nxt), state, last_only=True)
        return toks, lgs, prs
