"""VisualRWKV-6 (BASELINE config 4) around the RWKV-6 stack: mirror of VisualRWKV-v6/v6.0/src/model.py:339-560.

CLIP ViT-L/14 features (any module returning `.last_hidden_state` with a leading CLS token), `grid_pooling`, a linear
projector, the image span inserted after the left-padded first text part (`preparing_embedding`), and the
"bidirectional" pass that flips the image span on odd layers (`bidirectional_forward`).  Loss = the v7 one
(`training_step`, model.py:441-458 is the same code)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .rwkv6 import RWKV
from .rwkv7 import IGNORE_INDEX, L2Wrap

IMAGE_TOKEN_INDEX = -200       # VisualRWKV-v6/v6.0/src/dataset.py (v7 moved it to 65535)


class VisualRWKV6(nn.Module):
    def __init__(self, args, vit: nn.Module, vit_hidden_size: int):
        """`vit`: the frozen vision tower (the reference builds `CLIPVisionModel.from_pretrained(args.vision_tower_name)`,
        model.py:346; pass that model, or any module with the same output attribute)."""
        super().__init__()
        self.args = args
        self.rwkv = RWKV(args)
        self.vit = vit
        self.vit.requires_grad_(False)
        self.proj = nn.Linear(vit_hidden_size, args.n_embd, bias=False)
        self.img_start = self.img_end = 0

    def freeze_rwkv(self, num_layers_to_freeze):
        if num_layers_to_freeze == self.args.n_layer:
            self.rwkv.requires_grad_(False)
        for i, block in enumerate(self.rwkv.blocks):
            block.requires_grad_(i >= num_layers_to_freeze)

    def freeze_emb(self):
        self.rwkv.emb.requires_grad_(False)

    def freeze_proj(self):
        self.proj.requires_grad_(False)

    # ---- model.py:401-427
    def forward(self, samples, wkv=None):
        x, targets, _ = self.preparing_embedding(samples)
        return self.bidirectional_forward(x, wkv), targets

    def bidirectional_forward(self, x, wkv=None):
        args = self.args
        if args.dropout > 0:
            x = self.rwkv.drop0(x)
        s, e = self.img_start, self.img_end
        for i, block in enumerate(self.rwkv.blocks):
            rev = i % 2 == 1
            if rev:                                                    # odd layers read the image span right to left
                x = torch.cat((x[:, :s], x[:, s:e].flip(1), x[:, e:]), dim=1)
            if args.grad_cp >= 1 and torch.is_grad_enabled():      # 2 (the v7 fused path's selective mode) means 1 here
                from torch.utils.checkpoint import checkpoint
                x = checkpoint(block, x, wkv, use_reentrant=False)
            else:
                x = block(x, wkv)
            if rev:
                x = torch.cat((x[:, :s], x[:, s:e].flip(1), x[:, e:]), dim=1)
        if getattr(args, "fused", False) and x.is_cuda and self.rwkv.head.weight.dtype == torch.bfloat16:
            from . import fused
            return fused.linear(self.rwkv.head, self.rwkv.ln_out(x))           # input gradient in the forward GEMMs' layout
        return self.rwkv.head(self.rwkv.ln_out(x))

    def training_step(self, batch, batch_idx=0, wkv=None):
        logits, targets = self(batch, wkv)
        if getattr(self.args, "fused", False):
            from . import fused
            if fused.ce_supported(logits):                                      # one-pass shifted CE + L2Wrap (csrc/loss_fused.hip)
                return fused.loss_from_logits(logits, targets, IGNORE_INDEX)
        shift_logits = logits[..., :-1, :].contiguous()
        shift_labels = targets[..., 1:].contiguous()
        valid = torch.max((shift_labels != IGNORE_INDEX).sum(1), torch.ones_like(shift_labels[:, 0]))
        loss = F.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1),
                               ignore_index=IGNORE_INDEX, reduction="none")
        loss = (loss.view(shift_labels.size()).sum(1) / valid).mean()
        return L2Wrap.apply(loss, logits)

    # ---- model.py:466-485
    def encode_images(self, images):
        B, n_img, C, H, W = images.shape
        feats = self.vit(images.view(B * n_img, C, H, W)).last_hidden_state
        feats = feats.view(B, n_img, feats.shape[1], feats.shape[2])[:, 0]
        return self.proj(self.grid_pooling(feats))

    def grid_pooling(self, image_features):
        cls_features = image_features[:, 0:1, :]
        image_features = image_features[:, 1:, :]
        gs = self.args.grid_size
        if gs == -1:
            return torch.cat((image_features, cls_features), dim=1)
        if gs == 0:
            return cls_features
        if gs == 1:
            return torch.cat((image_features.mean(dim=1, keepdim=True), cls_features), dim=1)
        B, L, D = image_features.shape
        side = int(L ** 0.5)
        stride = side // gs
        pooled = F.avg_pool2d(image_features.view(B, side, side, D).permute(0, 3, 1, 2), kernel_size=stride, stride=stride)
        return torch.cat((pooled.permute(0, 2, 3, 1).reshape(B, -1, D), cls_features), dim=1)

    # ---- model.py:487-560: [left-padded first text part | image features (+CLS) | rest of the text]
    def preparing_embedding(self, samples, truncate=True):
        ids, labels = samples["input_ids"], samples["labels"]
        device = labels.device
        image_features = self.encode_images(samples["images"])
        is_img = ids == IMAGE_TOKEN_INDEX
        n_img = is_img.sum(1)
        if int(n_img.max()) > 1:
            raise ValueError("Too many images in one sample, should be 0 or 1.")
        pos = torch.where(n_img == 1, is_img.int().argmax(1), torch.zeros_like(n_img))
        max_pos = int(pos.max())
        self.img_start = max_pos
        self.img_end = max_pos + (image_features.shape[1] - 1)                 # CLS excluded
        embeds, new_labels = [], []
        for b in range(ids.shape[0]):
            p = int(pos[b])
            head_ids = torch.zeros(max_pos, dtype=ids.dtype, device=device)
            head_lab = torch.full((max_pos,), IGNORE_INDEX, dtype=labels.dtype, device=device)
            feats = image_features[b]
            if int(n_img[b]) == 1:
                if p > 0:
                    head_ids[-p:] = ids[b, :p]
                    head_lab[-p:] = labels[b, :p]
                tail_ids, tail_lab = ids[b, p + 1:], labels[b, p + 1:]
            else:
                feats = torch.zeros_like(feats)
                tail_ids, tail_lab = ids[b], labels[b]
            embeds.append(torch.cat((self.rwkv.emb(head_ids), feats, self.rwkv.emb(tail_ids))))
            new_labels.append(torch.cat((head_lab, torch.full((feats.shape[0],), IGNORE_INDEX, dtype=labels.dtype, device=device),
                                         tail_lab)))
        if truncate:                                   # keep the beginning unless it carries no label (model.py:496-509)
            L = self.args.ctx_len
            for i, (x, y) in enumerate(zip(embeds, new_labels)):
                if bool((y[:L] != IGNORE_INDEX).any()):
                    embeds[i], new_labels[i] = x[:L], y[:L]
                else:
                    embeds[i], new_labels[i] = x[-L:], y[-L:]
        max_len = max(x.shape[0] for x in embeds)
        out = torch.zeros(len(embeds), max_len, self.args.n_embd, dtype=samples["images"].dtype, device=device)
        lab = torch.full((len(embeds), max_len), IGNORE_INDEX, dtype=labels.dtype, device=device)
        for i, (x, y) in enumerate(zip(embeds, new_labels)):
            out[i, :x.shape[0]] = x
            lab[i, :y.shape[0]] = y
        return out, lab, image_features
