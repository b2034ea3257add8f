"""Host-side mirror of the reference's RWKV-6 modules (VisualRWKV-v6/v6.0/src/model.py:92-226): same class names,
constructor arguments, parameter names and initialisers (=> the reference's state-dict keys), forward through
RUN_CUDA_RWKV6.  BASELINE config 4 (VisualRWKV-6 7B) runs the same Block/RWKV/VisualRWKV scaffolding around these."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import wkv6 as _wkv6


def _lora_mm(x, w):
    """x @ w for a LoRA factor; on the GPU under autograd the weight gradient runs in csrc/lora_wgrad.h (fused.lora_mm)."""
    if x.is_cuda and torch.is_grad_enabled():
        from .fused import lora_mm
        return lora_mm(x, w)
    return x @ w


def time_shift(x):
    """nn.ZeroPad2d((0, 0, 1, -1)): x[t-1], zero at t = 0."""
    return F.pad(x, (0, 0, 1, -1))


class RWKV_Tmix_x060(nn.Module):
    """RWKV-6 time-mix with the 5-way data-dependent token-shift LoRA (model.py:92-194)."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        self.head_size = args.head_size_a
        self.n_head = args.dim_att // self.head_size
        assert args.dim_att % self.n_head == 0
        C, A = args.n_embd, args.dim_att
        with torch.no_grad():
            r01 = layer_id / (args.n_layer - 1)
            r10 = 1.0 - (layer_id / args.n_layer)
            ddd = (torch.arange(C, dtype=torch.float32) / C).view(1, 1, C)
            self.time_maa_x = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_w = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_k = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_v = nn.Parameter(1.0 - (torch.pow(ddd, r10) + 0.3 * r01))
            self.time_maa_r = nn.Parameter(1.0 - torch.pow(ddd, 0.5 * r10))
            self.time_maa_g = nn.Parameter(1.0 - torch.pow(ddd, 0.5 * r10))
            d_mix = 64 if C >= 4096 else 32
            self.time_maa_w1 = nn.Parameter(torch.zeros(C, d_mix * 5))
            self.time_maa_w2 = nn.Parameter(torch.zeros(5, d_mix, C).uniform_(-0.01, 0.01))
            n = torch.arange(A, dtype=torch.float32)
            self.time_decay = nn.Parameter((-6 + 5 * (n / (A - 1)) ** (0.7 + 1.3 * r01)).reshape(1, 1, A))
            d_decay = 128 if C >= 4096 else 64
            self.time_decay_w1 = nn.Parameter(torch.zeros(C, d_decay))
            self.time_decay_w2 = nn.Parameter(torch.zeros(d_decay, A).uniform_(-0.01, 0.01))
            zigzag = ((torch.arange(A) + 1) % 3 - 1).float() * 0.1
            self.time_faaaa = nn.Parameter((r01 * (1 - n / (A - 1)) + zigzag).reshape(self.n_head, self.head_size))
        self.receptance = nn.Linear(C, A, bias=False)
        self.key = nn.Linear(C, A, bias=False)
        self.value = nn.Linear(C, A, bias=False)
        self.output = nn.Linear(A, C, bias=False)
        self.gate = nn.Linear(C, A, bias=False)
        self.ln_x = nn.GroupNorm(self.n_head, A, eps=(1e-5) * (args.head_size_divisor ** 2))

    def mix(self, x):
        B, T, C = x.size()
        xx = time_shift(x) - x
        xxx = x + xx * self.time_maa_x
        xxx = torch.tanh(_lora_mm(xxx, self.time_maa_w1)).view(B * T, 5, -1).transpose(0, 1)
        xxx = torch.bmm(xxx, self.time_maa_w2).view(5, B, T, -1)
        mw, mk, mv, mr, mg = xxx.unbind(dim=0)
        xw = x + xx * (self.time_maa_w + mw)
        xk = x + xx * (self.time_maa_k + mk)
        xv = x + xx * (self.time_maa_v + mv)
        xr = x + xx * (self.time_maa_r + mr)
        xg = x + xx * (self.time_maa_g + mg)
        r = self.receptance(xr)
        k = self.key(xk)
        v = self.value(xv)
        g = F.silu(self.gate(xg))
        w = self.time_decay + _lora_mm(torch.tanh(_lora_mm(xw, self.time_decay_w1)), self.time_decay_w2)
        return r, k, v, g, w

    def forward(self, x, wkv=None):
        B, T, C = x.size()
        if getattr(self.args, "fused", False):
            from . import fused
            if fused.supported6(x, self) and self.receptance.weight.dtype == torch.bfloat16:
                return fused.tmix6_forward(self, x, wkv)
        r, k, v, g, w = self.mix(x)
        run = wkv if wkv is not None else _wkv6.RUN_CUDA_RWKV6
        y = run(B, T, C, self.n_head, r, k, v, w, self.time_faaaa)
        y = self.ln_x(y.view(B * T, C)).view(B, T, C)
        return self.output(y * g)


class RWKV_CMix_x060(nn.Module):
    """RWKV-6 channel-mix with the receptance gate (model.py:198-226)."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        C = args.n_embd
        with torch.no_grad():
            r10 = 1.0 - (layer_id / args.n_layer)
            ddd = (torch.arange(C, dtype=torch.float32) / C).view(1, 1, C)
            self.time_maa_k = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_r = nn.Parameter(1.0 - torch.pow(ddd, r10))
        self.key = nn.Linear(C, args.dim_ffn, bias=False)
        self.receptance = nn.Linear(C, C, bias=False)
        self.value = nn.Linear(args.dim_ffn, C, bias=False)

    def forward(self, x):
        if getattr(self.args, "fused", False):
            from . import fused
            if fused.supported6(x) and self.key.weight.dtype == torch.bfloat16:
                return fused.cmix6_forward(self, x)
        xx = time_shift(x) - x
        xk = x + xx * self.time_maa_k
        xr = x + xx * self.time_maa_r
        k = torch.relu(self.key(xk)) ** 2
        return torch.sigmoid(self.receptance(xr)) * self.value(k)


class Block(nn.Module):
    """Pre-LN residual block of RWKV-6 (model.py:233-258); block 0 also owns ln0."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        self.ln1 = nn.LayerNorm(args.n_embd)
        self.ln2 = nn.LayerNorm(args.n_embd)
        if layer_id == 0:
            self.ln0 = nn.LayerNorm(args.n_embd)
        self.att = RWKV_Tmix_x060(args, layer_id)
        self.ffn = RWKV_CMix_x060(args, layer_id)
        if args.dropout > 0:
            self.drop0 = nn.Dropout(p=args.dropout)
            self.drop1 = nn.Dropout(p=args.dropout)

    def forward(self, x, wkv=None):
        if self.layer_id == 0:
            x = self.ln0(x)
        if getattr(self.args, "fused", False) and self.args.dropout == 0:
            from . import fused
            if fused.supported6(x) and fused.add_ln_supported(x) and self.ln1.weight.dtype == torch.bfloat16:
                _, h = fused.add_ln(x, None, self.ln1)                         # LayerNorm kernel; the attention's residual add is
                x, h = fused.add_ln(x, self.att(h, wkv), self.ln2)            # fused with ln2 (csrc/ln_fused.hip)
                return x + self.ffn(h)
        x = x + self.att(self.ln1(x), wkv)
        x = x + self.ffn(self.ln2(x))
        return x


class RWKV(nn.Module):
    """Embedding -> n_layer Blocks -> ln_out -> head (model.py:277-325); same state-dict keys as the reference."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.emb = nn.Embedding(args.vocab_size, args.n_embd)
        self.blocks = nn.ModuleList([Block(args, i) for i in range(args.n_layer)])
        self.ln_out = nn.LayerNorm(args.n_embd)
        self.head = nn.Linear(args.n_embd, args.vocab_size, bias=False)
        if args.dropout > 0:
            self.drop0 = nn.Dropout(p=args.dropout)

    def forward(self, x, wkv=None):
        args = self.args
        if args.dropout > 0:
            x = self.drop0(x)
        if getattr(args, "fused", False) and args.dropout == 0:
            from . import fused
            if fused.supported6(x) and fused.add_ln_supported(x) and self.head.weight.dtype == torch.bfloat16:
                h = fused.blocks6_forward(self, x, wkv, grad_cp=args.grad_cp >= 1 and torch.is_grad_enabled())
                return fused.linear(self.head, h)
        for block in self.blocks:
            if args.grad_cp >= 1 and torch.is_grad_enabled():      # 2 (the v7 fused path's selective mode) means 1 here
                from torch.utils.checkpoint import checkpoint
                x = checkpoint(block, x, wkv, use_reentrant=False)
            else:
                x = block(x, wkv)
        return self.head(self.ln_out(x))
