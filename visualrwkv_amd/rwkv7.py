"""RWKV-7 ("x070") language-model blocks on top of the gfx950 WKV7 operator.

Host-side mirror of the reference's operator/module surface for the hot path
(VisualRWKV-v7/v7.00/src/model.py:76-325): same class names, constructor arguments, forward
signatures, parameter names (=> identical state-dict keys) and the same arithmetic, so that a
checkpoint of the reference loads unchanged and the v7.00 trainer/evaluator can call these classes
in place of its own.  What differs is underneath: the WKV7 recurrence is the chunked MFMA kernel of
libvisualrwkv_hip.so, and the element-wise glue around it can run as fused HIP kernels
(`visualrwkv_amd.fused`, opt-in per module through `fused=True`).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.nn import functional as F

from .wkv7 import CHUNK_LEN, RUN_CUDA_RWKV7g

STOP_TOKEN_INDEX = 261        # src/dataset.py:20  ("\n\n"), used to left-pad to a multiple of CHUNK_LEN
IGNORE_INDEX = -100           # src/dataset.py:17
IMAGE_TOKEN_INDEX = 65535     # src/dataset.py:18


def _lora_rank(C: int, factor: float, power: float = 0.5) -> int:
    """max(32, round(factor * C**power / 32) * 32)   (src/model.py:118,127,133,140)"""
    return max(32, int(round((factor * (C ** power)) / 32) * 32))


def _ortho(rows: int, cols: int, scale: float) -> torch.Tensor:
    w = torch.zeros(rows, cols)
    gain = math.sqrt(rows / cols) if rows > cols else 1.0
    nn.init.orthogonal_(w, gain=gain * scale)
    return w


def time_shift(x: torch.Tensor) -> torch.Tensor:
    """x_{t-1} with zeros at t = 0 of every sample (nn.ZeroPad2d((0,0,1,-1)), src/model.py:149)."""
    return F.pad(x, (0, 0, 1, -1))


class RWKV_Tmix_x070(nn.Module):
    """Time-mix of RWKV-7 (src/model.py:76-195)."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        self.head_size = args.head_size_a
        self.n_head = args.dim_att // self.head_size
        assert args.dim_att % self.n_head == 0
        H, N, C = self.n_head, self.head_size, args.n_embd

        with torch.no_grad():
            r01 = layer_id / (args.n_layer - 1) if args.n_layer > 1 else 0.0   # 0 -> 1 over depth
            r10 = 1.0 - (layer_id / args.n_layer)                                # 1 -> ~0
            ddd = (torch.arange(C, dtype=torch.float32) / C).view(1, 1, C)
            self.x_r = nn.Parameter(1.0 - torch.pow(ddd, 0.2 * r10))
            self.x_w = nn.Parameter(1.0 - torch.pow(ddd, 0.9 * r10))
            self.x_k = nn.Parameter(1.0 - (torch.pow(ddd, 0.9 * r10) + 0.4 * r01))
            self.x_v = nn.Parameter(1.0 - (torch.pow(ddd, 0.4 * r10) + 0.6 * r01))
            self.x_a = nn.Parameter(1.0 - torch.pow(ddd, 0.9 * r10))
            self.x_g = nn.Parameter(1.0 - torch.pow(ddd, 0.2 * r10))

            d_decay = _lora_rank(C, 1.8)
            self.w1 = nn.Parameter(torch.zeros(C, d_decay))
            self.w2 = nn.Parameter(_ortho(d_decay, C, 0.1))
            n = torch.arange(C, dtype=torch.float32)
            decay_speed = -7 + 5 * (n / (C - 1)) ** (0.85 + 1.0 * r01 ** 0.5)
            self.w0 = nn.Parameter(decay_speed.reshape(1, 1, C) + 0.5)          # +0.5: soft-clamp offset

            d_aaa = _lora_rank(C, 1.8)
            self.a1 = nn.Parameter(torch.zeros(C, d_aaa))
            self.a2 = nn.Parameter(_ortho(d_aaa, C, 0.1))
            self.a0 = nn.Parameter(torch.zeros(1, 1, C))

            d_mv = _lora_rank(C, 1.3)
            if layer_id != 0:                                                    # layer 0 defines v_first
                self.v1 = nn.Parameter(torch.zeros(C, d_mv))
                self.v2 = nn.Parameter(_ortho(d_mv, C, 0.1))
                self.v0 = nn.Parameter(torch.zeros(1, 1, C) + 1.0)

            d_gate = _lora_rank(C, 0.6, 0.8)
            self.g1 = nn.Parameter(torch.zeros(C, d_gate))
            self.g2 = nn.Parameter(_ortho(d_gate, C, 0.1))

            self.k_k = nn.Parameter(torch.ones(1, 1, C) * 0.85)
            self.k_a = nn.Parameter(torch.ones(1, 1, C))
            self.r_k = nn.Parameter(torch.zeros(H, N))

            self.receptance = nn.Linear(C, C, bias=False)
            self.key = nn.Linear(C, C, bias=False)
            self.value = nn.Linear(C, C, bias=False)
            self.output = nn.Linear(C, C, bias=False)
            self.ln_x = nn.GroupNorm(H, C, eps=(1e-5) * (args.head_size_divisor ** 2))   # eps = 64e-5

            self.receptance.weight.data.uniform_(-0.5 / (C ** 0.5), 0.5 / (C ** 0.5))
            self.key.weight.data.uniform_(-0.05 / (C ** 0.5), 0.05 / (C ** 0.5))
            self.value.weight.data.uniform_(-0.5 / (C ** 0.5), 0.5 / (C ** 0.5))
            self.output.weight.data.zero_()

    def forward(self, x, v_first, state=None):
        """`state` (an RWKV7State, inference only) carries the previous token and the WKV state across calls."""
        B, T, C = x.size()
        H = self.n_head
        if getattr(self.args, "fused", False) and x.is_cuda:
            from . import fused
            if state is None:
                return fused.tmix_forward(self, x, v_first)
            if not torch.is_grad_enabled() and x.dtype == torch.bfloat16:
                return fused.tmix_forward_stateful(self, x, v_first, state)
        if state is None:
            xx = time_shift(x) - x
        else:
            xx = torch.cat((state.att_x[self.layer_id].unsqueeze(1), x[:, :-1]), dim=1) - x
            state.att_x[self.layer_id].copy_(x[:, -1])          # in place: the state tensors are stable addresses (HIP graphs)
        xr = x + xx * self.x_r
        xw = x + xx * self.x_w
        xk = x + xx * self.x_k
        xv = x + xx * self.x_v
        xa = x + xx * self.x_a
        xg = x + xx * self.x_g

        r = self.receptance(xr)
        w = -F.softplus(-(self.w0 + torch.tanh(xw @ self.w1) @ self.w2)) - 0.5   # w_raw <= -0.5
        k = self.key(xk)
        v = self.value(xv)
        if self.layer_id == 0:
            v_first = v
        else:
            v = v + (v_first - v) * torch.sigmoid(self.v0 + (xv @ self.v1) @ self.v2)
        a = torch.sigmoid(self.a0 + (xa @ self.a1) @ self.a2)
        g = torch.sigmoid(xg @ self.g1) @ self.g2

        kk = k * self.k_k
        kk = F.normalize(kk.view(B, T, H, -1), dim=-1, p=2.0).view(B, T, C)
        k = k * (1 + (a - 1) * self.k_a)

        if state is None:
            x = RUN_CUDA_RWKV7g(r, w, k, v, -kk, kk * a)
        else:
            x = state.wkv(self.layer_id, r, w, k, v, -kk, kk * a)
        x = self.ln_x(x.view(B * T, C)).view(B, T, C)
        x = x + ((r.view(B, T, H, -1) * k.view(B, T, H, -1) * self.r_k).sum(dim=-1, keepdim=True)
                 * v.view(B, T, H, -1)).view(B, T, C)
        x = self.output(x * g)
        return x, v_first


class RWKV_CMix_x070(nn.Module):
    """Channel-mix FFN of RWKV-7 (src/model.py:200-227): shift-lerp -> C->4C -> relu^2 -> 4C->C."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        C = args.n_embd
        with torch.no_grad():
            r10 = 1.0 - (layer_id / args.n_layer)
            ddd = (torch.arange(C, dtype=torch.float32) / C).view(1, 1, C)
            self.x_k = nn.Parameter(1.0 - torch.pow(ddd, r10 ** 4))
        self.key = nn.Linear(C, C * 4, bias=False)
        self.value = nn.Linear(C * 4, C, bias=False)
        self.key.weight.data.uniform_(-0.5 / (C ** 0.5), 0.5 / (C ** 0.5))
        self.value.weight.data.zero_()

    def forward(self, x, state=None):
        if getattr(self.args, "fused", False) and x.is_cuda:
            from . import fused
            if state is None:
                return fused.cmix_forward(self, x)
            if not torch.is_grad_enabled() and x.dtype == torch.bfloat16:
                return fused.cmix_forward_stateful(self, x, state)
        if state is None:
            xx = time_shift(x) - x
        else:
            xx = torch.cat((state.ffn_x[self.layer_id].unsqueeze(1), x[:, :-1]), dim=1) - x
            state.ffn_x[self.layer_id].copy_(x[:, -1])
        k = x + xx * self.x_k
        k = torch.relu(self.key(k)) ** 2
        return self.value(k)


class Block(nn.Module):
    """Pre-LN residual block (src/model.py:233-254); block 0 also owns ln0."""

    def __init__(self, args, layer_id):
        super().__init__()
        self.args = args
        self.layer_id = layer_id
        if layer_id == 0:
            self.ln0 = nn.LayerNorm(args.n_embd)
        self.ln1 = nn.LayerNorm(args.n_embd)
        self.ln2 = nn.LayerNorm(args.n_embd)
        self.att = RWKV_Tmix_x070(args, layer_id)
        self.ffn = RWKV_CMix_x070(args, layer_id)

    def forward(self, x, v_first, state=None):
        if self.layer_id == 0:
            x = self.ln0(x)
        if state is None:
            xx, v_first = self.att(self.ln1(x), v_first)
            x = x + xx
            x = x + self.ffn(self.ln2(x))
        else:
            xx, v_first = self.att(self.ln1(x), v_first, state)
            x = x + xx
            x = x + self.ffn(self.ln2(x), state)
        return x, v_first


class L2Wrap(torch.autograd.Function):
    """Identity on the loss; adds 1e-4/(B*T) * max-logit at the arg-max to the logits' gradient
    (src/model.py:257-271)."""

    @staticmethod
    def forward(ctx, loss, y):
        ctx.save_for_backward(y)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        y = ctx.saved_tensors[0]
        factor = 1e-4 / (y.shape[0] * y.shape[1])
        maxx, ids = torch.max(y, -1, keepdim=True)
        gy = torch.zeros_like(y)
        gy.scatter_(-1, ids, maxx * factor)
        return grad_output, gy


class RWKV7State:
    """Recurrent state of an RWKV-7 stack for stateful generation (the reference re-runs the
    whole sequence per generated token, src/model.py:513-529).  Per layer: the last token fed to the time-mix and
    channel-mix shifts (the reference's ZeroPad2d shift sees zeros before the first token) and the WKV state
    S (B,H,64,64) fp32.  Whole 16-token chunks go through the chunked MFMA forward kernel continuing from S (cut into
    sequence-parallel segments when the heads alone cannot fill the chip); the ragged tail and single tokens are one
    `wkv7_step` launch each."""

    def __init__(self, args, batch, device, dtype=torch.bfloat16):
        L, C = args.n_layer, args.n_embd
        self.att_x = [torch.zeros(batch, C, device=device, dtype=dtype) for _ in range(L)]
        self.ffn_x = [torch.zeros(batch, C, device=device, dtype=dtype) for _ in range(L)]
        self.S = [torch.zeros(batch, args.dim_att // 64, 64, 64, device=device, dtype=torch.float32) for _ in range(L)]
        self.fresh = [True] * L
        self.n_tokens = 0

    def wkv(self, layer, r, w, k, v, z, b):
        from . import wkv7
        B, T, HC = r.shape
        ops = [i.view(B, T, HC // 64, 64) for i in (w, r, k, v, z, b)]      # the op's (w,q,k,v,z,a) order
        S, outs, t0 = self.S[layer], [], 0
        if T >= CHUNK_LEN:                              # whole chunks through the chunked MFMA kernel, from the carried state
            t0 = T // CHUNK_LEN * CHUNK_LEN
            head = [i[:, :t0].contiguous() for i in ops]
            y, s_end = wkv7.wkv7_forward_tparallel(*head, None if self.fresh[layer] else S)
            S.copy_(s_end)
            outs.append(y)
        for t in range(t0, T):                          # ragged tail / single tokens: one step launch each
            outs.append(wkv7.wkv7_step(*[i[:, t].contiguous() for i in ops], S).unsqueeze(1))
        self.fresh[layer] = False
        return (outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)).view(B, T, HC)


class RWKV(nn.Module):
    """Embedding -> n_layer Blocks -> ln_out -> head, on already-embedded inputs (src/model.py:273-325).

    `forward(x_emb)` left-pads T to a multiple of CHUNK_LEN with emb(STOP_TOKEN_INDEX) and strips the
    pad from the logits.  `args.grad_cp` is the reference's memory-saving switch (src/model.py:318-319: deepspeed.checkpointing.checkpoint per
    Block).  1 re-computes each Block in the backward, as the reference does (eager path: torch.utils.checkpoint, fused path: the same schedule
    through the fused kernels); 2 (not in the reference; fused path) = selective recompute -- WKV7 checkpoints + relu^2 dropped, every GEMM output
    kept: a third of the activation memory saved for ~4 % of the step; on the eager path 2 falls back to 1."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.emb = nn.Embedding(args.vocab_size, args.n_embd)
        self.blocks = nn.ModuleList([Block(args, i) for i in range(args.n_layer)])
        self.ln_out = nn.LayerNorm(args.n_embd)
        self.head = nn.Linear(args.n_embd, args.vocab_size, bias=False)
        if args.dropout > 0:
            self.drop0 = nn.Dropout(p=args.dropout)

    def pad_left(self, x, num_tokens_to_pad):
        if num_tokens_to_pad != 0:
            eos_idx = torch.full((x.size(0), num_tokens_to_pad), STOP_TOKEN_INDEX, dtype=torch.long, device=x.device)
            x = torch.cat((self.emb(eos_idx), x), dim=1)
        return x

    def unpad(self, x, num_tokens_to_pad):
        return x[:, num_tokens_to_pad:] if num_tokens_to_pad > 0 else x

    def forward_features(self, x):
        """Everything up to (not including) the head; returns (hidden, num_tokens_to_pad)."""
        args = self.args
        rem = x.size(1) % CHUNK_LEN
        num_tokens_to_pad = CHUNK_LEN - rem if rem != 0 else 0
        x = self.pad_left(x, num_tokens_to_pad)
        if args.dropout > 0:
            x = self.drop0(x)
        if getattr(args, "fused", False):
            from . import fused
            if fused.add_ln_supported(x):
                return fused.blocks_forward(self, x, grad_cp=int(args.grad_cp) if torch.is_grad_enabled() else 0), num_tokens_to_pad
        v_first = torch.empty_like(x)
        for block in self.blocks:
            if args.grad_cp >= 1 and torch.is_grad_enabled():
                from torch.utils.checkpoint import checkpoint
                x, v_first = checkpoint(block, x, v_first, use_reentrant=False)
            else:
                x, v_first = block(x, v_first)
        return self.ln_out(x), num_tokens_to_pad

    def forward(self, x):
        x, num_tokens_to_pad = self.forward_features(x)
        if x.is_cuda and getattr(self.args, "fused", False):
            from . import fused
            x = fused.linear(self.head, x)              # input gradient in the forward GEMMs' layout
        else:
            x = self.head(x)
        return self.unpad(x, num_tokens_to_pad)

    @torch.no_grad()
    def forward_stateful(self, x, state=None, last_only=False):
        """Inference on embedded tokens x (B,T,C) continuing from `state` (None: empty context).  No padding is
        added: logits equal those `forward` gives for the same absolute positions of the concatenated sequence.
        Returns (logits (B,T,V) or (B,V) with last_only, state)."""
        if state is None:
            state = RWKV7State(self.args, x.size(0), x.device, x.dtype)
        v_first = torch.empty_like(x)
        use_decode = False
        if getattr(self.args, "fused", False) and x.shape[1] == 1:
            from . import decode
            use_decode = decode.supported(x)
        for block in self.blocks:
            if use_decode:                               # single token: batched-GEMV step (decode.py)
                x, v_first = decode.block_decode(block, x, v_first, state)
                state.fresh[block.layer_id] = False
            else:
                x, v_first = block(x, v_first, state)
        state.n_tokens += x.size(1)
        if use_decode and self.head.weight.dtype == torch.bfloat16:
            logits = decode.head_decode(self, x)
            return (logits if last_only else logits.unsqueeze(1)), state
        if last_only:
            x = x[:, -1]
        return self.head(self.ln_out(x)), state

    def make_decoder(self, state):
        """A single-token decode step captured in a HIP graph (the eager step is ~1000 small launches for 24 layers and
        is bound by launch overhead, not by the GPU).  Returns `step(x_emb (B,1,C)) -> logits (B,V)`; `state` is
        advanced in place by every call, exactly as `forward_stateful(x_emb, state, last_only=True)` would."""
        return GraphDecoder(self, state)

    def decoder_for(self, state):
        """`make_decoder` with the captured graph re-used across prompts: one capture per (batch size, device, parameter
        versions); later calls copy `state` into the graph's own state tensors.  The returned decoder advances ITS
        state (`decoder.state`), not the argument."""
        S0 = state.S[0]
        from . import param_state
        key = (S0.shape[0], S0.device, sum(p._version for p in self.parameters()), param_state.generation())
        cache = self.__dict__.setdefault("_decoders", {})
        dec = cache.get(key)
        if dec is None:
            cache.clear()                                 # parameters changed (or first use): stale graphs go
            dec = cache[key] = GraphDecoder(self, state)
        elif dec.state is not state:
            dec.load_state(state)
        return dec


class GraphDecoder:
    def __init__(self, rwkv: "RWKV", state: RWKV7State):
        self.rwkv, self.state = rwkv, state
        p = next(rwkv.parameters())
        B = state.S[0].shape[0]
        self.x_in = torch.zeros(B, 1, rwkv.args.n_embd, device=p.device, dtype=p.dtype)
        keep = [t.clone() for t in state.att_x + state.ffn_x + state.S]       # warm-up and capture run the step for real
        fresh, n_tok = list(state.fresh), state.n_tokens
        side = torch.cuda.Stream(device=p.device)
        side.wait_stream(torch.cuda.current_stream(p.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                rwkv.forward_stateful(self.x_in, state, last_only=True)
        torch.cuda.current_stream(p.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits, _ = rwkv.forward_stateful(self.x_in, state, last_only=True)
        for dst, src in zip(state.att_x + state.ffn_x + state.S, keep):
            dst.copy_(src)
        state.fresh, state.n_tokens = fresh, n_tok

    @torch.no_grad()
    def load_state(self, other: RWKV7State):
        """Continue from another prompt's state: copied into the tensors the graph was captured on."""
        mine = self.state
        for dst, src in zip(mine.att_x + mine.ffn_x + mine.S, other.att_x + other.ffn_x + other.S):
            dst.copy_(src)
        mine.fresh, mine.n_tokens = list(other.fresh), other.n_tokens

    @torch.no_grad()
    def __call__(self, x_emb):
        self.x_in.copy_(x_emb)
        self.graph.replay()
        self.state.n_tokens += 1
        self.state.fresh = [False] * len(self.state.fresh)
        return self.logits
