"""Data-parallel training engine with ZeRO-1 semantics for one node of MI355X (RCCL over xGMI).

What the reference delegates to Lightning's DeepSpeedStrategy (`--strategy deepspeed_stage_1`, bf16,
bucket size `ds_bucket_mb`, FusedAdam, gradient_clip_val=1.0; VisualRWKV-v7/v7.00/train.py:55,75-76,92,
214-216 and src/model.py:390-410) is restated here MI355X-first:

* one process per GPU; the model replica keeps its trainable parameters and gradients as views into two
  flat bf16 buffers (weight-decayed tensors first, then the < 2-D ones -- src/model.py:391-393);
* the flat buffer is cut into buckets (default 200 MB like `ds_bucket_mb`, train.py:55); as soon as the
  backward has produced every gradient of a bucket, that bucket is reduce-scattered on a side HIP stream
  (one RCCL reduce-scatter per bucket: every GPU sends 1/W of the bucket to each peer, so all seven xGMI
  links of the mesh carry traffic, instead of a ring all-reduce that is bound by one link), overlapping the
  remaining WKV7/GEMM backward, which launches on the compute stream (the WKV op uses the current stream);
* each rank owns piece r of every bucket: fp32 master weights + Adam moments live only for that piece
  (optimizer state sharded W ways);  gradient clipping needs one scalar all-reduce of the squared norm;
* fused AdamW (HIP kernel, vrwkv_adamw_step_bf16) updates the piece and writes the bf16 parameters, which
  are all-gathered back bucket by bucket.

With world_size == 1 the collectives disappear and the same code path runs.  On CPU/gloo (tests) the
reduce-scatter is emulated with all-reduce + slice and the optimizer uses a torch restatement of the kernel.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.distributed as dist


def lr_wd_schedule(real_step: int, lr_init: float, lr_final: float, warmup_steps: int, epoch_begin: int,
                   epoch_count: int, epoch_steps: int, weight_decay: float = 0.0, weight_decay_final: float = -1.0):
    """LR / weight-decay schedule of the reference's train_callback.on_train_batch_start
    (src/trainer.py:24-49): cosine decay from lr_init to lr_final over (epoch_begin+epoch_count)*epoch_steps
    steps with progress = (step - warmup + 1)/(total - warmup), multiplied by (0.1 + 0.9 step/warmup) during
    warm-up; exponential weight-decay interpolation when weight_decay_final > 0.  Returns (lr, wd_now).
    (Quirk kept by callers that mirror the reference: it writes `lr` only into the param groups whose
    weight_decay is 0 and `wd_now` into the others, trainer.py:45-49.)"""
    progress = 0.0
    if lr_final == lr_init or epoch_count == 0:
        lr = lr_init
    else:
        decay_total = (epoch_begin + epoch_count) * epoch_steps
        progress = (real_step - warmup_steps + 1) / (decay_total - warmup_steps)
        progress = min(1, max(0, progress))
        cosine_decay = max(0.0, 0.5 * (1 + math.cos(math.pi * progress)))
        lr = lr_final + (lr_init - lr_final) * cosine_decay
    if real_step < warmup_steps:
        lr = lr * (0.1 + 0.9 * real_step / warmup_steps)
    if weight_decay_final > 0:
        wd_now = weight_decay * math.exp(math.log(weight_decay_final / weight_decay) * progress)
    else:
        wd_now = weight_decay
    return lr, wd_now


class _Bucket:
    __slots__ = ("start", "end", "piece", "pending", "n_params", "master", "m", "v", "work", "event", "param_ids", "launched")


class Zero1Engine:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-4, betas=(0.9, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.0, grad_clip: float = 1.0, bucket_mb: float = 200.0,
                 process_group=None, overlap: bool = True, force_collectives: bool = False, async_gather: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.grad_clip = grad_clip
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.step_count = 0
        # force_collectives: run the full hook / side-stream / reduce-scatter / all-gather path even with one
        # rank (used to exercise the RCCL code path on a single-GPU box)
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        if self.world > 1:
            from . import gemm_tuning
            gemm_tuning.note_collectives(True)      # stream-K library GEMMs on two streams beside RCCL kernels: never validated, so not done (gemm_tuning.py)
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "nothing to train"
        self.device = params[0].device
        self.dtype = params[0].dtype
        self.on_gpu = self.device.type == "cuda"
        self.overlap = overlap and self.on_gpu and self.collective
        # async_gather=False (default): step() returns with the updated parameters visible to the compute stream -- anything
        # may read them (state_dict / torch.save, the decode path's raw-pointer GEMV kernels and cached transposed weights,
        # graph capture).  async_gather=True: step() returns while the all-gather still runs on the communication stream and
        # only module forwards (pre-hooks) wait for it; every other reader must call wait_params() first.
        self.async_gather = bool(async_gather) and self.overlap
        # Lay the flat buffer out in the order the backward produces the gradients, so that buckets fill front to back:
        # reverse registration order for the language model (head, blocks N-1 .. 0), and whatever feeds the language model's
        # INPUT (the image projector `proj`, the embedding) last -- their gradients only exist when the backward has walked
        # the whole stack.  (Plain reverse registration order put `proj`, which is registered last, at the front of bucket 0
        # and kept that bucket from being reduced until the very end of the backward.)
        names = {id(p): n for n, p in model.named_parameters()}
        late = lambda p: names.get(id(p), "").startswith(("proj.", "rwkv.emb.", "emb."))
        by_ready = [p for p in params[::-1] if not late(p)] + [p for p in params[::-1] if late(p)]
        wd = [p for p in by_ready if len(p.squeeze().shape) >= 2]
        nowd = [p for p in by_ready if len(p.squeeze().shape) < 2]
        ordered = wd + nowd
        align = 8 * self.world                       # every piece 16-byte aligned in bf16
        offs, total = [], 0
        for p in ordered:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        self.wd_boundary = sum((p.numel() + 7) // 8 * 8 for p in wd)
        bucket_elems = max(int(bucket_mb * 1e6) // 2 // align * align, align)
        total = (total + align - 1) // align * align
        self.numel = total
        self.flat_param = torch.zeros(total, dtype=self.dtype, device=self.device)
        self.flat_grad = torch.zeros(total, dtype=self.dtype, device=self.device)
        self.params = ordered
        self.offsets = offs
        with torch.no_grad():
            for p, o in zip(ordered, offs):
                self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + p.numel()].view_as(p)
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                # where this parameter's gradient lives: a backward that owns its weight-gradient GEMM (fused._LinearTN) writes it
                # there directly and returns that view, which the hook below recognises by its address -- no copy into the bucket
                p._vrwkv_flat_grad = (self.flat_grad, o)
        # buckets
        self.buckets: List[_Bucket] = []
        s = 0
        while s < total:
            b = _Bucket()
            b.start, b.end = s, min(s + bucket_elems, total)
            b.piece = (b.end - b.start) // self.world
            ps = b.start + self.rank * b.piece
            b.master = self.flat_param[ps:ps + b.piece].float().clone()
            b.m = torch.zeros_like(b.master)
            b.v = torch.zeros_like(b.master)
            b.n_params, b.pending, b.work, b.event, b.launched = 0, 0, None, None, False
            b.param_ids = []
            self.buckets.append(b)
            s = b.end
        # which buckets does each parameter touch
        self._param_buckets = []
        for p, o in zip(ordered, offs):
            first = o // bucket_elems
            last = (o + max(p.numel(), 1) - 1) // bucket_elems
            idx = list(range(first, min(last, len(self.buckets) - 1) + 1))
            self._param_buckets.append(idx)
            for i in idx:
                self.buckets[i].n_params += 1
                self.buckets[i].param_ids.append(len(self._param_buckets) - 1)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # Gradients reach the flat buffer in one of two ways.  With `.grad` attached to the flat views autograd
        # accumulates in place: one small add kernel per parameter (~1200 per step for the 1.5B model) on top of
        # the memset of zero_grad().  After zero_grad(set_to_none=True) autograd instead hands over the freshly
        # computed gradient tensors (no kernel); the hook stashes them and a bucket is filled by ONE multi-tensor
        # copy when its last parameter arrives.
        self._stash = [None] * len(ordered)
        self._fired = [False] * len(ordered)
        self._gloo_ok = {}
        self._hooks = []
        for k, p in enumerate(ordered):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(k)))
        self._sq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._gather = False
        self._next_launch = 0
        self._gather_event = None
        self.generation = 0
        # comm_timing (bench.py, N > 1): when a list, every collective is bracketed by timing events on the communication stream and every
        # point where the COMPUTE stream waits for communication by timing events on the compute stream; comm_report() turns them into
        # "how long did the collectives run" and "how much of that did the compute stream stand still for" (the exposed part)
        self.comm_timing = None
        self._wait_hooks = []
        if self.async_gather:
            # the parameter all-gather runs on the side stream; whoever first touches a trainable parameter in the next
            # forward waits for it (the frozen ViT encode ahead of the projector overlaps with it).  Hooks sit on the model, its
            # children and grandchildren that hold trainable parameters (`rwkv`, `rwkv.emb`, `rwkv.head`, `proj`, `proj.gate` ..):
            # about ten Python calls per step instead of one per leaf module.  Calling a deeper sub-module directly (e.g.
            # `model.rwkv.blocks[3](x)`) right after step() needs an explicit wait_params().
            def trainable(mod):
                return any(p.requires_grad for p in mod.parameters())
            kids = [c for c in model.children() if trainable(c)]
            own = any(p.requires_grad for p in model.parameters(recurse=False))
            hooked = ([model] if own or not kids else []) + kids + [g for c in kids for g in c.children() if trainable(g)]
            for mod in hooked:
                self._wait_hooks.append(mod.register_forward_pre_hook(lambda m, a: self.wait_params()))
        self._reset_pending()

    # ------------------------------------------------------------------ gradient reduction
    def _reset_pending(self):
        for b in self.buckets:
            b.pending = b.n_params
            b.work, b.event, b.launched = None, None, False
        self._fired = [False] * len(self.params)
        self._next_launch = 0

    def _view(self, k):
        o, p = self.offsets[k], self.params[k]
        return self.flat_grad[o:o + p.numel()].view_as(p)

    def _make_hook(self, k):
        def hook(param):
            g = param.grad
            o = self.offsets[k]
            param._vrwkv_wgrad_pending = False          # see fused._LinearTN.backward
            if g.data_ptr() != self.flat_grad.data_ptr() + o * self.flat_grad.element_size():
                self._stash[k] = g                   # autograd handed over a fresh tensor: copied bucket-wise
            self._fired[k] = True
            for i in self._param_buckets[k]:
                self.buckets[i].pending -= 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        """Reduce complete buckets strictly in index order: every rank issues the same sequence of collectives, whatever
        order its own gradients arrive in (a rank whose batch leaves a parameter without gradient -- e.g. no image
        placeholder, so no `proj` gradient -- completes that bucket only in step(), and must not overtake it with later
        buckets: NCCL pairs collectives by issue order)."""
        while self._next_launch < len(self.buckets) and self.buckets[self._next_launch].pending == 0:
            b = self.buckets[self._next_launch]
            self._flush(b)
            self._launch_reduce(b)
            b.launched = True
            self._next_launch += 1

    @torch.no_grad()
    def _flush(self, b: _Bucket):
        """Move the stashed gradients of bucket `b` into the flat buffer (one multi-tensor copy) and re-attach the
        flat views as `.grad`."""
        ks = [k for k in b.param_ids if self._stash[k] is not None]
        if not ks:
            return
        views = [self._view(k) for k in ks]
        srcs = [self._stash[k] for k in ks]
        if len(ks) == 1:
            views[0].copy_(srcs[0])
        else:
            torch._foreach_copy_(views, srcs)
        for k, v in zip(ks, views):
            self.params[k].grad = v
            self._stash[k] = None

    def _launch_reduce(self, b: _Bucket):
        if not self.collective:
            return
        buf = self.flat_grad[b.start:b.end]
        if self.overlap:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self.comm_stream.wait_event(ready)
            with torch.cuda.stream(self.comm_stream):
                t0 = self._mark(self.comm_stream)
                self._reduce_scatter(buf, b)
                self._mark_end("reduce_scatter", t0, self.comm_stream)
                b.event = torch.cuda.Event()
                b.event.record(self.comm_stream)
        else:
            self._reduce_scatter(buf, b)

    def _mark(self, stream):
        if self.comm_timing is None or not self.on_gpu:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def _mark_end(self, kind, t0, stream):
        if t0 is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            self.comm_timing.append((kind, t0, e))

    def comm_report(self, steps: int):
        """Per-step averages (ms) over the events collected since comm_timing was set to a list: time the collectives occupied the
        communication stream, and time the compute stream stood still waiting for them (after the backward for the last reduce-scatters,
        before the first trainable module of the next forward for the all-gather).  Synchronises."""
        if not self.comm_timing:
            return None
        torch.cuda.synchronize(self.device)
        tot = {}
        for kind, a, b in self.comm_timing:
            tot[kind] = tot.get(kind, 0.0) + a.elapsed_time(b)
        out = {k + "_ms_per_step": v / max(steps, 1) for k, v in tot.items()}
        out["exposed_ms_per_step"] = (tot.get("wait_reduce_scatter", 0.0) + tot.get("wait_all_gather", 0.0)) / max(steps, 1)
        out["what"] = ("reduce_scatter / all_gather: time on the communication stream (includes waiting for the slowest rank); wait_*: time the "
                       "compute stream stood still for them = the exposed communication")
        return out

    def _reduce_scatter(self, buf, b: _Bucket):
        piece = buf[self.rank * b.piece:(self.rank + 1) * b.piece]
        backend = dist.get_backend(self.pg)
        if backend == "nccl":
            dist.reduce_scatter_tensor(piece, buf, op=dist.ReduceOp.SUM, group=self.pg)   # in place on own piece
        elif self._gloo_native_dtype(buf.dtype):   # gloo has no reduce-scatter: all-reduce in the buffer's own dtype (bf16 sums in bf16, like RCCL), keep the own slice
            dist.all_reduce(buf, group=self.pg)
        else:                                    # a gloo build without reductions in this dtype: sum in fp32, round once
            wide = buf.float()
            dist.all_reduce(wide, group=self.pg)
            buf.copy_(wide)

    def _gloo_native_dtype(self, dtype) -> bool:
        """Whether this gloo build reduces `dtype` -- probed ONCE per dtype with a one-element all-reduce that every rank issues at the
        same point (the first bucket of the first step), never by catching errors around a real collective: a communication failure
        there must surface, not be retried."""
        ok = self._gloo_ok.get(dtype)
        if ok is None:
            try:
                dist.all_reduce(torch.zeros(1, dtype=dtype), group=self.pg)
                ok = True
            except RuntimeError:
                ok = False
            self._gloo_ok[dtype] = ok
        return ok

    # ------------------------------------------------------------------ optimizer step
    @torch.no_grad()
    def step(self, lr: Optional[float] = None):
        """Finish gradient reduction, clip to `grad_clip` (global L2 norm), AdamW on the owned pieces, publish
        the updated bf16 parameters.  Returns the pre-clip global gradient norm as a 0-dim float32 tensor on the engine's
        device on every path (float() of it synchronises; the step itself does not)."""
        lr = self.lr if lr is None else lr
        self.step_count += 1
        for p in self.params:                        # until the next zero_grad a backward gets fresh gradient tensors again (a caller of
            p._vrwkv_flat_armed = False              # torch.autograd.grad must never be handed an alias of the flat buffer)
        self.wait_params()
        # Buckets whose hooks never all fired (unused parameters), in index order.  The slots of the unused parameters are zeroed
        # FIRST, all of them: a parameter may span two buckets, and zeroing its whole view while launching the second would wipe
        # the part the first bucket's reduction has already written.
        if self._gather:
            for b in self.buckets[self._next_launch:]:
                if b.pending > 0:
                    for k in b.param_ids:
                        if not self._fired[k]:
                            v = self._view(k)
                            v.zero_()
                            self.params[k].grad = v
                            self._fired[k] = True
        for b in self.buckets[self._next_launch:]:
            self._flush(b)
            self._launch_reduce(b)
            b.launched = True
        self._next_launch = len(self.buckets)
        if self.collective:
            if self.overlap:
                cur = torch.cuda.current_stream(self.device)
                t0 = self._mark(cur)
                for b in self.buckets:
                    if b.event is not None:
                        cur.wait_event(b.event)
                self._mark_end("wait_reduce_scatter", t0, cur)
        inv_world = 1.0 / self.world
        # global gradient norm over the owned pieces (each element is owned by exactly one rank)
        self._sq.zero_()
        for b in self.buckets:
            g = self._piece(self.flat_grad, b)
            self._sqnorm(g, self._sq)
        if self.collective:
            dist.all_reduce(self._sq, group=self.pg)
        clip = float(self.grad_clip) if self.grad_clip and self.grad_clip > 0 else 0.0
        if self.on_gpu and self.dtype == torch.bfloat16:
            for b in self.buckets:               # clip factor formed on the device from self._sq: no host synchronisation
                self._adamw(b, lr, None, inv_world, clip)
            gnorm = (self._sq.sqrt() * inv_world).reshape(())
        else:
            gn = float(self._sq.sqrt()) * inv_world
            gnorm = torch.tensor(gn, dtype=torch.float32, device=self.device)
            scale = inv_world * (min(1.0, clip / (gn + 1e-6)) if clip > 0 else 1.0)
            for b in self.buckets:
                self._adamw(b, lr, scale, inv_world, clip)
        if self.collective:
            if self.overlap:                      # publish on the side stream; the next forward waits where it needs them
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                self.comm_stream.wait_event(done)
                with torch.cuda.stream(self.comm_stream):
                    t0 = self._mark(self.comm_stream)
                    self._all_gather()
                    self._mark_end("all_gather", t0, self.comm_stream)
                    self._gather_event = torch.cuda.Event()
                    self._gather_event.record(self.comm_stream)
                if not self.async_gather:
                    self.wait_params()
            else:
                self._all_gather()
        from . import param_state
        self.generation = param_state.bump()      # decode caches (transposed factors, captured graphs) key on this
        self._reset_pending()
        return gnorm

    def _all_gather(self):
        for b in self.buckets:
            buf = self.flat_param[b.start:b.end]
            piece = buf[self.rank * b.piece:(self.rank + 1) * b.piece]
            if dist.get_backend(self.pg) == "nccl":
                dist.all_gather_into_tensor(buf, piece, group=self.pg)          # in place (own slot = input)
            else:   # gloo (CPU tests): bit-cast to int16, list form
                raw = buf.view(torch.int16) if buf.dtype == torch.bfloat16 else buf
                dist.all_gather(list(raw.chunk(self.world)), raw[self.rank * b.piece:(self.rank + 1) * b.piece].clone(), group=self.pg)

    def wait_params(self):
        """Make the current stream wait for the parameter all-gather of the last step (no-op when none is pending)."""
        ev, self._gather_event = self._gather_event, None
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            t0 = self._mark(cur)
            cur.wait_event(ev)
            self._mark_end("wait_all_gather", t0, cur)

    def close(self):
        """Detach the engine from its parameters: weight-gradient GEMMs stop writing into this engine's flat buffer (a model that
        outlives its engine, or gets a new one, must not keep the old buffer alive through its parameters)."""
        for h in self._hooks + self._wait_hooks:      # or every later backward / forward of the model would still drive THIS engine
            h.remove()
        self._hooks, self._wait_hooks = [], []
        self._stash = [None] * len(self._stash)
        for p in self.params:
            for attr in ("_vrwkv_flat_grad", "_vrwkv_flat_armed", "_vrwkv_wgrad_pending"):
                if hasattr(p, attr):
                    delattr(p, attr)

    def zero_grad(self, set_to_none: bool = True):
        """set_to_none (default): detach `.grad` so that the next backward hands its gradient tensors over instead
        of adding into the flat buffer (no memset, no per-parameter add kernels; one backward per step).
        set_to_none=False: zero the flat buffer and keep accumulating in place."""
        self._gather = set_to_none
        if set_to_none:
            for p in self.params:
                p.grad = None
                p._vrwkv_wgrad_pending = False
                p._vrwkv_flat_armed = True          # this engine expects ONE backward into its flat buffer before the next step()
        else:
            self.flat_grad.zero_()
            for k, p in enumerate(self.params):
                p.grad = self._view(k)

    def _piece(self, flat, b: _Bucket):
        s = b.start + self.rank * b.piece
        return flat[s:s + b.piece]

    def _sqnorm(self, g, out):
        if self.on_gpu and g.dtype == torch.bfloat16 and g.numel() % 8 == 0:
            from . import hip_lib
            rc = hip_lib.load().vrwkv_sqnorm_bf16(g.numel(), g.data_ptr(), out.data_ptr(), hip_lib.launch_stream(self.device))
            hip_lib.check(rc, "vrwkv_sqnorm_bf16")
        else:
            out += g.float().pow(2).sum()

    def _adamw(self, b: _Bucket, lr, scale, inv_world=1.0, clip=0.0):
        g = self._piece(self.flat_grad, b)
        p = self._piece(self.flat_param, b)
        off = b.start + self.rank * b.piece
        b1, b2 = self.betas
        if self.on_gpu and g.dtype == torch.bfloat16:
            from . import hip_lib
            st = hip_lib.launch_stream(self.device)
            if scale is None:
                rc = hip_lib.load().vrwkv_adamw_step_clip_bf16(
                    b.piece, b.master.data_ptr(), b.m.data_ptr(), b.v.data_ptr(), g.data_ptr(), p.data_ptr(),
                    lr, b1, b2, self.eps, self.weight_decay, self.step_count, self._sq.data_ptr(), inv_world, clip, off,
                    self.wd_boundary, st)
                hip_lib.check(rc, "vrwkv_adamw_step_clip_bf16")
                return
            rc = hip_lib.load().vrwkv_adamw_step_bf16(
                b.piece, b.master.data_ptr(), b.m.data_ptr(), b.v.data_ptr(), g.data_ptr(), p.data_ptr(),
                lr, b1, b2, self.eps, self.weight_decay, self.step_count, scale, off, self.wd_boundary, st)
            hip_lib.check(rc, "vrwkv_adamw_step_bf16")
            return
        # host restatement of the kernel (CPU tests of the distributed logic)
        gf = g.float() * scale
        b.m.mul_(b1).add_(gf, alpha=1 - b1)
        b.v.mul_(b2).addcmul_(gf, gf, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
        idx = torch.arange(off, off + b.piece, device=g.device)
        decay = torch.where(idx < self.wd_boundary, self.weight_decay, 0.0).to(torch.float32)
        upd = (b.m / bc1) / ((b.v / bc2).sqrt() + self.eps) + decay * b.master
        b.master.add_(upd, alpha=-lr)
        p.copy_(b.master.to(p.dtype))


def largest_3n_plus_2_prime(x: int) -> int:
    """Largest prime p <= x with p % 3 == 2 (src/utils.py:29-45): cubing is then a bijection mod p."""
    def is_prime(n):
        if n < 2:
            return False
        i = 2
        while i * i <= n:
            if n % i == 0:
                return False
            i += 1
        return True
    for p in range(x, 1, -1):
        if p % 3 == 2 and is_prime(p):
            return p
    return -1


def rank_strided_sample(epoch: int, idx: int, rank: int, world: int, samples_per_epoch: int, magic_prime: int):
    """Deterministic sample choice of the reference dataset (src/dataset.py:182-195):
    step = epoch*samples_per_epoch + idx*world + rank ; index = step^3 mod magic_prime; the second pass
    (step >= magic_prime) reads the reversed list.  Returns (index, use_reversed_list)."""
    step = epoch * samples_per_epoch + idx * world + rank
    return (step * step * step) % magic_prime, step >= magic_prime
