"""ZeRO-1 engine on CPU: single process against torch.optim.AdamW, and world_size 2 over gloo against the
single-process result on the concatenated batch (the N>1 path of bench.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(seed=0):
    torch.manual_seed(seed)
    m = nn.Sequential(nn.Linear(24, 40), nn.LayerNorm(40), nn.Tanh(), nn.Linear(40, 8))
    return m


def _data(n=16, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 24, generator=g), torch.randn(n, 8, generator=g)


def _loss(m, x, y):
    return ((m(x) - y) ** 2).mean()


def test_single_process_matches_torch_adamw():
    from visualrwkv_amd.dp import Zero1Engine
    ref = _model()
    mine = _model()
    x, y = _data()
    wd_params = [p for p in ref.parameters() if len(p.squeeze().shape) >= 2]
    no_wd = [p for p in ref.parameters() if len(p.squeeze().shape) < 2]
    opt = torch.optim.AdamW([{"params": wd_params, "weight_decay": 0.1}, {"params": no_wd, "weight_decay": 0.0}],
                            lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    eng = Zero1Engine(mine, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1, grad_clip=1.0, bucket_mb=0.001)
    assert len(eng.buckets) > 1
    for _ in range(5):
        opt.zero_grad(); _loss(ref, x, y).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        eng.zero_grad(); _loss(mine, x, y).backward(); eng.step()
    for a, b in zip(ref.parameters(), mine.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_a_closed_engine_no_longer_drives_the_model():
    """close() removes the engine's autograd / forward hooks: a model that gets a NEW engine must not keep feeding
    the old one's flat buffer on every backward."""
    from visualrwkv_amd.dp import Zero1Engine
    m = _model()
    x, y = _data()
    old = Zero1Engine(m, lr=1e-2, bucket_mb=0.001)
    old.zero_grad(); _loss(m, x, y).backward(); old.step()
    old.close()
    assert old._hooks == [] and old._wait_hooks == []
    snap = old.flat_grad.clone() if hasattr(old, "flat_grad") else None
    fired_before = list(old._fired)
    new = Zero1Engine(m, lr=1e-2, bucket_mb=0.001)
    new.zero_grad(); _loss(m, x, y).backward()
    assert old._fired == fired_before and all(g is None for g in old._stash)      # the old engine saw nothing of this backward
    if snap is not None:
        assert torch.equal(old.flat_grad, snap)
    new.step()
    new.close()


def test_gradient_gather_modes_agree_and_unused_parameters_get_zero():
    """zero_grad(set_to_none=True) (bucket-wise multi-tensor gather of autograd's gradient tensors) and
    set_to_none=False (in-place accumulation into the zeroed flat buffer) give the same update; a parameter that
    received no gradient in a step is treated as zero gradient, not as last step's."""
    from visualrwkv_amd.dp import Zero1Engine

    class WithUnused(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = _model()
            self.unused = nn.Parameter(torch.ones(7, 5))
            self.use_it = True

        def forward(self, x):
            y = self.body(x)
            return y + self.unused.sum() * 1e-3 if self.use_it else y

    x, y = _data()
    results = []
    for set_to_none in (True, False):
        torch.manual_seed(0)
        m = WithUnused()
        eng = Zero1Engine(m, lr=1e-2, weight_decay=0.0, grad_clip=0.0, bucket_mb=0.001)
        for step in range(4):
            m.use_it = step == 0                      # only the first step produces a gradient for `unused`
            eng.zero_grad(set_to_none=set_to_none)
            _loss(m, x, y).backward()
            eng.step()
            if step >= 1:
                assert float(m.unused.grad.abs().sum()) == 0.0
            for p in m.parameters():                 # .grad is (again) a view of the flat buffer
                assert p.grad.data_ptr() >= eng.flat_grad.data_ptr()
        results.append([p.detach().clone() for p in m.parameters()])
    for a, b in zip(*results):
        assert torch.equal(a, b)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visualrwkv_amd.dp import Zero1Engine
    m = _model()
    x, y = _data()
    xs, ys = x[rank::world], y[rank::world]                 # rank-strided shard of the global batch
    eng = Zero1Engine(m, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1, grad_clip=1.0, bucket_mb=0.001)
    for _ in range(4):
        eng.zero_grad(); _loss(m, xs, ys).backward(); eng.step()
    torch.save([p.detach().clone() for p in m.parameters()], os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_world2_gloo_matches_single_process(tmp_path):
    from visualrwkv_amd.dp import Zero1Engine
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt"); r1 = torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)                            # replicas stay identical
    m = _model()
    x, y = _data()
    eng = Zero1Engine(m, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1, grad_clip=1.0, bucket_mb=0.001)
    for _ in range(4):
        eng.zero_grad(); _loss(m, x, y).backward(); eng.step()
    for a, b in zip(r0, m.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_schedule_and_sampler_match_reference_formulas():
    from visualrwkv_amd.dp import largest_3n_plus_2_prime, lr_wd_schedule, rank_strided_sample
    # values of src/trainer.py:24-38 for the shipped 0.1B script (lr 6e-4 -> 6e-5... here generic numbers)
    lr0, _ = lr_wd_schedule(0, 6e-4, 6e-5, 100, 0, 10, 1000)
    assert lr0 == pytest.approx(6e-4 * 0.1, rel=1e-3)
    lr_mid, _ = lr_wd_schedule(5000, 6e-4, 6e-5, 100, 0, 10, 1000)
    assert lr_mid == pytest.approx(6e-5 + (6e-4 - 6e-5) * 0.5 * (1 + __import__("math").cos(__import__("math").pi * (5000 - 100 + 1) / 9900)), rel=1e-9)
    lr_end, _ = lr_wd_schedule(10000, 6e-4, 6e-5, 100, 0, 10, 1000)
    assert lr_end == pytest.approx(6e-5)
    assert largest_3n_plus_2_prime(665298) == 665279 and largest_3n_plus_2_prime(10) == 5
    idx, rev = rank_strided_sample(0, 3, 1, 8, 8000, 665279)
    assert idx == (25 ** 3) % 665279 and rev is False


def test_buckets_are_reduced_in_index_order_whatever_order_gradients_arrive():
    """NCCL pairs collectives by issue order: every rank must reduce bucket 0, 1, 2, ... even when its own gradients
    complete the buckets in another order (e.g. a rank whose batch has no image placeholder gets no `proj` gradient)."""
    from visualrwkv_amd.dp import Zero1Engine
    orders = []
    for perm_seed in (0, 1, 2):
        torch.manual_seed(0)
        m = _model()
        eng = Zero1Engine(m, lr=1e-2, weight_decay=0.0, grad_clip=0.0, bucket_mb=0.0005)
        assert len(eng.buckets) >= 4
        launched = []
        eng._launch_reduce = lambda b, rec=launched, e=eng: rec.append(e.buckets.index(b))
        eng.zero_grad(set_to_none=False)
        g = torch.Generator().manual_seed(perm_seed)
        order = torch.randperm(len(eng.params), generator=g).tolist()
        skip = order[0] if perm_seed == 2 else None            # one parameter without gradient on this "rank"
        for k in order:
            if k == skip:
                continue
            eng.params[k].grad = eng._view(k)
            eng._hooks[k]  # (hooks are registered; call the hook body directly)
            eng._make_hook(k)(eng.params[k])
        eng.step()
        orders.append(launched)
    assert all(o == list(range(len(o))) for o in orders) and len({len(o) for o in orders}) == 1


# ---- N = 4 over gloo with the real model and the real hooks: one rank's batch has no image ----------------------------------
# (src/dataset.py:214-226: a text-only record carries no image placeholder; on that rank the projector gets NO gradient, so its
# autograd hooks never fire there while the three other ranks reduce the projector's bucket.)

def _tiny_visual():
    import bench
    from visualrwkv_amd.visual import VisualRWKV
    args = bench.build_args("tiny", 48, 16, ("siglip",), 0, False)
    torch.manual_seed(7)
    m = VisualRWKV(args)
    with torch.no_grad():
        for _, p in m.rwkv.named_parameters():
            if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.01)
    m.freeze_emb()
    return m


def _rank_batch(rank):
    import bench
    b = bench.synthetic_batch(1, 48, 16, ("siglip",), torch.device("cpu"), seed=500 + rank, side=56, dtype=torch.float32)
    if rank == 3:                                            # the text-only record: no placeholder run, no pixels
        g = torch.Generator().manual_seed(77)
        b["input_ids"][:, 4:20] = torch.randint(0, 65535, (1, 16), generator=g)
        del b["images"]
    return b


# eps = 1: the update is (nearly) linear in the gradient, so that the comparison below is one of gradients -- with the usual 1e-8 a
# gradient that is zero up to rounding (LayerNorm biases in front of a shift-invariant op) becomes a full +-lr step of either sign
_ENG4 = dict(lr=0.5, betas=(0.9, 0.99), eps=1.0, weight_decay=0.0, grad_clip=1.0, bucket_mb=0.05)


def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visualrwkv_amd.dp import Zero1Engine
    m = _tiny_visual()
    eng = Zero1Engine(m, **_ENG4)
    batch = _rank_batch(rank)
    init = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    fired = []
    for _ in range(2):
        eng.zero_grad()
        m.training_step(batch).backward()
        fired.append(sorted(n for n, p in m.named_parameters() if n.startswith("proj.") and p.grad is not None and float(p.grad.abs().sum()) > 0))
        eng.step()
    torch.save({"params": {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, "init": init, "proj_grads_step0": fired[0],
                "n_buckets": len(eng.buckets), "bucket_bounds": [(b.start, b.end, b.piece) for b in eng.buckets], "numel": eng.numel},
               os.path.join(out_dir, f"r{rank}.pt"))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("W", [4, 8])                      # 8 = the node the bench line is quoted for (BASELINE config 3: DP over 8 GPUs)
def test_world4_gloo_with_a_rank_without_image(tmp_path, W):
    from visualrwkv_amd.dp import Zero1Engine
    port = 31500 + (os.getpid() + 37 * W) % 2000
    mp.spawn(_worker4, args=(W, port, str(tmp_path)), nprocs=W, join=True)
    rs = [torch.load(tmp_path / f"r{r}.pt") for r in range(W)]
    assert rs[0]["n_buckets"] > 2                             # several buckets: the projector's is not the only one
    assert len(rs[0]["proj_grads_step0"]) > 0 and rs[3]["proj_grads_step0"] == []     # rank 3 produced no projector gradient
    # bucket geometry at this world size: buckets tile the flat buffer, every bucket is W equal pieces of whole 16-byte groups
    bounds = rs[0]["bucket_bounds"]
    assert bounds[0][0] == 0 and bounds[-1][1] == rs[0]["numel"] and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    for st, en, piece in bounds:
        assert piece * W == en - st and piece % 8 == 0 and st % (8 * W) == 0
    assert all(r["bucket_bounds"] == bounds for r in rs)
    for r in range(1, W):
        for n, a in rs[0]["params"].items():
            assert torch.equal(a, rs[r]["params"][n]), n     # replicas stay identical
    # one process: per-record backward (L2Wrap's penalty gradient does not scale with the loss weight, src/model.py:38-46, so the
    # mean is taken over gradients, as data parallelism does), global-norm clip, torch's AdamW
    m = _tiny_visual()
    train = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(train, lr=_ENG4["lr"], betas=_ENG4["betas"], eps=_ENG4["eps"], weight_decay=0.0)
    for _ in range(2):
        acc = [torch.zeros_like(p) for p in train]
        for r in range(W):
            gs = torch.autograd.grad(m.training_step(_rank_batch(r)), train, allow_unused=True)
            for a, g in zip(acc, gs):
                if g is not None:
                    a.add_(g, alpha=1.0 / W)
        gn = float(torch.sqrt(sum((a.double() ** 2).sum() for a in acc)))
        c = min(1.0, _ENG4["grad_clip"] / (gn + 1e-6))
        for p, a in zip(train, acc):
            p.grad = a * c
        opt.step()
    moved = 0
    for n, p in m.named_parameters():
        if p.requires_grad:
            d_dp, d_one = rs[0]["params"][n] - rs[0]["init"][n], p.detach() - rs[0]["init"][n]
            scale = float(d_one.abs().max())
            assert float((d_dp - d_one).abs().max()) <= 1e-3 * scale + 1e-7, (n, scale, float((d_dp - d_one).abs().max()))
            moved += int(n.startswith("proj.") and scale > 0)
    assert moved > 0
