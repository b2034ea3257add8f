#!/usr/bin/env python
"""Training-step benchmark of the VisualRWKV-7 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N`, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or -- when WORLD_SIZE is not set --
bench.py re-executes itself under torch.distributed.run with N ranks on 127.0.0.1 (the reference: `--devices 8 --strategy
deepspeed_stage_1`, VisualRWKV-v7/v7.00/train.py:75-76,98).  `--backend gloo --model tiny` runs the same entry point on host
cores (fp32, eager modules, the op's CPU key): the multi-rank plumbing test of tests/test_bench_cpu.py, not a benchmark.

One step = ViT encode (frozen SigLIP + DINOv2) -> pool -> projector -> scatter -> RWKV-7 forward ->
shifted CE (+L2Wrap) -> backward -> bucketed reduce-scatter -> clip 1.0 -> fused AdamW -> all-gather, on a
synthetic LLaVA-style batch (576 image tokens + 2048 text tokens = 2624 tokens per sample, bf16), random-init
weights of the VisualRWKV-7 1.5B architecture (L24 C2048 H32, vocab 65536).  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = {   # sizes of the RWKV-x070 checkpoints the reference README points to
    "1b5": dict(n_layer=24, n_embd=2048),
    "0b4": dict(n_layer=24, n_embd=1024),
    "0b1": dict(n_layer=12, n_embd=768),
    "tiny": dict(n_layer=2, n_embd=128),       # plumbing only (CPU / gloo test of the N > 1 launch path)
}
TINY_TOWERS = {"dino": dict(depth=2, dim=64, heads=1), "siglip": dict(depth=2, dim=64, heads=1, mlp_hidden=96),
               "sam": dict(img_size=128, dim=64, depth=2, heads=1, out_chans=16, window=3, global_attn_indexes=(1,))}
FWD_B, BWD_B = 34, 46          # algorithmic bytes per bf16 element at chunk length 16
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md


def _concurrency_keys():
    """Which groups of library GEMMs ran on more than one HIP stream (True) or one after the other (False): visualrwkv_amd/gemm_tuning.py."""
    from visualrwkv_amd import gemm_tuning
    return gemm_tuning.concurrency_report()


def build_args(name, ctx_len, n_img_tokens, towers, grad_cp, fused, vit_minibatch=4, image_size=None):
    m = MODELS[name]
    tiny = name == "tiny"
    return SimpleNamespace(n_layer=m["n_layer"], n_embd=m["n_embd"], dim_att=m["n_embd"], head_size_a=64,
                           head_size_divisor=8, vocab_size=65536, dropout=0, grad_cp=grad_cp, ctx_len=ctx_len,
                           load_model="", num_token_per_image=n_img_tokens, proj_type="mlp", vision_towers=towers,
                           vision_image_size=image_size or (56 if tiny else 448), vision_tower_kwargs=TINY_TOWERS if tiny else None,
                           weight_decay=0.0, fused=fused, check_image_tokens=False, vit_minibatch=vit_minibatch)


def synthetic_batch(B, ctx_len, n_img, towers, device, seed, side=448, sam_side=1024, dtype=torch.bfloat16):
    """ids uniform in [0,65535) with a run of n_img image placeholders after a 4-token prefix,
    labels -100 on the first 60 %, pixels N(0,1) bf16."""
    g = torch.Generator(device=device).manual_seed(seed)
    ids = torch.randint(0, 65535, (B, ctx_len), device=device, generator=g)
    ids[:, 4:4 + n_img] = 65535
    labels = ids.clone()
    labels[:, : int(ctx_len * 0.6)] = -100
    labels[ids == 65535] = -100
    images = {}
    for t in towers:
        px = sam_side if t == "sam" else side
        images[t] = torch.randn(B, 3, px, px, device=device, generator=g).to(dtype)
    return {"input_ids": ids, "labels": labels, "images": images, "sample_id": [str(i) for i in range(B)]}


class ByteTokenizer:
    """Stand-in for the RWKV world tokenizer (not shipped with the reference checkout): any object with
    encode(str) -> list[int] serves data.MyDataset; UTF-8 bytes shifted past the special ids."""
    def encode(self, text):
        return [b + 300 for b in text.encode("utf-8")]


def loader_batches(a, towers, dev, rank, world, n_files=64, n_records=512):
    """--data loader: the reference's input pipeline end to end (train.py:219-222, src/dataset.py:167-246) on synthetic
    files -- n_files JPEGs (640 x 480, smooth random content so that the entropy decode is photo-like) and a LLaVA-style
    JSON on local disk -> data.MyDataset (rank-strided sampling, templating, masking; workers only decode) ->
    data.make_loader -> data.DevicePrefetcher (H2D copies + the tower transforms on a side stream, one batch ahead).
    Yields the `samples` dicts VisualRWKV.forward takes, forever."""
    import json as _json
    import tempfile
    import numpy as np
    from PIL import Image
    from visualrwkv_amd import data as vdata
    root = tempfile.mkdtemp(prefix=f"vrwkv_bench_data_r{rank}_")
    rng = np.random.default_rng(1234 + rank)
    for i in range(n_files):
        low = rng.integers(0, 255, size=(15, 20, 3), dtype=np.uint8)
        Image.fromarray(low).resize((640, 480), Image.BICUBIC).save(os.path.join(root, f"img{i:03d}.jpg"), quality=90)
    words = ["the", "image", "shows", "a", "view", "of", "some", "objects", "near", "window", "and", "light", "table"]
    n_text = a.ctx_len - a.img_tokens
    recs = []
    for i in range(n_records):
        q = "<image>\nDescribe the picture in detail."
        ans = " ".join(words[(i + j) % len(words)] for j in range(n_text))[: n_text + 64]      # truncated to ctx_len by preprocess
        recs.append({"id": f"s{i}", "image": f"img{i % n_files:03d}.jpg",
                     "conversations": [{"from": "human", "value": q}, {"from": "gpt", "value": ans}]})
    with open(os.path.join(root, "data.json"), "w") as f:
        _json.dump(recs, f)
    dargs = SimpleNamespace(data_file=os.path.join(root, "data.json"), image_folder=root, tokenizer=ByteTokenizer(), ctx_len=a.ctx_len,
                            num_token_per_image=a.img_tokens, epoch_steps=1 << 20, real_bsz=a.micro_bsz * world, micro_bsz=a.micro_bsz)
    ds = vdata.MyDataset(dargs, decode_only=True)
    ds.global_rank, ds.world_size, ds.real_epoch = rank, world, 0
    loader = vdata.make_loader(ds, a.micro_bsz, num_workers=a.loader_workers)
    for batch in vdata.DevicePrefetcher(loader, dev, towers=towers, dtype=torch.bfloat16):
        imgs = batch.get("images", {})
        batch["images"] = {k: v for k, v in imgs.items() if k in towers}       # tensors only (bench batches have one image each)
        yield batch


def stream_copy_gbps(dev, nbytes=2 << 30, iters=10, fill="random"):
    """HBM GB/s (read + write) of the library's streaming-copy kernel on `nbytes`, HIP events on the launch stream.
    fill: "random" bytes (what a kernel working on real data sees) or "zeros" -- the chip clocks to its power budget and
    switching power depends on the data (benchmarks/dvfs_probe.py), so the same copy is faster on zeros; both are reported."""
    from visualrwkv_amd import hip_lib
    lib = hip_lib.load()
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.random_(0, 255) if fill == "random" else src.zero_()
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream(dev)
    run = lambda: hip_lib.check(lib.vrwkv_stream_copy(src.data_ptr(), dst.data_ptr(), nbytes, st.cuda_stream), "vrwkv_stream_copy")
    run()
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            run()
        e1.record(st)
        e1.synchronize()
        best = max(best, 2 * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    assert torch.equal(dst[:4096], src[:4096]) and torch.equal(dst[-4096:], src[-4096:])
    return best


def cpu_baseline(n_embd, T):
    """CPU leg: one RWKV-7 block (time-mix incl. the WKV7 C oracle + channel-mix) forward+backward in fp32 on
    the host cores, B=1 at the bench sequence length; reported as tokens/s for a 24-layer stack of such
    blocks (head, loss, ViT and optimizer excluded -- they only make the CPU slower)."""
    from oracle import rwkv7_cpu, wkv7_c
    from visualrwkv_amd.rwkv7 import Block
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1      # all physical cores of the host
    except ImportError:
        cores = os.cpu_count() or 1
    logical = os.cpu_count() or cores
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    wkv7_c.load()
    a = SimpleNamespace(n_layer=24, n_embd=n_embd, dim_att=n_embd, head_size_a=64, head_size_divisor=8, dropout=0, grad_cp=0)
    blk = Block(a, 1)
    with torch.no_grad():
        for p in blk.parameters():
            if float(p.abs().sum()) == 0.0:
                p.normal_(0, 0.02)
    st = {"b." + k: v.detach().requires_grad_(True) for k, v in blk.state_dict().items()}
    x = (torch.randn(1, T, n_embd) * 0.5).requires_grad_(True)
    vf = torch.randn(1, T, n_embd) * 0.5
    times = []
    for it in range(6):                         # 1 warm-up + 5 timed runs, median
        t0 = time.perf_counter()
        y, _ = rwkv7_cpu.block(st, "b.", x, vf, 1, n_embd // 64)
        y.sum().backward()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    # the WKV7 operator alone (the C restatement of the reference kernels, OpenMP over heads), same byte formula as the
    # GPU roofline: tokens/s per layer and effective GB/s
    from oracle.wkv7_oracle import make_inputs
    H = n_embd // 64
    ow, oq, ok_, ov, oz, oa, ody = make_inputs(1, T, H, seed=42)
    kt = {"fwd": [], "bwd": []}
    for it in range(6):                         # 1 warm-up + 5 timed runs, median
        t0 = time.perf_counter()
        _, os_, osa = wkv7_c.forward(ow, oq, ok_, ov, oz, oa)
        t1 = time.perf_counter()
        wkv7_c.backward(ow, oq, ok_, ov, oz, oa, ody, os_, osa)
        kt["fwd"].append(t1 - t0); kt["bwd"].append(time.perf_counter() - t1)
    kf, kb = sorted(kt["fwd"][1:])[2], sorted(kt["bwd"][1:])[2]
    elems = T * H * 64
    kernel = {"shape": [1, T, H, 64], "fwd_ms": kf * 1e3, "bwd_ms": kb * 1e3, "fwd_GBps": elems * FWD_B / kf / 1e9,
              "bwd_GBps": elems * BWD_B / kb / 1e9, "tokens_per_s_per_layer": T / (kf + kb)}
    # the reference's own pure-PyTorch statement of the recurrence (VisualRWKV-v6/v6.xx/RWKV-v7_simple.py:20-32, restated in
    # oracle.wkv7_oracle.wkv7_naive): per-token matmuls over (B,H,64,64), fp32, forward + autograd backward, same shape
    # (bounded sample: the first 128 tokens -- the per-token cost does not depend on T; 512 tokens took 9.4 s a run on the 128-core host)
    from oracle.wkv7_oracle import wkv7_naive
    Tp = min(T, 128)
    pt = []
    for it in range(6):
        leaves = [x[:, :Tp].float().requires_grad_(True) for x in (ow, oq, ok_, ov, oz, oa)]
        t0 = time.perf_counter()
        yy, _ = wkv7_naive(*leaves)
        yy.backward(ody[:, :Tp].float())
        pt.append(time.perf_counter() - t0)
    tp = sorted(pt[1:])[2]
    kernel["pytorch_loop"] = {"what": "RWKV-v7_simple.py-style per-token loop, fp32, fwd + autograd bwd, median of 5", "tokens": Tp,
                              "fwd_bwd_ms": tp * 1e3, "tokens_per_s_per_layer": Tp / tp, "GBps": Tp * H * 64 * (FWD_B + BWD_B) / tp / 1e9}
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"value": T / (24 * t), "unit": "tokens/s", "cores": cores, "logical_cpus": logical, "kind": "port", "cpu_model": model, "wkv7_kernel": kernel,
            "sample": f"1 of 24 RWKV-7 1.5B blocks (Tmix with the C WKV7 oracle + CMix), fwd+bwd, fp32, B=1 T={T}; median of 5 "
                      f"runs after 1 warm-up, {t:.2f} s per block, scaled x24; head/loss/ViT/optimizer not included"}


class _StdoutToStderr:
    """Route file descriptor 1 to descriptor 2 for the duration: the collective library prints a version banner with C stdio on stdout (flushed when
    the process exits, i.e. AFTER the JSON line); stdout of this program is the one JSON line of rank 0 and nothing else."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)          # whatever C stdio buffered for "stdout" goes where fd 1 points now: stderr
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _wkv7_kernel_name(backward: int, B: int, T: int, H: int) -> str:
    """The kernel a WKV7 launch of this shape resolves to (vrwkv_wkv7_resolve_variant: a pure function of the shape and the A/B override --
    the launches themselves come from two threads, the Python thread and autograd's): what the roofline entry is about.
    backward: 0 = training forward, 1 = backward, 2 = the by-product-free forward of the selective-recompute mode."""
    from visualrwkv_amd import hip_lib
    v = hip_lib.load().vrwkv_wkv7_resolve_variant(backward, B, T, H)
    names = ({5: "wkv7v5::bwd_kernel_v5", 6: "wkv7v6::bwd_kernel_v6", 7: "wkv7v7::bwd_kernel_v7 (experiment)", 8: "wkv7v8::bwd_kernel_v8", 9: "wkv7v8::bwd_kernel_v8<AHEAD>",
              10: "wkv7v8::bwd_kernel_v8<AHEAD,JTAIL> (experiment)", 11: "wkv7v8::bwd_kernel_v8<JTAIL> (experiment)"} if backward == 1 else
             {7: "wkv7f4::fwd_kernel_v4", 6: "wkv7c::fwd_kernel_v3<two workgroups per head>"})
    return names.get(v, f"wkv7c::fwd_kernel_v3<variant {v}>" if backward != 1 else f"backward variant {v}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="1b5", choices=list(MODELS))
    ap.add_argument("--micro-bsz", type=int, default=16)      # 210 GB of the 288 GB HBM3E, no recompute
    ap.add_argument("--ctx-len", type=int, default=2624)          # 576 image + 2048 text tokens
    ap.add_argument("--img-tokens", type=int, default=576)
    ap.add_argument("--towers", default="dino,siglip")
    ap.add_argument("--image-size", type=int, default=0, help="side of the SigLIP / DINOv2 input (default 448; BASELINE config 1: 224 -> 256 patch tokens)")
    ap.add_argument("--grad-cp", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--vit-minibatch", type=int, default=16, help="images per ViT forward (the reference's loop uses 4 to save memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-cp-companion", action="store_true")
    ap.add_argument("--profile-ops", type=str, default="",
                    help="debugging aid: run one extra step under torch.profiler and write the GPU-time table of aten ops "
                         "with input shapes to this file")
    ap.add_argument("--fast-init", action="store_true",
                    help="profiling runs only: normal instead of orthogonal initialisers (rocprofv3 counter collection "
                         "segfaults inside the ~30 k tiny rocsolver kernels of the QR-based initialiser)")
    ap.add_argument("--gemm-tuning", type=int, default=1)         # library-GEMM kernel choice from the shipped TunableOp file
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one MI355X per rank (the benchmark); gloo = host cores, tiny model only "
                         "(plumbing test of the multi-rank launch path)")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "loader"],
                    help="loader: synthetic JPEGs on disk -> data.MyDataset workers -> DevicePrefetcher (train.py:219-222)")
    ap.add_argument("--loader-workers", type=int, default=8)
    ap.add_argument("--async-gather", type=int, default=1, help="overlap the parameter all-gather with the next ViT encode")
    a = ap.parse_args()
    cpu_mode = a.backend == "gloo"
    if cpu_mode and a.model not in ("tiny", "0b1"):
        ap.error("--backend gloo is the host-core mode: --model tiny (plumbing) or --model 0b1 (BASELINE config 1: fp32, the op's CPU key)")
    if a.model == "tiny":          # shapes of the tiny plumbing model unless given explicitly
        given = {x.split("=")[0] for x in sys.argv[1:] if x.startswith("--")}
        if "--ctx-len" not in given: a.ctx_len = 64
        if "--img-tokens" not in given: a.img_tokens = 16
        if "--micro-bsz" not in given: a.micro_bsz = 2
        if "--towers" not in given: a.towers = "dino,siglip,sam"

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nobody started the ranks: start them (one process per GPU, rendezvous on 127.0.0.1, a free port)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes on this driver)
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if cpu_mode:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1) // 2))
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (or --backend gloo --model tiny for the CPU plumbing run)"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    ranks_seen = 1
    if world > 1 or os.environ.get("VRWKV_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        with _StdoutToStderr():
            if cpu_mode:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            ones = torch.ones(1, device=dev)
            dist.all_reduce(ones)                       # the number of ranks the collective library actually connected
            ranks_seen = int(ones.item())
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from visualrwkv_amd import build, wkv7
    build.build()
    n_tuned = 0
    if a.gemm_tuning and not cpu_mode:
        from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
        n_tuned = enable_tuned_gemms()
    from visualrwkv_amd.dp import Zero1Engine
    from visualrwkv_amd.visual import VisualRWKV
    towers = tuple(t for t in a.towers.split(",") if t)
    args = build_args(a.model, a.ctx_len, a.img_tokens, towers, a.grad_cp, bool(a.fused) and not cpu_mode, a.vit_minibatch, a.image_size or None)
    dtype = torch.float32 if cpu_mode else torch.bfloat16     # host cores: BASELINE config 1's fp32 mode (the op's CPU key)
    torch.manual_seed(42)
    if a.fast_init:
        torch.nn.init.orthogonal_ = lambda t, gain=1.0: t.normal_(0, 0.02 * gain)
    with torch.device(dev):
        model = VisualRWKV(args)
    with torch.no_grad():      # random-init: make the zero-initialised projections non-degenerate
        for n, p in model.rwkv.named_parameters():
            if p.dim() >= 2 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.01)
    model = model.to(dtype)
    model.freeze_emb()         # fine-tune recipe: ViT and embedding frozen (train.py:196, model.py:349)
    engine = Zero1Engine(model, lr=2e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, grad_clip=1.0, bucket_mb=200.0,
                         force_collectives=os.environ.get("VRWKV_FORCE_COLLECTIVES") == "1", async_gather=bool(a.async_gather))
    tiny = a.model == "tiny"
    batch = synthetic_batch(a.micro_bsz, a.ctx_len, a.img_tokens, towers, dev, seed=1000 + rank, side=a.image_size or (56 if tiny else 448),
                            sam_side=128 if tiny else 1024, dtype=dtype)

    data_kind = "synthetic"
    feed = None
    if a.data == "loader":
        assert not cpu_mode, "--data loader needs the GPU transform path"
        feed = loader_batches(a, towers, dev, rank, world)
        data_kind = (f"loader: synthetic 640x480 JPEGs + LLaVA-style JSON on local disk -> MyDataset ({a.loader_workers} decode workers) "
                     "-> DevicePrefetcher (device transforms, one batch ahead)")

    def step():
        nonlocal batch
        if feed is not None:
            batch = next(feed)
        engine.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        engine.step(2e-5)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        if not cpu_mode:
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    if a.profile_ops and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            fence()
        with open(a.profile_ops, "w") as f:
            f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=400, max_name_column_width=50,
                                                                       max_shapes_column_width=90))
    wkv7.EVENT_LOG = [] if rank == 0 else None
    if engine.collective and engine.overlap:
        engine.comm_timing = []                     # exposed-communication estimate from stream events (dp.Zero1Engine.comm_report)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log, wkv7.EVENT_LOG = wkv7.EVENT_LOG, None
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    tokens = world * a.micro_bsz * a.ctx_len * a.steps

    # The reference's shipped scripts pass `--grad_cp 1` (scripts/train/*.sh): every Block re-computed in the backward (src/model.py:318-319);
    # BASELINE.json's config does not name grad_cp and the headline keeps all activations in the 288 GB of HBM (grad_cp=0).  Same model, same
    # batch, 10 timed steps each in the two memory-saving modes so that their cost stands beside the headline (not part of `value`):
    #   grad_cp 1 = the reference's recipe (block inputs only are kept), grad_cp 2 = selective recompute (WKV7 checkpoints and relu^2 re-formed
    #   in the backward, every GEMM output kept; not in the reference).
    peak_headline = torch.cuda.max_memory_allocated() / 2**30 if dev.type == "cuda" else 0.0
    comm = engine.comm_report(a.steps) if engine.comm_timing else None
    engine.comm_timing = None
    cp_ref = cp_sel = None
    if a.grad_cp == 0 and not a.no_grad_cp_companion and not cpu_mode:
        def companion(mode, n):
            args.grad_cp = mode
            step(); fence()
            torch.cuda.reset_peak_memory_stats()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            fence()
            tcp = torch.tensor([time.perf_counter() - t1], device=dev)
            if world > 1:
                dist.all_reduce(tcp, op=dist.ReduceOp.MAX)
            args.grad_cp = 0
            return {"grad_cp": mode, "tokens_per_s": world * a.micro_bsz * a.ctx_len * n / float(tcp), "ms_per_step": float(tcp) / n * 1e3, "steps": n,
                    "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        cp_sel = companion(2, 10)
        cp_sel["what"] = "selective recompute (not in the reference): WKV7 checkpoints (s, sa) and relu(h)^2 re-formed in the backward, every GEMM output kept"
        cp_ref = companion(1, 10)
        cp_ref["what"] = "the reference's --grad_cp 1 (every shipped script): every block re-computed in the backward (deepspeed.checkpointing per block)"

    if rank == 0:
        out = {
            "metric": f"train tokens/sec/node VisualRWKV-7 {a.model.upper()} {'fp32 (host cores)' if cpu_mode else 'bf16'}", "value": tokens / dt, "unit": "tokens/s",
            "n_gpus": world, "ranks_seen_by_collective": ranks_seen, "backend": ("gloo (host cores)" if cpu_mode else "rccl") if dist.is_initialized() else "none (1 rank)",
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp32" if cpu_mode else "bf16", "data": data_kind,
            "config": {"workload": f"VisualRWKV-7 {a.model} + {'+'.join(towers)} ViT, {a.img_tokens} img + {a.ctx_len - a.img_tokens} text tokens, "
                                   f"full train step (fwd+bwd+ZeRO-1 AdamW)", "model": f"VisualRWKV-7 {a.model}",
                       "global_batch": world * a.micro_bsz, "seq_len": a.ctx_len, "parallelism": f"dp{world}",
                       "grad_cp": a.grad_cp, "fused_elementwise": bool(args.fused), "loss": float(loss.detach()),
                       "micro_bsz": a.micro_bsz, "peak_mem_GB": None if cpu_mode else round(peak_headline, 1),
                       "gemm_kernels": f"TunableOp file, {n_tuned} shapes" if n_tuned else "library default",
                       "library_gemms_on_two_streams": _concurrency_keys(),
                       "grad_cp_meaning": "0 keep all activations (headline) | 1 = the reference's --grad_cp 1: every block re-computed | 2 = selective recompute",
                       "grad_cp1_reference_recipe_same_run": cp_ref, "grad_cp2_selective_same_run": cp_sel},
            "per_rank": {"tokens_per_step": a.micro_bsz * a.ctx_len, "micro_bsz": a.micro_bsz, "ranks": world,
                         "data_path_collectives": "none (batch shards); gradients: bucketed reduce-scatter on a side stream during the backward, "
                                                  "one scalar all-reduce (clip norm), parameter all-gather on the side stream after AdamW"},
            "comm": comm,
        }
        # roofline of the dominant hot-path kernel (WKV7 backward), HIP events on the launch stream
        kinds = {}
        for kind, e0, e1, elems in log:
            kinds.setdefault(kind, []).append((e0.elapsed_time(e1), elems))
        if "bwd" in kinds:
            ms = sum(x for x, _ in kinds["bwd"]) / len(kinds["bwd"])
            elems = kinds["bwd"][0][1]
            ach = elems * BWD_B / ms / 1e6
            shp = (a.micro_bsz, a.ctx_len + (-a.ctx_len) % 16, args.n_embd // 64)
            out["roofline"] = {"bound": "hbm", "kernel": _wkv7_kernel_name(1, *shp), "achieved": ach, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                               "avg_ms": ms, "launches": len(kinds["bwd"]), "algorithmic_bytes": elems * BWD_B}
            if "fwd" in kinds:
                msf = sum(x for x, _ in kinds["fwd"]) / len(kinds["fwd"])
                out["roofline"]["fwd_kernel"] = {"kernel": _wkv7_kernel_name(0, *shp), "avg_ms": msf, "launches": len(kinds["fwd"]),
                                                 "achieved": elems * FWD_B / msf / 1e6, "frac": elems * FWD_B / msf / 1e6 / HBM_PEAK_GBPS}
            if "fwd_state" in kinds:                  # --grad-cp 2: the training forward is the by-product-free entry (2 + 12 B / element)
                msf = sum(x for x, _ in kinds["fwd_state"]) / len(kinds["fwd_state"])
                out["roofline"]["fwd_state_kernel"] = {"kernel": _wkv7_kernel_name(2, *shp), "avg_ms": msf, "launches": len(kinds["fwd_state"]),
                                                       "algorithmic_bytes_per_element": 14, "achieved": elems * 14 / msf / 1e6,
                                                       "frac": elems * 14 / msf / 1e6 / HBM_PEAK_GBPS}
            copy = stream_copy_gbps(dev)                  # what a plain copy reaches on this box, random bytes
            out["roofline"]["stream_copy_GBps"] = copy
            out["roofline"]["stream_copy_zero_data_GBps"] = stream_copy_gbps(dev, fill="zeros")     # same kernel, higher clock
            out["roofline"]["frac_of_stream_copy"] = ach / copy
            out["roofline"]["guide_copy_GBps"] = 6290.0   # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s spec
            out["roofline"]["frac_of_guide_copy"] = ach / 6290.0
            pmc = os.path.join(ROOT, "profiles", "wkv7_pmc.json")
            if os.path.exists(pmc):
                rec = json.load(open(pmc)).get(f"bwd_B{a.micro_bsz}_T{a.ctx_len}_H{args.n_embd // 64}")
                if rec:                                    # NOT measured in this run: the committed rocprofv3 PMC passes
                    out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_kernel"] = rec.get("kernel")       # the kernel the counters were collected on (must be the one named in "kernel")
                    out["roofline"]["traffic_source"] = "profiles/wkv7_pmc.json (separate rocprofv3 --pmc passes of benchmarks/wkv7_pmc.sh, same shape)"
        if world == 1 and not a.no_cpu_baseline and not cpu_mode:
            out["cpu_baseline"] = cpu_baseline(args.n_embd, a.ctx_len)
        print(json.dumps(out))
    if dist.is_initialized():
        sys.stdout.flush()
        os.dup2(2, 1)                                # anything the collective library still prints (its banner is flushed at exit) goes to stderr
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
