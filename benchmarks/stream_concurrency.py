"""Do two HIP streams of one process run kernels CONCURRENTLY on this box?  (Background: profiles/r5_rccl_*: with one rank, the engine's
communication-stream work added to the step time in full.)  Stream A = a fixed train of large bf16 GEMMs (the training step's dominant
kernel class).  Stream B, started at the same moment: nothing | a train of SMALL-grid kernels (few workgroups, long: how an RCCL kernel
looks to the chip) | a train of full-GPU device-to-device copies of 200 MB (what a one-rank RCCL "collective" degenerates to).  Reported:
wall time of stream A alone, of B alone, and of both started together; overlap = (A + B - both) / min(A, B)."""
import json
import os
import sys

import torch


def timed(fn_a, fn_b, sa, sb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go = torch.cuda.Event(); go.record()
    if fn_a:
        sa.wait_event(go)
        with torch.cuda.stream(sa):
            fn_a()
        torch.cuda.current_stream().wait_stream(sa)
    if fn_b:
        sb.wait_event(go)
        with torch.cuda.stream(sb):
            fn_b()
        torch.cuda.current_stream().wait_stream(sb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def main():
    dev = "cuda:0"
    prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0              # priority of stream B (torch: -1 = high, 0 = default)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
    x = torch.randn(16 * 2624, 2048, device=dev, dtype=torch.bfloat16)
    w = torch.randn(8192, 2048, device=dev, dtype=torch.bfloat16)
    y = torch.empty(16 * 2624, 8192, device=dev, dtype=torch.bfloat16)
    small = torch.randn(64 * 256 * 8, device=dev)                     # 64 workgroups of 256 threads x 8 elements
    big_src = torch.empty(100 * 1000 * 1000, device=dev, dtype=torch.bfloat16)
    big_dst = torch.empty_like(big_src)

    def gemms():
        for _ in range(40):
            torch.mm(x, w.t(), out=y)

    def small_train():
        for _ in range(3000):
            small.mul_(1.0001)

    def copies():
        for _ in range(40):
            big_dst.copy_(big_src)

    import ctypes
    proxy = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_alt", "libcomm_proxy.so")).comm_proxy_copy
    proxy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def rccl_like(wgs, reps):
        def f():
            for _ in range(8):          # eight "collectives" of one long-running kernel each
                assert proxy(big_src.data_ptr(), big_dst.data_ptr(), big_src.numel() * 2, wgs, reps, torch.cuda.current_stream().cuda_stream) == 0
        return f

    for f in (gemms, small_train, copies, rccl_like(16, 1)):
        f()
    torch.cuda.synchronize()
    out = {"stream_b_priority": prio, "priority_range": list(torch.cuda.Stream.priority_range()) if hasattr(torch.cuda.Stream, "priority_range") else None, "env": {k: os.environ.get(k) for k in ("AMD_SERIALIZE_KERNEL", "AMD_SERIALIZE_COPY", "HIP_LAUNCH_BLOCKING", "GPU_MAX_HW_QUEUES", "HSA_ENABLE_SDMA")}}
    a = min(timed(gemms, None, sa, sb) for _ in range(3))
    out["gemm_train_alone_ms"] = round(a, 2)
    for name, fb in (("small_grid_train", small_train), ("copy_200MB_train", copies), ("rccl_like_16wg_x8", rccl_like(16, 2)), ("rccl_like_32wg_x8", rccl_like(32, 4)),
                     ("rccl_like_64wg_x8", rccl_like(64, 8))):
        b = min(timed(None, fb, sa, sb) for _ in range(3))
        both = min(timed(gemms, fb, sa, sb) for _ in range(3))
        out[name] = {"alone_ms": round(b, 2), "both_ms": round(both, 2), "overlap_of_the_shorter": round((a + b - both) / min(a, b), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
