"""The LoRA products of RWKV_Tmix_x070 (src/model.py:176,181-184) as the library runs them: x (M x 2048) @ w1 (2048 x r), h (M x r) @ w2 (r x 2048)
and their input gradients dy @ w^T, against the time of reading / writing the wide operand once.  python benchmarks/lora_gemm_micro.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
    enable_tuned_gemms()
    M, C = 16 * 2624, 2048
    dev = "cuda"
    x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    for r in (64, 96, 256):
        w1 = torch.randn(C, r, device=dev, dtype=torch.bfloat16) * 0.02
        w2 = torch.randn(r, C, device=dev, dtype=torch.bfloat16) * 0.02
        h = torch.randn(M, r, device=dev, dtype=torch.bfloat16)
        rec = {"M": M, "C": C, "r": r,
               "down_x@w1_us": timeit(lambda: x @ w1), "up_h@w2_us": timeit(lambda: h @ w2),
               "dgrad_up_dy@w2T_us": timeit(lambda: x @ w2.t()), "dgrad_down_dh@w1T_us": timeit(lambda: h @ w1.t()),
               "wide_once_us_at_5TBps": M * C * 2 / 5e6}
        print(json.dumps({k: round(v, 1) if isinstance(v, float) else v for k, v in rec.items()}))


if __name__ == "__main__":
    main()
