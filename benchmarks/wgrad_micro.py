"""Skinny weight-gradient product (csrc/lora_wgrad.h) against the library GEMM on the LoRA shapes of the 1.5B step:
python benchmarks/wgrad_micro.py [M]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd import fused, gemm_tuning  # noqa: E402


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 41984
    gemm_tuning.enable_tuned_gemms()                 # the library side uses the kernels the step uses
    for K, N in ((2048, 96), (96, 2048), (2048, 64), (64, 2048), (2048, 256), (256, 2048)):
        x = torch.randn(M, K, device="cuda").bfloat16()
        dy = torch.randn(M, N, device="cuda").bfloat16()
        us_lib = t(lambda: x.t() @ dy)
        us_new = t(lambda: fused.wgrad_skinny(x, dy))
        floor = (M * max(K, N) * 2) / 5.6e12 * 1e6
        print(json.dumps({"M": M, "K": K, "N": N, "library_us": round(us_lib, 1), "kernel_us": round(us_new, 1),
                          "read_wide_once_us_at_5.6TBps": round(floor, 1)}))


if __name__ == "__main__":
    main()
