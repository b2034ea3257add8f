"""WKV6 forward/backward timing at the BASELINE config-4 shape (B x 2624 x 64 heads): ms and achieved fraction of the
8 TB/s HBM roofline for the algorithmic bytes (forward 12 B/elem = r,k,v bf16 + ew f32 in, y out; plus the
16 B/elem chunk-state checkpoint this implementation writes for the backward; backward: 6 bf16 in/out pairs ... see
below).  python benchmarks/wkv6_micro.py [B ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd import wkv6  # noqa: E402

FWD_B = 2 * 3 + 4 + 2 + 16          # r,k,v + ew + y + checkpoint (64*64*4 / (16*64))
BWD_B = 2 * 4 + 4 + 16 + 2 * 4      # r,k,v,gy + ew + checkpoint + gr,gk,gv,gw
PEAK = 8000.0


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [4, 8]
    T, H = 2624, 64
    C = H * 64
    for B in Bs:
        g = torch.Generator(device="cuda").manual_seed(0)
        uni = lambda *s, lo=-1.0, hi=1.0: (torch.rand(*s, device="cuda", generator=g) * (hi - lo) + lo).bfloat16()
        r, k, v, gy = uni(B, T, C), uni(B, T, C), uni(B, T, C), uni(B, T, C)
        w, u = uni(B, T, C, lo=-8.0, hi=1.0), uni(H, 64)
        ew = (-torch.exp(w.float())).contiguous()
        y = torch.empty_like(r)
        ck = wkv6.ckpt_tensor(B, T, H, "cuda")
        outs = [torch.empty_like(r) for _ in range(4)]
        gu = torch.empty(B, C, dtype=torch.bfloat16, device="cuda")

        def t(fn, iters=10):
            fn(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters)
            return best
        f_ms = t(lambda: wkv6.forward_hip(B, T, C, H, r, k, v, ew, u, y, ck))
        fi_ms = t(lambda: wkv6.forward_hip(B, T, C, H, r, k, v, ew, u, y, None))
        b_ms = t(lambda: wkv6.backward_hip(B, T, C, H, r, k, v, ew, u, gy, *outs, gu, ck))
        n = B * T * C
        print(json.dumps({"B": B, "T": T, "H": H, "fwd_ms": round(f_ms, 4), "fwd_frac": round(n * FWD_B / f_ms / 1e6 / PEAK, 4),
                          "fwd_nockpt_ms": round(fi_ms, 4), "fwd_nockpt_frac": round(n * 12 / fi_ms / 1e6 / PEAK, 4),
                          "bwd_ms": round(b_ms, 4), "bwd_frac": round(n * BWD_B / b_ms / 1e6 / PEAK, 4)}))


if __name__ == "__main__":
    main()
