"""Single-stream decode of the 1.5B RWKV-7 stack with carried state: eager steps vs the HIP-graph-captured step.
python benchmarks/decode_micro.py [n_tokens] [batch]"""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visualrwkv_amd.rwkv7 import RWKV  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    args = SimpleNamespace(n_embd=2048, n_layer=24, dim_att=2048, head_size_a=64, head_size_divisor=8, vocab_size=65536,
                           dropout=0, grad_cp=0, ctx_len=4096, load_model="", fused=True)
    torch.manual_seed(0)
    with torch.device("cuda"):
        m = RWKV(args)
    m = m.bfloat16().eval()
    prompt = torch.randn(B, 2624, 2048, device="cuda", dtype=torch.bfloat16)
    res = {"batch": B}
    with torch.no_grad():
        m.forward_stateful(prompt, None, last_only=True)               # cold: library initialisation, kernel loading
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            logits, st = m.forward_stateful(prompt, None, last_only=True)
        torch.cuda.synchronize(); res["prefill_2624_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        for mode in ("eager", "graph"):
            _, st = m.forward_stateful(prompt[:, :64], None, last_only=True)
            dec = m.make_decoder(st) if mode == "graph" else None
            x = torch.randn(B, 1, 2048, device="cuda", dtype=torch.bfloat16)
            for _ in range(3):
                (dec(x) if dec else m.forward_stateful(x, st, last_only=True))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                lg = dec(x) if dec else m.forward_stateful(x, st, last_only=True)[0]
                nxt = lg.argmax(-1).tolist()                     # the sampling step's host sync, as in generate_stateful
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res[f"{mode}_ms_per_token"] = round(dt * 1e3, 3)
            res[f"{mode}_tokens_per_s"] = round(B / dt, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
