"""hipBLASLt throughput on the GEMM shapes of the 1.5B step (M = B*T tokens)."""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 2624
dev = "cuda"
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
shapes = [("att CxC", 2048, 2048), ("ffn key C->4C", 2048, 8192), ("ffn value 4C->C", 8192, 2048), ("head C->V", 2048, 65536),
          ("lora C->96", 2048, 96), ("lora 96->C", 96, 2048), ("lora C->256", 2048, 256), ("lora 256->C", 256, 2048)]
for name, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)      # nn.Linear weight (N,K)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    t_f = bench(lambda: torch.nn.functional.linear(x, w))        # y = x W^T
    t_d = bench(lambda: dy @ w)                                  # dx = dy W
    t_w = bench(lambda: dy.t() @ x)                              # dW = dy^T x
    print(f"{name:18s} K={K:5d} N={N:6d}  fwd {t_f:7.3f} ms {fl/t_f/1e9:7.0f} TF | dgrad {t_d:7.3f} ms {fl/t_d/1e9:7.0f} TF | wgrad {t_w:7.3f} ms {fl/t_w/1e9:7.0f} TF")
