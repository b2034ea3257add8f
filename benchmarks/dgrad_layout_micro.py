"""Input-gradient GEMM of a Linear layer, dx = dy @ W, two ways on the step's shapes (M = 41 984 tokens):
  nn   what autograd issues: dy.mm(W)            -> hipBLASLt "N,N" kernel (W is (N_out, K_in) row-major: the contraction dim strided)
  tn   F.linear(dy, Wt), Wt = fused.transpose2d(W) (tiled HIP transpose) -> the "T,N" layout of the forward GEMMs (both operands contraction-contiguous)
`tn_ms` includes the transpose copy of the weight.  With the shipped TunableOp selections, as in bench.py."""
import json, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
from visualrwkv_amd import fused

def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

torch.cuda.set_device(0)
n = enable_tuned_gemms() if "--no-tuned" not in sys.argv else 0
M = 41984
for name, N, K in [("att r/k/v/o", 2048, 2048), ("ffn key", 8192, 2048), ("ffn value", 2048, 8192), ("head", 65536, 2048)]:
    dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16() * 0.02
    x = torch.randn(M, K, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    t_nn = bench(lambda: dy.mm(W))
    t_tn = bench(lambda: F.linear(dy, fused.transpose2d(W)))
    t_tr = bench(lambda: fused.transpose2d(W))
    t_tr_torch = bench(lambda: W.t().contiguous())
    t_fwd = bench(lambda: F.linear(x, W))
    t_wg = bench(lambda: dy.t().mm(x))
    err = float((dy.mm(W).float() - F.linear(dy, W.t().contiguous()).float()).abs().max())
    print(json.dumps({"layer": name, "N_out": N, "K_in": K, "tuned_shapes": n, "nn_ms": round(t_nn, 4), "nn_TFLOPs": round(fl / t_nn / 1e9),
                      "tn_ms": round(t_tn, 4), "tn_TFLOPs": round(fl / t_tn / 1e9), "transpose_ms": round(t_tr, 4), "torch_transpose_ms": round(t_tr_torch, 4),
                      "fwd_ms": round(t_fwd, 4), "fwd_TFLOPs": round(fl / t_fwd / 1e9), "wgrad_ms": round(t_wg, 4), "wgrad_TFLOPs": round(fl / t_wg / 1e9),
                      "max_abs_diff": err}), flush=True)
