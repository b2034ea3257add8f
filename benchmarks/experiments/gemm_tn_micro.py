"""The channel-mix GEMMs with their epilogue inside (csrc/gemm_tn.h) against the library's T,N kernel + the streaming pass it replaces
(M = 41 984 tokens, 2048 -> 8192): key projection + relu^2 (forward) and value's input gradient * 2 sqrt(h2) (backward)."""
import json, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
import ctypes
from visualrwkv_amd import fused

def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best

torch.cuda.set_device(0)
n = enable_tuned_gemms() if "--no-tuned" not in sys.argv else 0
lib = ctypes.CDLL(os.path.join(ROOT, "benchmarks", "_alt", "libgemm_tn.so"))          # built as benchmarks/experiments/README.md says
lib.vrwkv_gemm_tn_bf16.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2
st = torch.cuda.current_stream().cuda_stream
M, N, K = 41984, 8192, 2048
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def own(epi, a, b, aux=None):
    rc = lib.vrwkv_gemm_tn_bf16(M, N, K, a.data_ptr(), b.data_ptr(), C.data_ptr(), epi, aux.data_ptr() if aux is not None else 0, st)
    assert rc == 0, rc
fl = 2.0 * M * N * K
out = {"M": M, "N": N, "K": K, "tuned_shapes": n}
# plain
own(0, x, W); ref = F.linear(x, W)
out["plain_rel_err"] = float((C.float() - ref.float()).norm() / ref.float().norm())
out["library_ms"] = round(bench(lambda: F.linear(x, W)), 4); out["own_plain_ms"] = round(bench(lambda: own(0, x, W)), 4)
# forward: key + relu^2
kk = F.linear(x, W)
h2_ref = torch.relu(kk.float()) ** 2
own(1, x, W)
out["relusq_rel_err"] = float((C.float() - h2_ref).norm() / h2_ref.norm())
out["library_plus_relusq_ms"] = round(bench(lambda: fused.relusq(F.linear(x, W)) if hasattr(fused, "relusq") else torch.relu(F.linear(x, W)) ** 2), 4)
out["own_relusq_ms"] = round(bench(lambda: own(1, x, W)), 4)
# backward: value's input gradient (dout (M x 2048) @ Wv (2048 x 8192) = F.linear(dout, Wv^T)) * 2 sqrt(h2)
h2 = C.clone()
dout = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
WvT = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
dref = F.linear(dout, WvT).float() * (2 * torch.sqrt(h2.float()))
own(2, dout, WvT, h2)
out["drelusq_rel_err"] = float((C.float() - dref).norm() / dref.norm())
out["own_drelusq_ms"] = round(bench(lambda: own(2, dout, WvT, h2)), 4)
for k in ("library_ms", "own_plain_ms", "own_relusq_ms", "own_drelusq_ms"):
    out[k.replace("_ms", "_TFLOPs")] = round(fl / out[k] / 1e9)
print(json.dumps(out))
