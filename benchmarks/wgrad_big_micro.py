"""Weight gradient dW = dy^T x of the step's Linear layers (M = 41 984 tokens): the library's "N,T" kernel (dy.t().mm(x), with the
shipped TunableOp selections as in bench.py) against csrc/wgrad_big.h (vrwkv_wgrad_big_bf16), on random operands and on the
small-magnitude operands of a real step (the matrix cores clock to their power budget: report both)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
from visualrwkv_amd import hip_lib

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
lib = hip_lib.load()
st = torch.cuda.current_stream().cuda_stream
M = 41984
shapes = [("att r/k/v/o", 2048, 2048), ("ffn key", 8192, 2048), ("ffn value", 2048, 8192)] + ([("head", 65536, 2048)] if "--head" in sys.argv else [])
for name, N, K in shapes:
    for scale in (1.0, 0.05):
        dy = (torch.randn(M, N, device="cuda") * scale).bfloat16()
        x = (torch.randn(M, K, device="cuda") * scale).bfloat16()
        out = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
        nws = lib.vrwkv_wgrad_big_ws_floats(M, N, K)
        assert nws >= 0
        ws = torch.empty(max(nws, 4), dtype=torch.float32, device="cuda")
        def own():
            rc = lib.vrwkv_wgrad_big_bf16(M, N, K, dy.data_ptr(), x.data_ptr(), out.data_ptr(), ws.data_ptr(), st)
            assert rc == 0, rc
        own(); torch.cuda.synchronize()
        ref = dy.t().mm(x)
        err = float((out.float() - ref.float()).norm() / ref.float().norm())
        fl = 2.0 * M * N * K
        t_lib, t_own = bench(lambda: dy.t().mm(x)), bench(own)
        print(json.dumps({"layer": name, "N_out": N, "K_in": K, "operand_scale": scale, "tuned_shapes": n, "split_k": int(nws // (N * K)) if nws else 1,
                          "library_ms": round(t_lib, 4), "library_TFLOPs": round(fl / t_lib / 1e9), "own_ms": round(t_own, 4),
                          "own_TFLOPs": round(fl / t_own / 1e9), "rel_err_vs_library": err}), flush=True)
