"""Per-phase shader-clock breakdown of the chunked WKV7 kernels (workgroup 0), via vrwkv_wkv7_profile_bf16."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs
from visualrwkv_amd import hip_lib

FWD = ["c_top", "c_main1", "c_waitA", "c_store_y_sa", "c_supdate_store_s", "c_waitB", "-", "-", "p_prep", "p_waitA", "p_scores", "p_waitB"]
BWD5 = ["c_top", "c_isplit", "c_waitX", "c_jsplit", "c_waitY", "c_seg3_tail", "c_waitZ", "-", "p_top", "p_prepA", "p_waitX", "p_prepB", "p_dM", "p_waitY", "p_scores", "p_waitZ"]

def run(B=8, T=2624, H=32, bwd_variant=-1, fwd_variant=-1):
    lib = hip_lib.load()
    lib.vrwkv_wkv7_set_backward_variant(bwd_variant)
    lib.vrwkv_wkv7_set_forward_variant(fwd_variant)
    dev = "cuda:0"
    w, q, k, v, z, a, dy = synth_inputs(B, T, H, dev)
    y = torch.empty_like(v); s = torch.empty(B, H, T // 16, 64, 64, device=dev); sa = torch.empty(B, T, H, 64, device=dev)
    g = [torch.empty_like(w) for _ in range(6)]
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for bw, names in ((0, FWD), (1, BWD5)):
        dbg = torch.zeros(16, dtype=torch.int64, device=dev)
        rc = lib.vrwkv_wkv7_profile_bf16(bw, B, T, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(), a.data_ptr(),
                                         dy.data_ptr(), y.data_ptr(), s.data_ptr(), sa.data_ptr(), *[x.data_ptr() for x in g], dbg.data_ptr(), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        nch = T // 16
        out["bwd" if bw else "fwd"] = {n: round(d[i] / nch) for i, n in enumerate(names)}
        out[("bwd" if bw else "fwd") + "_total_per_chunk"] = round(sum(d) / nch)
    return out

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bv = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    fv = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    print(json.dumps(run(B=B, bwd_variant=bv, fwd_variant=fv)))
