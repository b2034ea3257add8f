"""Per-phase shader-clock breakdown of the chunked WKV7 kernels (workgroup 0), via vrwkv_wkv7_profile_bf16."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs
from visualrwkv_amd import hip_lib

FWD = ["c_top", "c_main1", "c_waitA", "c_store_y_sa", "c_supdate_store_s", "c_waitB", "-", "-", "p_prep", "p_waitA", "p_scores", "p_waitB"]
BWD5 = ["c_top", "c_isplit", "c_waitX", "c_jsplit", "c_waitY", "c_seg3_tail", "c_waitZ", "realtime_100MHz", "p_top", "p_prepA", "p_waitX", "p_prepB", "p_dM", "p_waitY", "p_scores", "p_waitZ"]

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
    BWD6 = ["I_scores_dM", "I_flagwait", "I_isplit", "I_barrier", "I_top", "J_jsplit", "J_dMwait", "J_products", "J_barrier", "J_top",
            "P_tail", "P_prep", "P_drain", "P_barrier", "P_top", "realtime_100MHz", "J_js_split", "J_js_outputs", "I_is_dSA_dR", "I_is_dV", "J_tail"]
    only = os.environ.get("VRWKV_PHASES_ONLY")        # e.g. "4": the v8 entry alone
    for bw, names in ((0, FWD), (1, BWD5), (2, BWD6), (3, BWD6), (4, BWD6)):
        if only is not None and str(bw) not in only.split(","):
            continue
        dbg = torch.zeros(32, dtype=torch.int64, device=dev)
        rc = lib.vrwkv_wkv7_profile_bf16(bw, B, T, H, w.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), z.data_ptr(), a.data_ptr(),
                                         dy.data_ptr(), y.data_ptr(), s.data_ptr(), sa.data_ptr(), *[x.data_ptr() for x in g], dbg.data_ptr(), st)
        if rc == -1:                 # VRWKV_EINVAL: this generation is not in the library that is loaded (bw = 3, wkv7_bwd_v7.h, lives in experiment builds only)
            continue
        assert rc == 0, rc
        torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        nch = T // 16
        key = ["fwd", "bwd", "bwd_v6", "bwd_v7", "bwd_v8"][bw]
        out[key] = {n: round(d[i] / nch) for i, n in enumerate(names)}
        out[key + "_total_per_chunk"] = round(sum(d[:15]) / nch)
        if bw >= 2 and d[15] > 0:
            cyc = sum(d[0:5]) + d[18] + d[19]            # I wave 0: the five role stamps + the two inner i-split stamps
            out[key + "_shader_clock_GHz"] = round(cyc / (d[15] * 10.0), 3)
            out[key + "_cycles_per_step"] = round(cyc / (nch + 3))
        if bw == 1 and d[7] > 0:      # shader clock while this kernel runs: consumer-wave cycles of workgroup 0 / its life on the 100 MHz counter
            out["bwd_shader_clock_GHz"] = round(sum(d[:7]) / (d[7] * 10.0), 3)
    return out

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bv = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    fv = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    print(json.dumps(run(B=B, bwd_variant=bv, fwd_variant=fv)))
