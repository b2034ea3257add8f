"""Bit-compare backward variants of one library build on random inputs at the bench shape: python benchmarks/variant_equal.py <ref variant> <variants ...>
(experiment variants need VRWKV_HIP_LIB=benchmarks/_alt/lib_<name>.so)."""
import sys, os, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs
from visualrwkv_amd import hip_lib

lib = hip_lib.load()
ref_v, others = int(sys.argv[1]), [int(x) for x in sys.argv[2:]]
out = []
for B, T, H in ((16, 2624, 32), (3, 208, 5), (1, 16, 1)):
    w, q, k, v, z, a, dy = synth_inputs(B, T, H, "cuda:0")
    y = torch.empty_like(v); s = torch.empty(B, H, T // 16, 64, 64, device="cuda:0"); sa = torch.empty(B, T, H, 64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.vrwkv_wkv7_forward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, a, y, s, sa)], st) == 0
    res = {}
    for var in [ref_v] + others:
        assert lib.vrwkv_wkv7_set_backward_variant(var) == 0, var
        g = [torch.full_like(w, float("nan")) for _ in range(6)]
        rc = lib.vrwkv_wkv7_backward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, a, dy, s, sa, *g)], st)
        torch.cuda.synchronize()
        assert rc == 0, (var, rc)
        res[var] = g
    lib.vrwkv_wkv7_set_backward_variant(-1)
    for var in others:
        out.append({"shape": [B, T, H], "variant": var, "ref": ref_v, "bit_equal": [bool(torch.equal(x.view(torch.int16), r.view(torch.int16))) for x, r in zip(res[var], res[ref_v])]})
print(json.dumps(out))
