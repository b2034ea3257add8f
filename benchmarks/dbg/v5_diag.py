import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wkv7_c
from oracle.wkv7_oracle import make_inputs, rel_rms
from visualrwkv_amd import hip_lib
lib = hip_lib.load()
NAMES = ["dw", "dq", "dk", "dv", "dz", "da"]
dev = "cuda:0"
for variant in (7, 8):
    for (B, T, H) in [(1, 32, 1), (1, 64, 1), (2, 64, 3)]:
        w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=B * 77 + T + H)
        _, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
        ref = wkv7_c.backward(w, q, k, v, z, a, dy, sr, sar)
        lib.vrwkv_wkv7_set_backward_variant(variant)
        d = [x.to(dev) for x in (w, q, k, v, z, a, dy, sr, sar)]
        for rep in range(2):
            outs = [torch.full_like(d[0], float("nan")) for _ in range(6)]
            rc = lib.vrwkv_wkv7_backward_bf16(B, T, H, *[x.data_ptr() for x in d], *[o.data_ptr() for o in outs], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            line = []
            for n, o, r in zip(NAMES, outs, ref):
                o = o.float().cpu(); r = r.float()
                per_chunk = [(float(rel_rms(o[:, c * 16:(c + 1) * 16], r[:, c * 16:(c + 1) * 16]))) for c in range(T // 16)]
                line.append(n + ":" + ",".join("%.0e" % e for e in per_chunk))
            print(variant, (B, T, H), rep, " ".join(line), flush=True)
        lib.vrwkv_wkv7_set_backward_variant(-1)
