"""Parity of one backward / forward variant of an experiment library against the C oracle at small shapes, one process per variant
(a faulting variant does not take the others down):  VRWKV_HIP_LIB=... python benchmarks/variant_check.py bwd 27 28   |   fwd 8"""
import subprocess
import sys


def one(kind, var):
    import torch
    from oracle import wkv7_c
    from oracle.wkv7_oracle import make_inputs, rel_rms
    from visualrwkv_amd import hip_lib
    lib = hip_lib.load()
    worst = 0.0
    for (B, T, H) in [(1, 16, 1), (2, 64, 3), (1, 384, 12), (3, 208, 5), (4, 2624, 8)]:
        w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=B * 1000 + T + H)
        yr, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
        ref = wkv7_c.backward(w, q, k, v, z, a, dy, sr, sar)
        d = [x.cuda() for x in (w, q, k, v, z, a, dy)]
        y = torch.empty_like(d[3]); s = torch.empty_like(sr).cuda(); sa = torch.empty_like(sar).cuda()
        st = torch.cuda.current_stream().cuda_stream
        if kind == "fwd":
            assert lib.vrwkv_wkv7_set_forward_variant(var) == 0
        assert lib.vrwkv_wkv7_forward_bf16(B, T, H, *[t.data_ptr() for t in d[:6]], y.data_ptr(), s.data_ptr(), sa.data_ptr(), st) == 0
        if kind == "fwd":
            torch.cuda.synchronize()
            e = max(rel_rms(y.float().cpu(), yr.float()), rel_rms(s.cpu(), sr) * 50, rel_rms(sa.cpu(), sar) * 50)
        else:
            assert lib.vrwkv_wkv7_set_backward_variant(var) == 0
            g = [torch.empty_like(d[0]) for _ in range(6)]
            assert lib.vrwkv_wkv7_backward_bf16(B, T, H, *[t.data_ptr() for t in d], s.data_ptr(), sa.data_ptr(), *[t.data_ptr() for t in g], st) == 0
            torch.cuda.synchronize()
            e = max(rel_rms(o.float().cpu(), r.float()) for o, r in zip(g, ref))
        worst = max(worst, e)
        print(f"{kind}{var} {B}x{T}x{H}: {e:.2e}", flush=True)
    print(f"{kind}{var} worst {worst:.2e} {'OK' if worst < 1e-3 else 'FAIL'}", flush=True)


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
    else:
        for v in sys.argv[2:]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", sys.argv[1], v], capture_output=True, text=True)
            print(r.stdout.strip()); 
            if r.returncode != 0:
                print(f"{sys.argv[1]}{v} rc={r.returncode}: {r.stderr.strip()[-300:]}")
