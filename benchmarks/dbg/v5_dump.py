"""GPU side: dump v5 register snapshots of workgroup 0 (variant 10) to gpurun_out/v5_dump_gpu.pt ;
host side (--emu): the same from the emulator, then compare."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wkv7_c
from oracle.wkv7_oracle import make_inputs
B, T, H = 1, 32, 1
w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=B * 77 + T + H)
_, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
P = lambda t: ctypes.c_void_p(t.data_ptr())
if "--emu" in sys.argv:
    from tests.emu.build import build_emu
    lib = ctypes.CDLL(build_emu())
    outs = [torch.zeros_like(w) for _ in range(6)]
    dbg = torch.zeros(2, 32, 256, 4)
    lib.emu_wkv7_backward_v5_dump(B, T, H, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(sr), P(sar), *[P(o) for o in outs], P(dbg))
    gpu = torch.load(os.path.join(ROOT, "gpurun_out", "v5_dump_gpu.pt"))
    names = ["S0_0","S0_1","S0_2","S0_3","dU0","dU1","dU2","dU3","dZt2","dQt2","dAh2","dKh2","dS2_0","dS2_1","dS2_2","dS2_3","qzh","qzl","dZt3","dQt3","dAh3","dKh3","x8_0","x8_1","x8_2","x8_3","xl0","xl1","xl2","xl3"]
    for it in range(2):
        for sl, n in enumerate(names):
            e, g_ = dbg[it, sl], gpu[it, sl]
            if n in ("qzh", "qzl") or n.startswith("x"):
                bad = (e.view(torch.int32) != g_.view(torch.int32)).sum().item()
                print(it, n, "bit mismatches", bad, "per dword col", (e.view(torch.int32) != g_.view(torch.int32)).sum(0).tolist(), "lanes", (e.view(torch.int32) != g_.view(torch.int32)).any(1).nonzero().flatten().tolist()[:40])
            else:
                d = (e - g_).abs().max().item(); m = e.abs().max().item()
                print(it, n, "max|diff| %.3e  max|ref| %.3e" % (d, m))
else:
    from visualrwkv_amd import hip_lib
    lib = hip_lib.load()
    dev = "cuda:0"
    d = [x.to(dev) for x in (w, q, k, v, z, a, dy)]
    s, sa = sr.to(dev), sar.to(dev)
    outs = [torch.zeros_like(d[0]) for _ in range(6)]
    y = torch.zeros_like(d[0])
    dbg = torch.zeros(2, 32, 256, 4, device=dev)
    lib.vrwkv_wkv7_set_backward_variant(10)
    rc = lib.vrwkv_wkv7_profile_bf16(1, B, T, H, *[x.data_ptr() for x in d], y.data_ptr(), s.data_ptr(), sa.data_ptr(),
                                     *[o.data_ptr() for o in outs], dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    torch.save(dbg.cpu(), os.path.join(ROOT, "gpurun_out", "v5_dump_gpu.pt"))
    print("saved")
