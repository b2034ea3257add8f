"""Why the same WKV7 kernel takes 0.98 ms in one collection and 1.08 ms in another: the chip clocks to its power budget
(MI355X_MICROARCH.md, "DVFS give-back"), and switching power depends on the DATA.  Same binary, same shape (B x 2624 x 32
heads), three input sets -- the micro-benchmark's random inputs, all-zero inputs, the random inputs scaled by 1/64 (fewer
mantissa/exponent bits toggling in the products) -- timed with HIP events, plus the shader clock the backward actually ran
at (shader cycles of workgroup 0 over its life on the constant 100 MHz counter, profiling build of the same kernel)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.wkv7_micro import synth_inputs  # noqa: E402
from visualrwkv_amd import hip_lib  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=30).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)"))]
        return keep[:8]
    except Exception as e:  # noqa: BLE001
        return [repr(e)]


def run(B=16, T=2624, H=32, iters=20):
    lib = hip_lib.load()
    dev = "cuda:0"
    base = synth_inputs(B, T, H, dev)
    sets = {"random (micro-benchmark inputs)": base, "zeros": [torch.zeros_like(x) for x in base],
            "random / 64": [(x.float() / 64).bfloat16() if i != 0 else x for i, x in enumerate(base)]}
    y = torch.empty_like(base[3]); s = torch.empty(B, H, T // 16, 64, 64, device=dev); sa = torch.empty(B, T, H, 64, device=dev)
    g = [torch.empty_like(base[0]) for _ in range(6)]
    st = torch.cuda.current_stream().cuda_stream
    out = {"B": B, "smi_before": smi(), "cases": {}}
    for name, (w, q, k, v, z, a, dy) in sets.items():
        def fwd():
            assert lib.vrwkv_wkv7_forward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, a, y, s, sa)], st) == 0

        def bwd():
            assert lib.vrwkv_wkv7_backward_bf16(B, T, H, *[t.data_ptr() for t in (w, q, k, v, z, a, dy, s, sa, *g)], st) == 0

        def t(fn):
            best = 1e9
            for _ in range(3):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / iters)
            return best
        fwd()
        rec = {"fwd_ms": round(t(fwd), 4), "bwd_ms": round(t(bwd), 4)}
        # shader clock during the backward (profiling build of the product kernel, wkv7_bwd_v8.h; until round 5 this probe stamped the round-3 kernel): cycles of workgroup 0's
        # I wave 0 over its life / that life in 10 ns ticks of the constant 100 MHz counter
        dbg = torch.zeros(32, dtype=torch.int64, device=dev)
        for _ in range(3):
            dbg.zero_()
            assert lib.vrwkv_wkv7_profile_bf16(4, B, T, H, *[t_.data_ptr() for t_ in (w, q, k, v, z, a, dy, y, s, sa, *g)], dbg.data_ptr(), st) == 0
            torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        if d[15] > 0:
            cyc = sum(d[0:5]) + d[18] + d[19]
            rec["bwd_shader_clock_GHz"] = round(cyc / (d[15] * 10.0), 3)
            rec["bwd_cycles_per_step_wg0"] = round(cyc / (T // 16 + 3))
        out["cases"][name] = rec
    out["smi_after"] = smi()
    return out


if __name__ == "__main__":
    print(json.dumps(run(B=int(sys.argv[1]) if len(sys.argv) > 1 else 16)))
