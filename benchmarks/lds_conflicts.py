"""LDS bank conflicts of the WKV7 kernels, per source line, WITHOUT a GPU: the host emulator (tests/emu) runs one workgroup with every LDS
access traced (address, lane, instruction kind, code address), the accesses of a wave-instruction are put back together and priced with the
banking rules of MI355X_MICROARCH.md (section LDS): lane groups and bank modulus per instruction kind, one extra LDS cycle per extra
distinct dword on a busy bank within a group.

    python benchmarks/lds_conflicts.py bwd9 | bwd8 | fwd7 | fwd4 | fwd6   [T=128] [--json out.json]

Output: per site (file:line of the access and of its callers), wave role, kind, wave-instructions per step / chunk, ideal and extra cycles.
What it does NOT model: the "further conflict classes" the guide mentions for ds_read_b64_tr_b16, the landing pattern of LDS-DMA (priced
like ds_write_b128), and conflicts between different instructions.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the real kernel
(profiles/r5_wkv7_pmc_b16.txt) are the check."""
import collections
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EMU = os.path.join(ROOT, "tests", "emu")
SO = os.path.join(ROOT, "benchmarks", "_alt", "libemu_trace.so")
KINDS = ["read_b32", "read_b64", "read_b128", "read_b64_tr_b16", "write_b32", "write_b64", "write_b128", "dma16"]
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
HALVES = [list(range(32)), list(range(32, 64))]
QUARTERS = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
EIGHTHS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
# kind -> (lane groups, dwords per lane, bank modulus)
RULES = {0: (HALVES, 1, 32), 1: (HALVES, 2, 64), 2: (B128_GROUPS, 4, 64), 3: (HALVES, 2, 64), 4: (HALVES, 1, 32), 5: (QUARTERS, 2, 32),
         6: (EIGHTHS, 4, 32), 7: (EIGHTHS, 4, 32)}


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(EMU, "emu_wkv7.cpp"), os.path.join(EMU, "emu_lds_trace.cpp")]
    deps = srcs + [os.path.join(EMU, f) for f in ("hip_emu.h", "gfx950_prims.h")] + [os.path.join(ROOT, "visualrwkv_amd", "csrc", f)
                                                                                         for f in os.listdir(os.path.join(ROOT, "visualrwkv_amd", "csrc")) if f.endswith(".h")]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-DEMU_LDS_TRACE", "-I", EMU, "-I", os.path.join(ROOT, "visualrwkv_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "benchmarks", "experiments"), *srcs, "-o", SO, "-ldl"]
    subprocess.run(cmd, check=True)
    return SO


def cycles(kind, lanes_off):
    """(ideal, extra) LDS-array cycles of one wave-instruction; lanes_off: {lane: byte offset}."""
    groups, ndw, mod = RULES[kind]
    ideal = extra = 0
    for g in groups:
        banks = collections.defaultdict(set)
        for l in g:
            if l in lanes_off:
                for d in range(ndw):
                    dw = lanes_off[l] // 4 + d
                    banks[dw % mod].add(dw)
        if banks:
            ideal += 1
            extra += max(len(v) for v in banks.values()) - 1
    return ideal, extra


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "bwd9"
    T = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 128
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    lib = ctypes.CDLL(build())
    lib.emu_lds_trace_base.restype = ctypes.c_uint64
    lib.emu_lds_trace_count.restype = ctypes.c_long
    from oracle.wkv7_oracle import make_inputs
    from oracle import wkv7_c
    w, q, k, v, z, a, dy = make_inputs(1, T, 1, seed=3)
    y, s, sa = wkv7_c.forward(w, q, k, v, z, a)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.emu_lds_trace_enable(1)
    if what.startswith("fwd"):
        var = int(what[3:])
        yo, so, sao = torch.empty_like(y), torch.empty_like(s), torch.empty_like(sa)
        lib.emu_wkv7_forward(1, T, 1, P(w), P(q), P(k), P(v), P(z), P(a), P(yo), P(so), P(sao), var)
        roles = {i: ("consumer" if i < 4 else "producer") for i in range(8)}
    else:
        mode = {"bwd8": 9, "bwd9": 10, "bwd5": 6, "bwd9opt": 13, "bwd8opt": 14}[what]
        g = [torch.empty_like(w) for _ in range(6)]
        lib.emu_wkv7_backward_chunked(1, T, 1, P(w), P(q), P(k), P(v), P(z), P(a), P(dy), P(s), P(sa), *[P(x) for x in g], mode)
        roles = {i: "IJP"[i // 4] for i in range(12)} if what != "bwd5" else {i: ("consumer" if i < 4 else "producer") for i in range(8)}
    lib.emu_lds_trace_enable(0)
    n = lib.emu_lds_trace_count()
    off = np.empty(n, np.uint32); tid = np.empty(n, np.uint16); kind = np.empty(n, np.uint16); pc = np.empty(n, np.uint64)
    lib.emu_lds_trace_copy(off.ctypes.data_as(ctypes.c_void_p), tid.ctypes.data_as(ctypes.c_void_p), kind.ctypes.data_as(ctypes.c_void_p),
                           pc.ctypes.data_as(ctypes.c_void_p))
    pc -= np.uint64(lib.emu_lds_trace_base())
    # put wave-instructions back together: (wave, pc, kind, k-th execution by that lane)
    occ = collections.Counter()
    inst = collections.defaultdict(dict)
    for i in range(n):
        t = int(tid[i]); key = (t, int(pc[i]), int(kind[i]))
        kth = occ[key]; occ[key] += 1
        inst[(t >> 6, int(pc[i]), int(kind[i]), kth)][t & 63] = int(off[i])
    site = collections.defaultdict(lambda: [0, 0, 0])       # (role, pc, kind) -> [wave-instructions, ideal, extra]
    for (wave, p_, kd, _), lanes in inst.items():
        i_, e_ = cycles(kd, lanes)
        r = site[(roles[wave], p_, kd)]
        r[0] += 1; r[1] += i_; r[2] += e_
    pcs = sorted({p_ for (_, p_, _) in site})
    sym = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + SO, "--inlines", "--functions=none", *[hex(p_) for p_ in pcs]],
                         capture_output=True, text=True, check=True).stdout.strip().split("\n\n")
    where = {}
    for p_, blk in zip(pcs, sym):
        frames = [ln.strip() for ln in blk.strip().split("\n") if ln.strip()]
        frames = [os.path.basename(f.rsplit(":", 1)[0]) for f in frames]                 # file:line (drop the column)
        frames = [f for f in frames if not f.startswith(("hip_emu.h", "gfx950_prims.h"))] or frames
        where[p_] = " <- ".join(frames[:3])
    nchunk = T // 16
    rows = []
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for (role, p_, kd), (cnt, ideal, extra) in site.items():
        r = agg[(role, where[p_], KINDS[kd])]
        r[0] += cnt; r[1] += ideal; r[2] += extra
    for (role, wh, kd), (cnt, ideal, extra) in agg.items():
        rows.append({"role": role, "site": wh, "kind": kd, "wave_instr_per_chunk": cnt / nchunk, "ideal_cycles_per_chunk": ideal / nchunk,
                     "extra_cycles_per_chunk": extra / nchunk})
    rows.sort(key=lambda r: -r["extra_cycles_per_chunk"])
    tot_i = sum(r["ideal_cycles_per_chunk"] for r in rows); tot_e = sum(r["extra_cycles_per_chunk"] for r in rows)
    print(f"{what} T={T}: LDS-array cycles per chunk (whole workgroup): ideal {tot_i:.0f} + conflicts {tot_e:.0f} = {tot_e / (tot_i + tot_e):.1%} of active")
    by_kind = collections.defaultdict(lambda: [0.0, 0.0])
    for r in rows:
        by_kind[r["kind"]][0] += r["ideal_cycles_per_chunk"]; by_kind[r["kind"]][1] += r["extra_cycles_per_chunk"]
    for kd, (i_, e_) in sorted(by_kind.items(), key=lambda kv: -kv[1][1]):
        print(f"   {kd:18s} ideal {i_:7.1f}  extra {e_:7.1f}")
    print(f"{'role':9s} {'kind':16s} {'instr/chunk':>11s} {'ideal':>7s} {'extra':>7s}  site")
    for r in rows:
        if r["extra_cycles_per_chunk"] >= 0.5:
            print(f"{r['role']:9s} {r['kind']:16s} {r['wave_instr_per_chunk']:11.1f} {r['ideal_cycles_per_chunk']:7.1f} {r['extra_cycles_per_chunk']:7.1f}  {r['site']}")
    if out_json:
        json.dump({"kernel": what, "T": T, "ideal_cycles_per_chunk": tot_i, "extra_cycles_per_chunk": tot_e, "sites": rows}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
