"""Numerical check of every kernel a TunableOp result file selects: run the GEMM of each entry with the selection on
and off and compare (TunableOp does not verify results when it tunes).  Entries whose result differs are dropped.
    python benchmarks/check_tuned_gemms.py visualrwkv_amd/tuning/tunableop_gfx950_1b5_mb16.csv [--write]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind, layout, m, n, k, gen):
    dev = "cuda"
    r = lambda *s: (torch.randn(*s, device=dev, generator=gen) * 0.5).bfloat16()
    if layout == "tn":                       # F.linear forward: (n,k) x (m,k)^T
        x, w = r(n, k), r(m, k)
        b = r(m) if kind.startswith("GemmAndBias") else None
        return lambda: F.linear(x, w, b)
    if layout == "nn":                       # dgrad: (n,k) x (k,m)
        dy, w = r(n, k), r(k, m)
        return lambda: dy @ w
    if layout == "nt":                       # wgrad: (k,n)^T x (k,m)
        dy, x = r(k, n), r(k, m)
        return lambda: dy.t() @ x
    raise ValueError(layout)


def main():
    path = sys.argv[1]
    write = "--write" in sys.argv
    import torch.cuda.tunable as tn
    from visualrwkv_amd.gemm_tuning import enable_tuned_gemms
    lines = open(path).read().splitlines()
    n_loaded = enable_tuned_gemms(path)
    print("loaded", n_loaded)
    keep, bad = [], []
    for ln in lines:
        if ln.startswith("Validator") or not ln.strip():
            keep.append(ln)
            continue
        kind, key, sol, t = ln.split(",")
        if sol == "Default":
            keep.append(ln)
            continue
        layout, m, n, k = key.split("_")[:4]
        m, n, k = int(m), int(n), int(k)
        gen = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
        fn = run(kind, layout, m, n, k, gen)
        tn.enable(False)
        ref = fn().float()
        tn.enable(True)
        got = fn().float()
        torch.cuda.synchronize()
        err = ((got - ref).norm() / ref.norm()).item() if torch.isfinite(got).all() else float("nan")
        ok = err == err and err < 2e-2
        print(f"{'ok ' if ok else 'BAD'} {kind:34s} {key:44s} {sol:22s} rel err {err:.2e}")
        (keep if ok else bad).append(ln)
    print(f"{len(bad)} bad entries")
    if write and bad:
        open(path, "w").write("\n".join(keep) + "\n")
        print("rewrote", path)


if __name__ == "__main__":
    main()
