"""Shared parity metric of the GPU tests: a bf16 result against an oracle value ROUNDED ONCE to bf16 (north_star: 1e-3 in
bf16; the same statement test_oracle_wkv7.py makes about the two oracle restatements).  Two numbers: the rel-RMS distance
`||x - bf16(ref)|| / ||bf16(ref)||` and the fraction of elements that differ at all ("flips": a result that is right to
fp32 accuracy differs from the rounded oracle only where the exact value sits next to a rounding boundary).  A systematic
error of a few bf16 ulps passes a loose rel-RMS bound against the UNROUNDED oracle (one rounding is 1.65e-3 there); it
cannot pass these two."""
import os

import torch

NOTES = os.environ.get("VRWKV_TEST_NOTES") == "1"


def bf16_close(x: torch.Tensor, ref: torch.Tensor, name: str = "", tol: float = 1e-3, max_flip: float = 0.10):
    xr = x.detach().float().cpu().reshape(-1)
    rr = ref.detach().to(torch.float64).cpu().reshape(-1).float().bfloat16().float()
    assert xr.shape == rr.shape, (name, xr.shape, rr.shape)
    rms = float((xr - rr).double().norm() / rr.double().norm().clamp_min(1e-30))
    flip = float((xr != rr).float().mean())
    if NOTES:
        print(f"[parity] {name}: rel_rms {rms:.3e} flips {flip:.4f}")
    assert rms < tol, f"{name}: rel-RMS {rms:.3e} >= {tol:.1e} vs the bf16-rounded oracle (flips {flip:.4f})"
    assert flip < max_flip, f"{name}: {flip:.4f} of the elements differ from the bf16-rounded oracle (limit {max_flip})"
    return rms, flip
