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


def group_bias(x: torch.Tensor, ref: torch.Tensor, name: str = "", max_scale_err: float = 5e-3, min_cos_gap: float = None):
    """A SYSTEMATIC error in one tensor -- a gradient group scaled by 1.01, a missing term -- hides inside a loose rel-RMS bound that
    has to admit the random rounding noise of a bf16 pipeline (1e-2 .. 3e-2 against fp32).  The noise is unbiased, a systematic error
    is not: the least-squares scale of x against ref, <x, ref> / <ref, ref>, is 1 up to noise / sqrt(n) for the former and off by
    the error for the latter.  Returns (scale - 1, 1 - cosine)."""
    xr = x.detach().double().cpu().reshape(-1)
    rr = ref.detach().double().cpu().reshape(-1)
    assert xr.shape == rr.shape, (name, xr.shape, rr.shape)
    rr2 = float((rr * rr).sum())
    scale = float((xr * rr).sum()) / max(rr2, 1e-300)
    cos = float((xr * rr).sum()) / max(float(xr.norm() * rr.norm()), 1e-300)
    if NOTES:
        print(f"[parity] {name}: scale-1 {scale - 1:+.2e}  1-cos {1 - cos:.2e}")
    assert abs(scale - 1) < max_scale_err, f"{name}: least-squares scale against the reference is {scale:.5f} (limit 1 +- {max_scale_err})"
    if min_cos_gap is not None:
        assert 1 - cos < min_cos_gap, f"{name}: 1 - cosine {1 - cos:.3e} >= {min_cos_gap}"
    return scale - 1, 1 - cos
