"""Library GEMMs run on more than one HIP stream only for keys listed beside the loaded tuning file (visualrwkv_amd/gemm_tuning.py): two stream-K
kernels of the library on three streams hang the GPU (round 6), so everything that is not listed must run one GEMM after the other."""
import os

import pytest
import torch

from visualrwkv_amd import gemm_tuning


def test_nothing_is_concurrent_without_a_loaded_tuning_file():
    assert gemm_tuning._CONCURRENT_OK == {} or torch.cuda.is_available()
    assert not gemm_tuning.concurrent_ok("3x tn_768_67200_768")
    assert not gemm_tuning.concurrent_ok()
    assert gemm_tuning.concurrency_report()["3x tn_768_67200_768"] is False


def test_sidecar_lists_keys_and_ignores_comments(tmp_path):
    f = tmp_path / "t.csv"
    f.write_text("Validator,PT_VERSION,0\n")
    (tmp_path / "t.csv.concurrent").write_text("# a comment\n3x tn_8_16_8\n\nvit a:1x3x4x4 b:1x3x4x4 !one-rank\n")
    assert gemm_tuning._read_sidecar(str(f)) == {"3x tn_8_16_8": False, "vit a:1x3x4x4 b:1x3x4x4": True}
    assert gemm_tuning._read_sidecar(str(tmp_path / "missing.csv")) == {}


def test_keys_marked_one_rank_are_withdrawn_when_a_multi_rank_engine_exists(monkeypatch):
    monkeypatch.setattr(gemm_tuning, "_CONCURRENT_OK", {"3x tn_8_16_8": False, "vit a b": True})
    monkeypatch.setattr(gemm_tuning, "_COLLECTIVES", False)
    assert gemm_tuning.concurrent_ok("3x tn_8_16_8") and gemm_tuning.concurrent_ok("vit a b")
    gemm_tuning.note_collectives(True)
    assert gemm_tuning.concurrent_ok("3x tn_8_16_8") and not gemm_tuning.concurrent_ok("vit a b")
    assert not gemm_tuning.concurrent_ok("3x tn_8_16_8", "vit a b")


def test_shipped_sidecar_names_only_shapes_of_the_shipped_tuning_file():
    keys = gemm_tuning._read_sidecar(gemm_tuning.DEFAULT_FILE)
    assert keys, "the shipped tuning file has its checked keys"
    sigs = open(gemm_tuning.DEFAULT_FILE).read()
    for k in keys:
        if k.startswith("3x "):
            assert k[3:] + "_ld_" in sigs, f"{k}: the shape must be pinned by the tuning file it rides on"


@pytest.mark.gpu
def test_unlisted_shape_runs_one_gemm_after_the_other_and_matches():
    """A shape outside the list (the library's default kernel for it is the stream-K one that hung three streams): linear3 must not use the
    side streams, and its outputs / gradients equal three separate Linear layers."""
    from visualrwkv_amd import fused
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    M, C = 8192, 768
    mods = [torch.nn.Linear(C, C, bias=False, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    xs = [torch.randn(M, C, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3)]
    outs = fused.linear3(mods, xs)
    assert gemm_tuning.concurrency_report()[f"3x tn_{C}_{M}_{C}"] is False
    sum(o.float().square().sum() for o in outs).backward()
    got = [x.grad.clone() for x in xs] + [m.weight.grad.clone() for m in mods]
    for t in xs + [m.weight for m in mods]:
        t.grad = None
    ref = [torch.nn.functional.linear(x, m.weight) for x, m in zip(xs, mods)]
    sum(o.float().square().sum() for o in ref).backward()
    for o, r in zip(outs, ref):
        assert torch.equal(o, r)
    for g, t in zip(got, xs + [m.weight for m in mods]):
        assert torch.allclose(g.float(), t.grad.float(), rtol=2e-2, atol=2e-2 * float(t.grad.float().abs().max()))
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_listed_shape_runs_on_three_streams_with_the_shipped_file():
    """The headline's r/k/v shape with the shipped kernels: three streams, same numbers as one stream (and it finishes)."""
    from visualrwkv_amd import fused
    import torch.cuda.tunable as tn
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n = gemm_tuning.enable_tuned_gemms()
    try:
        if n == 0:
            pytest.skip("the shipped tuning file does not match this box's library versions: nothing is concurrent, nothing to check")
        M, C = 41984, 2048
        torch.manual_seed(1)
        mods = [torch.nn.Linear(C, C, bias=False, device=dev, dtype=torch.bfloat16) for _ in range(3)]
        xs = [torch.randn(M, C, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3)]
        outs = fused.linear3(mods, xs)
        assert gemm_tuning.concurrency_report()[f"3x tn_{C}_{M}_{C}"] is True
        ref = [torch.nn.functional.linear(x, m.weight) for x, m in zip(xs, mods)]
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)
        torch.cuda.synchronize()
    finally:
        tn.enable(False)
        gemm_tuning._CONCURRENT_OK = {}
