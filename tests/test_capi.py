"""The C-ABI shared library: loads here (no GPU) and exports exactly what include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "visualrwkv_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vrwkv_\w+)\s*\(", txt)))


def test_header_symbols_exported(hip_lib):
    names = _declared()
    assert "vrwkv_wkv7_forward_bf16" in names and "vrwkv_wkv7_backward_bf16" in names
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/visualrwkv_hip.h but not exported"


def test_python_prototypes_cover_header():
    from visualrwkv_amd import hip_lib as hl
    assert sorted(hl.PROTOTYPES) == _declared()


def test_argument_validation_without_gpu(hip_lib):
    """Argument errors are reported before any HIP call, so they can be checked on a CPU box."""
    assert hip_lib.vrwkv_abi_version() == 1
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    args = [p] * 10
    assert hip_lib.vrwkv_wkv7_forward_bf16(1, 15, 1, *args) == -2          # T % 16
    assert hip_lib.vrwkv_wkv7_forward_bf16(0, 16, 1, *args) == -1          # B <= 0
    assert hip_lib.vrwkv_wkv7_forward_bf16(1, 16, 1, null, *args[1:]) == -1
    mis = ctypes.c_void_p(p.value + 2)
    if p.value % 16 == 0:
        assert hip_lib.vrwkv_wkv7_forward_bf16(1, 16, 1, mis, *args[1:]) == -3
    assert b"multiple of 16" in hip_lib.vrwkv_strerror(-2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from visualrwkv_amd import hip_lib as hl
    monkeypatch.setattr(hl, "_lib", None)
    monkeypatch.setattr(hl, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        hl.load()
    except hl.HipLibraryError as e:
        assert "no PyTorch fallback" in str(e)
    else:
        raise AssertionError("expected HipLibraryError")


def test_library_older_than_its_sources_is_rebuilt_or_refused(monkeypatch):
    """A library built from other sources than the ones beside it must not be used silently (round 6: a GPU run measured the previous build of a
    kernel): with a toolchain it is rebuilt, without one loading fails."""
    import pytest
    from visualrwkv_amd import build, hip_lib as hl
    monkeypatch.delenv("VRWKV_HIP_LIB", raising=False)
    monkeypatch.setattr(hl, "_lib", None)
    monkeypatch.setattr(build, "_stale", lambda: True)
    monkeypatch.setattr(build, "hipcc", lambda: (_ for _ in ()).throw(RuntimeError("hipcc not found")))
    with pytest.raises(hl.HipLibraryError, match="other sources"):
        hl.load()
    called = []
    monkeypatch.setattr(build, "hipcc", lambda: "/opt/rocm/bin/hipcc")
    monkeypatch.setattr(build, "build", lambda *a, **k: called.append(1))
    with pytest.warns(RuntimeWarning, match="older than its sources"):
        hl.load()
    assert called == [1]


def test_op_has_a_cpu_key_and_validates_its_arguments():
    """the op is registered for the CPU key too (BASELINE config 1); wrong dtypes / shapes are errors."""
    import pytest
    import torch
    import visualrwkv_amd.wkv7 as wk  # noqa: F401
    x = torch.zeros(1, 16, 1, 64, dtype=torch.bfloat16)
    s = torch.zeros(1, 1, 1, 64, 64)
    sa = torch.zeros(1, 16, 1, 64)
    y = torch.ones_like(x)
    torch.ops.wind_backstepping.forward(x, x, x, x, x, x, y, s, sa)
    assert float(y.abs().max()) == 0.0
    with pytest.raises(TypeError):
        torch.ops.wind_backstepping.forward(x.half(), x, x, x, x, x, y, s, sa)
    with pytest.raises(TypeError):
        torch.ops.wind_backstepping.forward(x, x.float(), x, x, x, x, y, s, sa)
    with pytest.raises(ValueError):
        torch.ops.wind_backstepping.forward(x[:, :8], x, x, x, x, x, y, s, sa)
    with pytest.raises(ValueError):
        torch.ops.wind_backstepping.forward(x, x, x, x, x, x, y, s[:, :, :, :32], sa)
