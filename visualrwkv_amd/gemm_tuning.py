"""Library-GEMM kernel selection for the training step.

The dense projections of the path are plain library GEMMs (hipBLASLt / rocBLAS through PyTorch).  hipBLASLt's default
heuristic is a poor pick for several of the step's shapes on gfx950 (measured on MI355X, 1.5B model, 16 x 2624 tokens:
the C x C forward GEMM 0.28 ms default vs 0.21 ms best, C x 4C 0.95 vs 0.73 ms; whole step 586 -> 533 ms), so the
benchmark loads a PyTorch TunableOp result file that names, per GEMM shape, the fastest kernel found among both
libraries.  Nothing is tuned at run time; shapes that are not in the file use the default.  The file is tied to the
library versions recorded in its `Validator` lines -- on any mismatch PyTorch ignores it and the defaults are used.

Regenerate (about 9 GPU-minutes):  bash benchmarks/tune_gemms.sh

Library GEMMs on more than one HIP stream
-----------------------------------------
hipBLASLt's default pick for many large shapes on gfx950 is a stream-K kernel (`..._SK3_...`): a persistent grid whose workgroups spin
on partial tiles of workgroups that may not be resident yet.  One such kernel next to ordinary kernels is fine; TWO of them on two
streams deadlock the GPU (round 6: three F.linear of 67 200 x 768 x 768, or of 41 984 x 2048 x 2048 with the default heuristic, never
finish -- benchmarks/concurrent_gemm_probe.py; with this file's kernels for the same shape they run side by side.  Two such grids of <= 256
workgroups fit the 512 resident slots of the chip together; with three streams the dispatcher leaves each one part-resident and all of them wait).
The package's own kernels never wait on another workgroup, so one library GEMM beside them is always safe.  So the package
runs library GEMMs concurrently only for the keys listed in the sidecar `<tuning file>.concurrent` -- shapes (and tower
configurations) whose pinned kernels were run side by side on an MI355X -- and only while that tuning file is the one loaded.
`concurrent_ok()` is False for everything else: any other shape, any other library version, no tuning file.
"""
from __future__ import annotations

import os
import tempfile

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950_1b5_mb16.csv")
_CONCURRENT_OK: dict = {}          # key -> True when granted only without collectives on another stream (a trailing " !one-rank" in the sidecar)
_COLLECTIVES = False               # a multi-rank ZeRO-1 engine drives RCCL kernels on its own stream (dp.Zero1Engine sets this)
_ASKED: dict = {}


def note_collectives(active: bool = True) -> None:
    """dp.Zero1Engine with more than one rank: keys marked `!one-rank` (their kernels are stream-K grids that need most of the chip's resident slots;
    beside RCCL kernels that is not validated -- there is no multi-GPU box in this project's pool) are no longer granted."""
    global _COLLECTIVES
    _COLLECTIVES = bool(active)


def concurrent_ok(*keys: str) -> bool:
    """True when every key is listed as checked in the loaded tuning file's `.concurrent` sidecar (see the module docstring)."""
    def granted(k):
        return k in _CONCURRENT_OK and not (_CONCURRENT_OK[k] and _COLLECTIVES)
    for k in keys:
        _ASKED[k] = granted(k)
    return bool(keys) and all(granted(k) for k in keys)


def concurrency_report() -> dict:
    """{key: granted} for every key the package asked about so far (bench.py prints it; a key that is False ran one GEMM after the other)."""
    return dict(_ASKED)


def _read_sidecar(path: str) -> dict:
    try:
        with open(path + ".concurrent") as f:
            lines = [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
    except OSError:
        return {}
    return {ln.removesuffix("!one-rank").strip(): ln.endswith("!one-rank") for ln in lines}


def enable_tuned_gemms(path: str | None = None) -> int:
    """Use the GEMM kernels listed in `path` (default: the shipped gfx950 file).  Call after the CUDA device is set.
    Returns the number of shapes loaded (0: file missing or rejected by the validators; defaults stay in use)."""
    import torch.cuda.tunable as tn
    global _CONCURRENT_OK
    _CONCURRENT_OK = {}
    path = path or DEFAULT_FILE
    if not os.path.exists(path):
        return 0
    tn.enable(True)
    tn.tuning_enable(False)
    # results are only read; keep PyTorch's write-on-exit away from the working directory
    tn.set_filename(os.path.join(tempfile.gettempdir(), f"vrwkv_tunableop_{os.getpid()}.csv"))
    if not tn.read_file(path):
        tn.enable(False)
        return 0
    # VRWKV_CONCURRENT_TRY="key;key": extra keys for the run that checks them on the hardware before they are listed (a wrong one can hang the GPU)
    _CONCURRENT_OK = {**_read_sidecar(path), **{k.strip(): False for k in os.environ.get("VRWKV_CONCURRENT_TRY", "").split(";") if k.strip()}}
    return len(tn.get_results())
