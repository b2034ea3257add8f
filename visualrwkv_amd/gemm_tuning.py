"""Library-GEMM kernel selection for the training step.

The dense projections of the path are plain library GEMMs (hipBLASLt / rocBLAS through PyTorch).  hipBLASLt's default
heuristic is a poor pick for several of the step's shapes on gfx950 (measured on MI355X, 1.5B model, 16 x 2624 tokens:
the C x C forward GEMM 0.28 ms default vs 0.21 ms best, C x 4C 0.95 vs 0.73 ms; whole step 586 -> 533 ms), so the
benchmark loads a PyTorch TunableOp result file that names, per GEMM shape, the fastest kernel found among both
libraries.  Nothing is tuned at run time; shapes that are not in the file use the default.  The file is tied to the
library versions recorded in its `Validator` lines -- on any mismatch PyTorch ignores it and the defaults are used.

Regenerate (about 9 GPU-minutes):  bash benchmarks/tune_gemms.sh
"""
from __future__ import annotations

import os
import tempfile

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950_1b5_mb16.csv")


def enable_tuned_gemms(path: str | None = None) -> int:
    """Use the GEMM kernels listed in `path` (default: the shipped gfx950 file).  Call after the CUDA device is set.
    Returns the number of shapes loaded (0: file missing or rejected by the validators; defaults stay in use)."""
    import torch.cuda.tunable as tn
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
    return len(tn.get_results())
