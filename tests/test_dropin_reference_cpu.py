"""The drop-in claim itself: the reference's own `src/model.py` -- its WindBackstepping, RUN_CUDA_RWKV7g, RWKV_Tmix_x070,
RWKV_CMix_x070, Block and RWKV, unchanged -- runs on top of the operator this package registers when the reference's import-time JIT build of its CUDA
sources is replaced by `import visualrwkv_amd.wkv7`, takes this package's state dict and agrees with this package's mirror of the same classes.  Only
where /root/reference exists (this container; not on the GPU box): the reference is imported at run time by a subprocess, nothing of it is stored."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference/VisualRWKV-v7/v7.00/src/model.py"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference checkout is not on this machine")
def test_reference_model_py_runs_on_this_operator():
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "dropin_reference_run.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["finite"] and rec["logits_shape"] == [2, 37, 512] and rec["same_grad_keys"] and rec["n_param_grads"] >= 40
    # both stacks call the SAME operator (the host-core kernels) with the same eager bf16 arithmetic around it: observed bit-identical (0.0)
    assert rec["logits_rel"] < 1e-6 and rec["dx_rel"] < 1e-6 and rec["worst_param_grad_rel"] < 1e-6, rec
