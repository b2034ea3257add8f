"""Ahead-of-time build of the gfx950 shared library (hipcc cross-compiles without a GPU).

The reference JIT-builds its CUDA op at import time with relative source paths
(VisualRWKV-v7/v7.00/src/model.py:40-43); here the library is built once, in-tree, so that it
travels with the repository to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvisualrwkv_hip.so")
ARCH = "gfx950"

SOURCES = ["wkv7_capi.hip", "probe.hip", "fused_ops.hip", "tmix_fused.hip", "attention.hip", "wkv7_step.hip", "ln_fused.hip", "wkv6_capi.hip", "loss_fused.hip"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built on this machine")
    return exe


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO_DIR, "include", "visualrwkv_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into visualrwkv_amd/libvisualrwkv_hip.so for gfx950."""
    if not force and not _stale():
        return LIB_PATH
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", CSRC, "-I", os.path.join(REPO_DIR, "include"),
           "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form", *_sources(), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
