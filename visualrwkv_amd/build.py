"""Ahead-of-time build of the gfx950 shared library (hipcc cross-compiles without a GPU).

The reference JIT-builds its CUDA op at import time with relative source paths
(VisualRWKV-v7/v7.00/src/model.py:40-43); here the library is built once, in-tree, so that it
travels with the repository to the GPU box.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvisualrwkv_hip.so")
HOST_LIB_PATH = os.path.join(PKG_DIR, "libvisualrwkv_host.so")    # the op's CPU key alone: plain C++ (g++), no ROCm runtime needed to load it
HASH_PATH = LIB_PATH + ".hash"      # content hash of the sources the library was built from; travels with it
ARCH = "gfx950"

SOURCES = ["wkv7_capi.hip", "wkv7_profile.hip", "wkv7_host.hip", "probe.hip", "fused_ops.hip", "tmix_fused.hip", "attention.hip", "wkv7_step.hip", "ln_fused.hip", "wkv6_capi.hip", "loss_fused.hip", "gemv_decode.hip", "decode_fused.hip", "lora_wgrad.hip", "visual_ops.hip", "patch_embed.hip", "image_ops.hip"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built on this machine")
    return exe


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    return sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO_DIR, "include", "visualrwkv_hip.h")])


def _digest() -> str:
    """Content hash of every source the library is built from (mtimes do not survive a copy to another box)."""
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(HASH_PATH) as f:
            return f.read().strip() != _digest()
    except OSError:
        # a library without a recorded hash (built by an older version of this file): fall back to mtimes
        t = os.path.getmtime(LIB_PATH)
        return any(os.path.getmtime(d) > t for d in _deps())


# Per-source extra flags.  attention.hip: no NaN ever enters the softmax (masked logits are -1e30, not -inf), and
# without -fno-honor-nans every fmaxf of an MFMA result is preceded by a canonicalising v_max x,x,x.
# wkv7_capi.hip: the SLP vectoriser packs the WKV7 kernels' fp32 element-wise math into v_pk_* instructions, which measured no
# faster than the two scalar instructions they replace and need v_mov shuffles to form aligned register
# pairs and cannot take DPP operands (every scan step becomes v_mov_dpp + v_pk_add).  Same-box A/B, alternating processes:
# forward -1..2 %, backward (v5, v6) -1..2 %.
# tmix_fused / ln_fused / fused_ops (the streaming row kernels and AdamW): the machine scheduler's max-ILP strategy issues the next token's loads earlier in
# the token loops -- three boxes: adamw -2 / -7 / -7 %, kva_fwd and mix6_fwd -3 % on one, ln_mix1_bwd 0 / -5 / -10 %, kva_bwd -1 / -3 / -3 %, the rest within
# +-2 %; the step -1.2 ms in three same-box alternations; `max-memory-clause` = no change; attention unchanged and the WKV7 kernels +2.5 % under max-ilp (not
# applied there) -- profiles/r6o_eltwise_micro_max_ilp.json, r6n_wkv7_ab_build_flags.jsonl.
# Same instructions in another order: results are bit-identical.
_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"], "wkv7_capi.hip": ["-fno-slp-vectorize"], "wkv7_profile.hip": ["-fno-slp-vectorize"],
               "tmix_fused.hip": _ILP, "ln_fused.hip": _ILP, "fused_ops.hip": _ILP}
OBJ_DIR = os.path.join(PKG_DIR, "_build")


def _common_flags():
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", os.path.join(REPO_DIR, "include"),
            "-Wno-unused-result", "-Wno-inline-asm", "-mllvm", "-amdgpu-mfma-vgpr-form", *os.environ.get("VRWKV_EXTRA_HIPCC_FLAGS", "").split()]


def _object_for(src: str, header_digest: str) -> tuple:
    """(object path, compile command) of one source; the object name carries the hash of the source, every header and
    the flags, so an unchanged source is not recompiled."""
    flags = _common_flags() + EXTRA_FLAGS.get(os.path.basename(src), [])
    h = hashlib.sha256(header_digest.encode())
    with open(src, "rb") as f:
        h.update(f.read())
    h.update(" ".join(flags).encode())
    obj = os.path.join(OBJ_DIR, f"{os.path.basename(src)}.{h.hexdigest()[:16]}.o")
    return obj, [hipcc(), *flags, "-c", src, "-o", obj]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into visualrwkv_amd/libvisualrwkv_hip.so for gfx950 (one object per source, compiled
    in parallel, then linked).  Safe to call from several processes at once (one rank per GPU): a file lock lets one of
    them build, the library appears atomically."""
    if not force and not _stale():
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():       # another process built it while we waited
                return LIB_PATH
            from concurrent.futures import ThreadPoolExecutor
            os.makedirs(OBJ_DIR, exist_ok=True)
            hd = hashlib.sha256()
            for d in _deps():
                if d.endswith(".h"):
                    with open(d, "rb") as f:
                        hd.update(os.path.basename(d).encode() + f.read())
            jobs = [_object_for(src, hd.hexdigest()) for src in _sources()]

            def compile_one(job):
                obj, cmd = job
                if force or not os.path.exists(obj):
                    if verbose:
                        print(" ".join(cmd), file=sys.stderr)
                    tmp_obj = f"{obj}.tmp{os.getpid()}"
                    subprocess.run(cmd[:-1] + [tmp_obj], check=True)
                    os.replace(tmp_obj, obj)
                return obj

            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
                objs = list(pool.map(compile_one, jobs))
            keep = set(objs)
            for f in os.listdir(OBJ_DIR):                      # objects of older source versions
                if os.path.join(OBJ_DIR, f) not in keep:
                    os.remove(os.path.join(OBJ_DIR, f))
            tmp = f"{LIB_PATH}.tmp{os.getpid()}"
            cmd = [hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", tmp]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
            _write_atomic(HASH_PATH, _digest())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _write_atomic(path: str, text: str) -> None:
    """A reader never sees the file empty or half written (a concurrent rank would take that for "stale" and rebuild)."""
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def build_host(force: bool = False) -> str:
    """csrc/wkv7_host.hip -- plain C++ despite its suffix: the host-core implementation behind the `CPU` dispatch key of
    torch.ops.wind_backstepping -- compiled with the host compiler into a small library of its own, so that BASELINE config 1
    (fp32, no GPU) runs on a machine without ROCm.  (The same translation unit is also part of libvisualrwkv_hip.so: the C-ABI
    header declares its two entry points.)"""
    src = os.path.join(CSRC, "wkv7_host.hip")
    hdr = os.path.join(REPO_DIR, "include", "visualrwkv_hip.h")
    h = hashlib.sha256()
    for d in (src, hdr):
        with open(d, "rb") as f:
            h.update(f.read())
    digest, hash_path = h.hexdigest(), HOST_LIB_PATH + ".hash"       # content hash, as for the HIP library: mtimes do not survive a copy

    def fresh():
        if force or not os.path.exists(HOST_LIB_PATH):
            return False
        try:
            with open(hash_path) as f:
                return f.read().strip() == digest
        except OSError:
            return False
    if fresh():
        return HOST_LIB_PATH
    cxx = shutil.which("g++") or shutil.which("clang++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("no host C++ compiler (g++ / clang++) found for libvisualrwkv_host.so")
    with open(HOST_LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if fresh():                                   # another rank built it while this one waited for the lock
                return HOST_LIB_PATH
            tmp = f"{HOST_LIB_PATH}.tmp{os.getpid()}"
            subprocess.run([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-x", "c++", "-I", os.path.join(REPO_DIR, "include"),
                            src, "-o", tmp], check=True)
            os.replace(tmp, HOST_LIB_PATH)
            _write_atomic(hash_path, digest)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
