"""ctypes binding of libvisualrwkv_hip.so (C-ABI declared in include/visualrwkv_hip.h).

There is deliberately no fallback: if the library is missing or a symbol cannot be resolved the
import of the operator fails with an error that says how to build it.
"""
from __future__ import annotations

import ctypes
import os

from .build import HOST_LIB_PATH, LIB_PATH

_c_int, _c_void_p, _c_long, _c_float = ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_float

# symbol -> (restype, argtypes); must list every function include/visualrwkv_hip.h declares
PROTOTYPES = {
    "vrwkv_abi_version": (_c_int, []),
    "vrwkv_strerror": (ctypes.c_char_p, [_c_int]),
    "vrwkv_wkv7_forward_bf16": (_c_int, [_c_int] * 3 + [_c_void_p] * 10),
    "vrwkv_wkv7_backward_bf16": (_c_int, [_c_int] * 3 + [_c_void_p] * 16),
    "vrwkv_wkv7_backward_segments_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 18),
    "vrwkv_wkv7_forward_host": (_c_int, [_c_int] * 4 + [_c_void_p] * 9 + [_c_int]),
    "vrwkv_wkv7_backward_host": (_c_int, [_c_int] * 4 + [_c_void_p] * 15 + [_c_int]),
    "vrwkv_wkv6_ckpt_floats": (_c_long, [_c_int] * 3),
    "vrwkv_wkv6_forward_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 8),
    "vrwkv_wkv6_backward_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 13),
    "vrwkv_wkv6_set_backward_variant": (_c_int, [_c_int]),
    "vrwkv_add_ln_ws_floats": (_c_long, [_c_long, _c_int]),
    "vrwkv_add_ln_fwd_bf16": (_c_int, [_c_long, _c_int, _c_float] + [_c_void_p] * 9),
    "vrwkv_add_ln_scaled_fwd_bf16": (_c_int, [_c_long, _c_int, _c_float] + [_c_void_p] * 8),
    "vrwkv_add_ln_bwd_bf16": (_c_int, [_c_long, _c_int] + [_c_void_p] * 10),
    "vrwkv_ln_mix_ws_floats": (_c_long, [_c_long, _c_int, _c_int]),
    "vrwkv_ln_mix_fwd_bf16": (_c_int, [_c_long, _c_int, _c_int, ctypes.c_float, _c_int] + [_c_void_p] * 10),
    "vrwkv_ln_mix_bwd_bf16": (_c_int, [_c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 14),
    "vrwkv_mix_bwd_ln_bf16": (_c_int, [_c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 12),
    "vrwkv_ce_fwd_bf16": (_c_int, [_c_long, _c_int] + [_c_void_p] * 7),
    "vrwkv_ce_bwd_bf16": (_c_int, [_c_long, _c_int] + [_c_void_p] * 6 + [_c_float] + [_c_void_p] * 2),
    "vrwkv_gemv_multi_bf16": (_c_int, [_c_int, _c_int] + [_c_void_p] * 8),
    "vrwkv_decode_ln_mix_bf16": (_c_int, [_c_int] * 3 + [_c_void_p] * 3 + [_c_float] + [_c_void_p] * 4),
    "vrwkv_decode_tmix_head_bf16": (_c_int, [_c_int] * 2 + [_c_void_p] * 15 + [_c_float] + [_c_void_p] * 5),
    "vrwkv_gemv_multi_copy_bf16": (_c_int, [_c_int, _c_int] + [_c_void_p] * 9 + [_c_long] + [_c_void_p]),
    "vrwkv_gemv_ln_multi_bf16": (_c_int, [_c_int] * 3 + [_c_void_p] * 4 + [_c_float] + [_c_void_p] * 7),
    "vrwkv_wkv7_forward_state_bf16": (_c_int, [_c_int] * 3 + [_c_void_p] * 12),
    "vrwkv_wkv7_step_bf16": (_c_int, [_c_int] * 2 + [_c_void_p] * 9),
    "vrwkv_wkv7_set_forward_variant": (_c_int, [_c_int]),
    "vrwkv_wkv7_set_backward_variant": (_c_int, [_c_int]),
    "vrwkv_wkv7_last_variant": (_c_int, [_c_int]),
    "vrwkv_wkv7_set_backward_slice_limit": (_c_int, [ctypes.c_ulonglong]),
    "vrwkv_wkv7_resolve_variant": (_c_int, [_c_int] * 4),
    "vrwkv_mix_fwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 4),
    "vrwkv_mix_fwd_prev_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 5),
    "vrwkv_param_grad_ws_floats": (ctypes.c_long, [ctypes.c_long, _c_int, _c_int]),
    "vrwkv_mix_bwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 7),
    "vrwkv_mix_bwd2_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int, _c_int] + [_c_void_p] * 8),
    "vrwkv_kva_bwd2_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 23),
    "vrwkv_kva_bwd3_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 24),
    "vrwkv_ddmix_fwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 5),
    "vrwkv_ddmix_bwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 9),
    "vrwkv_gn_silu_fwd_bf16": (_c_int, [ctypes.c_long, _c_int, ctypes.c_float] + [_c_void_p] * 6),
    "vrwkv_gn_silu_bwd_bf16": (_c_int, [ctypes.c_long, _c_int, ctypes.c_float] + [_c_void_p] * 10),
    "vrwkv_decay_fwd_bf16": (_c_int, [ctypes.c_long, _c_int] + [_c_void_p] * 4),
    "vrwkv_decay_bwd_bf16": (_c_int, [ctypes.c_long, _c_int] + [_c_void_p] * 7),
    "vrwkv_kva_fwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 14),
    "vrwkv_kva_bwd_bf16": (_c_int, [ctypes.c_long, _c_int, _c_int] + [_c_void_p] * 21),
    "vrwkv_post_fwd_bf16": (_c_int, [ctypes.c_long, _c_int, ctypes.c_float] + [_c_void_p] * 10),
    "vrwkv_post_bwd_bf16": (_c_int, [ctypes.c_long, _c_int, ctypes.c_float] + [_c_void_p] * 17),
    "vrwkv_relusq_fwd_bf16": (_c_int, [ctypes.c_long] + [_c_void_p] * 3),
    "vrwkv_relusq_bwd_bf16": (_c_int, [ctypes.c_long] + [_c_void_p] * 4),
    "vrwkv_attention_fwd_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 3 + [ctypes.c_long] * 3 + [_c_void_p] * 2),
    "vrwkv_attention_relpos_fwd_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 3 + [ctypes.c_long] * 3 + [_c_void_p] * 4),
    "vrwkv_attention_set_qtiles": (_c_int, [_c_int]),
    "vrwkv_patch_embed_bf16": (_c_int, [_c_int] * 5 + [_c_void_p] * 5 + [_c_int] * 2 + [_c_void_p]),
    "vrwkv_patch_embed_kp": (_c_int, [_c_int]),
    "vrwkv_adamw_step_bf16": (_c_int, [ctypes.c_long] + [_c_void_p] * 5 + [ctypes.c_float] * 5 + [_c_int, ctypes.c_float, ctypes.c_long, ctypes.c_long, _c_void_p]),
    "vrwkv_adaptive_pool_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 3),
    "vrwkv_gelu_bf16": (_c_int, [_c_long, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "vrwkv_gate_fwd_bf16": (_c_int, [_c_long] + [_c_void_p] * 4),
    "vrwkv_gate_bwd_bf16": (_c_int, [_c_long] + [_c_void_p] * 6),
    "vrwkv_ln_scatter_fwd_bf16": (_c_int, [_c_long, _c_int, _c_float] + [_c_void_p] * 8),
    "vrwkv_ln_gather_bwd_bf16": (_c_int, [_c_long, _c_int] + [_c_void_p] * 10),
    "vrwkv_adamw_step_clip_bf16": (_c_int, [ctypes.c_long] + [_c_void_p] * 5 + [ctypes.c_float] * 5 + [_c_int, _c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_long, ctypes.c_long, _c_void_p]),
    "vrwkv_sqnorm_bf16": (_c_int, [ctypes.c_long, _c_void_p, _c_void_p, _c_void_p]),
    "vrwkv_wkv7_profile_bf16": (_c_int, [_c_int] * 4 + [_c_void_p] * 18),
    "vrwkv_wgrad_skinny_ws_floats": (_c_long, [_c_long, _c_int, _c_int]),
    "vrwkv_wgrad_skinny_bf16": (_c_int, [_c_long, _c_int, _c_int] + [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 2),
    "vrwkv_wgrad_big_ws_floats": (_c_long, [_c_long, _c_int, _c_int]),
    "vrwkv_wgrad_big_bf16": (_c_int, [_c_long, _c_int, _c_int] + [_c_void_p] * 5),
    "vrwkv_stream_copy":(_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p]),
    "vrwkv_stream_probe": (_c_int, [_c_int] + [_c_void_p] * 4 + [_c_long, _c_void_p]),
    "vrwkv_transpose_bf16": (_c_int, [_c_long, _c_long, _c_void_p, _c_void_p, _c_void_p]),
    "vrwkv_resize_normalize_u8": (_c_int, [_c_int, _c_int, _c_void_p, _c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                           _c_void_p, _c_int, _c_void_p]),
    "vrwkv_debug_probe": (_c_int, [_c_int] + [_c_void_p] * 4),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VRWKV_HIP_LIB", LIB_PATH)      # benchmarking aid: a library built with other compiler flags
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} is missing. Build it with `python -m visualrwkv_amd.build` "
            "(hipcc --offload-arch=gfx950); the GPU operators have no PyTorch fallback (CPU tensors go through the op's CPU key: "
            "libvisualrwkv_host.so, hip_lib.load_host()).")
    if "VRWKV_HIP_LIB" not in os.environ:
        # A library older than its sources computes what the sources USED to say (round 6: a GPU run against the previous build of a kernel that had just
        # been changed).  The content hash recorded beside the library travels with it; on a mismatch rebuild (incremental, locked) or refuse.
        from . import build as _build
        if os.path.exists(_build.HASH_PATH) and _build._stale():      # no recorded hash (the file did not travel): mtimes of a copied tree say nothing, load as is
            import warnings
            try:
                _build.hipcc()
            except RuntimeError as e:
                raise HipLibraryError(f"{path} was built from other sources than the ones in {_build.CSRC} and there is no hipcc to rebuild it: "
                                      "run `python -m visualrwkv_amd.build` where the toolchain is") from e
            warnings.warn(f"{path} is older than its sources: rebuilding it (python -m visualrwkv_amd.build)", RuntimeWarning, stacklevel=2)
            _build.build()
    lib = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


_host_lib = None


def load_host() -> ctypes.CDLL:
    """The host-core WKV7 operator alone (libvisualrwkv_host.so, plain C++: `python -m visualrwkv_amd.build` or
    build.build_host()): what the `CPU` dispatch key of torch.ops.wind_backstepping calls.  Loading it needs no ROCm runtime.
    Falls back to the same two entry points inside libvisualrwkv_hip.so when only that library exists."""
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    path = os.environ.get("VRWKV_HOST_LIB")
    if path is None:
        try:
            from .build import build_host
            path = build_host()                              # no-op when the library is newer than its source; rebuilds a stale one
        except Exception as e:      # noqa: BLE001
            if os.path.exists(HOST_LIB_PATH):                # no host compiler here (e.g. a deployment box): use what travelled with the tree
                import warnings
                warnings.warn(f"libvisualrwkv_host.so could not be rebuilt ({type(e).__name__}: {e}); using the existing library, which may "
                              "not match csrc/wkv7_host.hip", RuntimeWarning)
                path = HOST_LIB_PATH
            else:                                            # the HIP library carries the same code
                _host_lib = load()
                return _host_lib
    lib = ctypes.CDLL(path)
    for name in ("vrwkv_wkv7_forward_host", "vrwkv_wkv7_backward_host"):
        res, args = PROTOTYPES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _host_lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().vrwkv_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def launch_stream(device) -> int:
    """Raw handle of PyTorch's current stream on `device` for a ctypes launch.  The kernels are launched on that stream,
    which belongs to `device`: HIP requires it to be the calling thread's current device (one process per GPU always is).
    Anything else is a caller error and is reported instead of launching on the wrong device (wkv7.py / wkv6.py wrap their
    launches in `torch.cuda.device(...)` because the reference's op surface accepts tensors of any device)."""
    import torch
    dev = torch.device(device)
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"visualrwkv_amd: tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"wrap the call in `with torch.cuda.device({dev.index}):`")
    return torch.cuda.current_stream(dev).cuda_stream
