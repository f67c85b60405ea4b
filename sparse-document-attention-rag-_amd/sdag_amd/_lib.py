"""ctypes binding of libsdag_hip.so (declarations: include/sdag_hip.h).

The library is built in-tree by ``csrc/Makefile`` (or ``__graft_entry__.build()``).  There is no CPU
fallback: if the shared object is missing or cannot be loaded, importing the symbols raises.
torch must be imported before the library is loaded so that both share one HIP runtime.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDAG_HIP_LIB") or os.path.join(_HERE, "libsdag_hip.so")  # override: diagnostic builds
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))

SDAG_OK = 0
SDAG_ERR_ARG = -1
SDAG_ERR_CAPACITY = -2
SDAG_ERR_HIP = -3
SDAG_ERR_DESC = -4
OUT_BF16 = 0
OUT_F32 = 1

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
vp = ctypes.c_void_p
i32 = ctypes.c_int
i64 = ctypes.c_int64

# name -> (restype, argtypes); mirrors include/sdag_hip.h one to one
SIGNATURES = {
    "sdag_version": (i32, []),
    "sdag_last_error": (ctypes.c_char_p, []),
    "sdag_desc_build": (i32, [i32, i32, i32, vp, i32, vp, vp, i32, i32, vp, i64, c_i64p]),
    "sdag_desc_from_dense": (i32, [vp, i32, i32, i32, vp, i64, c_i64p]),
    "sdag_desc_stats": (i32, [vp, c_i64p, c_i32p, c_i32p, c_i32p, c_i32p]),
    "sdag_mask_materialize": (i32, [vp, vp, vp, vp]),
    "sdag_attn_workspace_bytes": (i64, [vp, i32, i32, i32]),
    "sdag_attn_fwd": (i32, [vp, vp, vp, vp] + [i64] * 12 + [i32] * 5 + [ctypes.c_float, vp, vp, vp, i64, i32, vp]),
    "sdag_attn_fwd_static": (i32, [vp, vp, vp, vp] + [i64] * 12 + [i32] * 5 + [ctypes.c_float, vp, vp, i32, vp]),
    "sdag_batch_build": (i32, [vp, i32, i32, vp, i64, c_i64p]),
    "sdag_attn_batch_workspace_bytes": (i64, [vp, i32, i32]),
    "sdag_attn_fwd_batch": (i32, [vp, vp, vp, vp] + [i64] * 12 + [i32] * 5 + [ctypes.c_float, vp, vp, vp, i64, i32, vp]),
    "sdag_decode_workspace_bytes": (i64, [i32, i32, i32]),
    "sdag_decode_attn": (i32, [vp, vp, vp, vp] + [i64] * 10 + [i32] * 5 + [ctypes.c_float, vp, i64, i32, vp]),
    "sdag_decode_attn_padded": (i32, [vp, vp, vp, vp] + [i64] * 10 + [i32] * 5 + [ctypes.c_float, vp, i32, vp, i64, i32, vp]),
    "sdag_decode_attn_prealloc": (i32, [vp, vp, vp, vp] + [i64] * 10 + [i32] * 5 + [ctypes.c_float, vp, vp, i32, vp, i64, i32, vp]),
    "sdag_dense_workspace_bytes": (i64, [i32, i64, i32, i32]),
    "sdag_dense_topk": (i32, [vp, i32, vp, i64, i32, i32, vp, vp, i64, vp, i64, vp]),
    "sdag_dense_topk_after": (i32, [vp, i32, vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, i64, vp]),
    "sdag_topk_merge": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "sdag_corpus_to_bf16": (i32, [vp, i64, i32, vp, vp, vp]),
    "sdag_dense_topk2_workspace_bytes": (i64, [i32, i64, i32, i32]),
    "sdag_dense_topk2_flag_offset": (i64, [i32, i64, i32, i32]),
    "sdag_dense_topk2": (i32, [vp, i32, vp, vp, vp, i64, i32, i32, vp, vp, i64, vp, i64, vp]),
    "sdag_gram": (i32, [vp, i32, i32, vp, vp]),
    "sdag_centroid_sims": (i32, [vp, i32, vp, i32, i32, vp, vp, vp]),
    "sdag_centroid_sims_batch": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
}

_lib = None


class SdagError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsdag_hip error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the library for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (there is no CPU fallback for this path)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != SDAG_OK:
        raise SdagError(rc, lib().sdag_last_error().decode(errors="replace"))


def stream_ptr(device=None) -> int:
    """Raw hipStream_t of torch's current stream on ``device``."""
    return int(torch.cuda.current_stream(device).cuda_stream)
