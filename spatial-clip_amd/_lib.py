"""ctypes binding of libspatialclip_hip.so (the C ABI declared in include/spatial_clip_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers + explicit sizes and the
HIP stream as a ``void*``.  Loading fails loudly -- there is no fallback path."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspatialclip_hip.so")
_lib = None


class SpatialClipHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpatialClipHipError(
                f"{LIB_PATH} is missing: build it with `python spatial-clip_amd/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sc_last_error.restype = c_char_p
        _declare(_lib)
    return _lib


I, F, P, LL = c_int, c_float, c_void_p, c_longlong

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPES
SIGNATURES = {
    "sc_abi_version": [],
    "sc_gemm_bf16": [I, I, P, I, P, I, I, I, I, P, I, P, I, P, P, I, P, I, I, P, P],
    "sc_gemm_slab_floats": [I, I, I, I],
}
_RESTYPES = {"sc_gemm_slab_floats": c_longlong}


def _declare(l: ctypes.CDLL) -> None:
    for name, args in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_int)


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SpatialClipHipError(f"{what} failed (rc={rc}): {lib().sc_last_error().decode()}")
