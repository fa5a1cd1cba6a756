"""ctypes binding of libspatialclip_hip.so, generated from the public header include/spatial_clip_hip.h.

No torch types cross the boundary: tensors are passed as raw device pointers + explicit sizes and the
HIP stream as a ``void*``.  Loading fails loudly -- there is no CPU / eager fallback path."""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_longlong, c_void_p
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# SC_HIP_LIB points at an alternative build of the same ABI (A/B benchmarking of kernel variants on one GPU box)
LIB_PATH = os.environ.get("SC_HIP_LIB") or os.path.join(_HERE, "lib", "libspatialclip_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "spatial_clip_hip.h")
_lib = None


class SpatialClipHipError(RuntimeError):
    pass


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return c_void_p
    if decl.startswith("long long"):
        return c_longlong
    if decl.startswith("float"):
        return c_float
    if decl.startswith("int"):
        return c_int
    raise ValueError(f"unsupported C type in header: {decl!r}")


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    """{name: (restype, [argtypes])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"enum\s*\{.*?\};", " ", src, flags=re.S)
    out: Dict[str, Tuple[object, List[object]]] = {}
    for m in re.finditer(r"(const char\*|long long|int)\s+(sc_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = {"const char*": c_char_p, "long long": c_longlong, "int": c_int}[ret]
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        out[name] = (restype, argtypes)
    return out


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- load torch's HIP runtime first so that the kernels share it (one libamdhip64)
        if not os.path.exists(LIB_PATH):
            raise SpatialClipHipError(
                f"{LIB_PATH} is missing: build it with `python spatial-clip_amd/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(l, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SpatialClipHipError(f"{what} failed (rc={rc}): {lib().sc_last_error().decode()}")
