"""Thin, validated wrappers over the C ABI.  Tensors in / tensors out, raw pointers underneath.
Every function enqueues on the current HIP stream and never synchronises."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check

NT, TN = 0, 1
EPI_BF16, EPI_BF16_BIAS, EPI_F32_BIAS_RES, EPI_GELU_PAIR, EPI_BF16_DGELU, EPI_F32 = range(6)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise _lib.SpatialClipHipError(f"{name}: expected a device tensor (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


_slab_cache = {}


def _slabs(nfloat: int, device) -> torch.Tensor:
    key = (device.index,)
    buf = _slab_cache.get(key)
    if buf is None or buf.numel() < nfloat:
        buf = torch.empty(max(nfloat, 1 << 22), dtype=torch.float32, device=device)
        _slab_cache[key] = buf
    return buf


def gemm(mode: int, epi: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int,
         out2: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, splitk: int = 1) -> torch.Tensor:
    """C[M,N] = A.B^T (NT: a[M,K], b[N,K]) or At^T.Bt (TN: a[K,M], b[K,N]) with a fused epilogue."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
    want = torch.float32 if epi in (EPI_F32, EPI_F32_BIAS_RES) else torch.bfloat16
    _req(out, want, "out")
    if bias is not None: _req(bias, torch.float32, "bias")
    if res is not None: _req(res, torch.float32, "res")
    if aux is not None: _req(aux, torch.bfloat16, "aux")
    if out2 is not None: _req(out2, torch.bfloat16, "out2")
    l = _lib.lib()
    slabs = None
    if splitk > 1:
        n = l.sc_gemm_slab_floats(M, N, K, splitk)
        slabs = _slabs(n, out.device) if n else None
        if slabs is None:
            splitk = 1
    rc = l.sc_gemm_bf16(mode, epi, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), M, N, K,
                        out.data_ptr(), out.stride(0), _ptr(out2), out2.stride(0) if out2 is not None else 0,
                        _ptr(bias), _ptr(res), res.stride(0) if res is not None else 0,
                        _ptr(aux), aux.stride(0) if aux is not None else 0, splitk, _ptr(slabs), _stream())
    check(rc, "sc_gemm_bf16")
    return out
