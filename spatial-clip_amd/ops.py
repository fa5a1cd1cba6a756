"""Thin, validated wrappers over the C ABI.  Tensors in / tensors out, raw pointers underneath.
Every function enqueues on the current HIP stream and never synchronises."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import check

NT, TN = 0, 1
(EPI_BF16, EPI_BF16_BIAS, EPI_F32_BIAS_RES, EPI_GELU_PAIR, EPI_BF16_DGELU, EPI_F32, EPI_BF16_BIAS_RES, EPI_GELU_GRAD_PAIR,
 EPI_BF16_MUL_AUX, EPI_QGELU_PAIR, EPI_BF16_DQGELU, EPI_QGELU_GRAD_PAIR) = range(12)


def act_epilogues(quick_gelu: bool):
    """(pair, grad_pair, dgelu) epilogues of a block's activation: exact-erf GELU (nn.GELU) or, for towers built with
    ``quick_gelu: true`` (src/open_clip/model.py:142-145), QuickGELU x * sigmoid(1.702 x)."""
    if quick_gelu:
        return EPI_QGELU_PAIR, EPI_QGELU_GRAD_PAIR, EPI_BF16_DQGELU
    return EPI_GELU_PAIR, EPI_GELU_GRAD_PAIR, EPI_BF16_DGELU


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
    # scratch is per STREAM: two streams of one step (the towers side by side, net.py; the weight-gradient side stream) must
    # never share a split-K slab buffer -- launches of one stream are ordered, launches of two are not
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
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
    if res is not None: _req(res, torch.bfloat16 if epi == EPI_BF16_BIAS_RES else torch.float32, "res")
    if aux is not None: _req(aux, torch.bfloat16, "aux")
    if out2 is not None: _req(out2, torch.bfloat16, "out2")
    l = _lib.lib()
    slabs = None
    if splitk > 1:
        n = l.sc_gemm_slab_floats(M, N, K, splitk)
        slabs = _slabs(n, out.device) if n else None
        if slabs is None:
            splitk = 1
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = l.sc_gemm_bf16(mode, epi, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), M, N, K,
                        out.data_ptr(), out.stride(0), _ptr(out2), out2.stride(0) if out2 is not None else 0,
                        _ptr(bias), _ptr(res), res.stride(0) if res is not None else 0,
                        _ptr(aux), aux.stride(0) if aux is not None else 0, splitk, _ptr(slabs), _stream())
    if ev is not None:
        ev[1].record()
        # algorithmic bytes of the launch: both operands once, every output once, every epilogue input once
        ob = 4 if want == torch.float32 else 2
        nbytes = 2.0 * M * K + 2.0 * N * K + float(ob) * M * N
        nbytes += (2.0 * M * N if out2 is not None else 0.0) + (float(res.element_size()) * M * N if res is not None else 0.0)
        nbytes += (2.0 * M * N if aux is not None else 0.0) + (4.0 * N if bias is not None else 0.0)
        KERNEL_EVENTS.append(("gemm_nt" if mode == NT else "gemm_tn", 2.0 * M * N * K, ev, nbytes))
    check(rc, "sc_gemm_bf16")
    return out


def gemm_wgrad_bias(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, dbias: torch.Tensor, *, M: int, N: int, K: int,
                    splitk: int = 1) -> None:
    """dW[M,N] = dY[K,M]^T . X[K,N] (fp32) and dbias[M] = column sums of dY, one fused pass."""
    _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw")
    _req(dbias, torch.float32, "dbias")
    l = _lib.lib()
    ws = workspace(l.sc_gemm_wgrad_ws_floats(M, N, K, splitk), dw.device, "wgrad")
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = l.sc_gemm_wgrad_bias(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K, dw.data_ptr(),
                              dw.stride(0), dbias.data_ptr(), splitk, ws.data_ptr(), _stream())
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append(("gemm_tn", 2.0 * M * N * K, ev))
    check(rc, "sc_gemm_wgrad_bias")


class _WgradDesc(ctypes.Structure):
    """``sc_wgrad_desc`` of include/spatial_clip_hip.h."""
    _fields_ = [("dY", ctypes.c_void_p), ("lddy", ctypes.c_longlong), ("X", ctypes.c_void_p), ("ldx", ctypes.c_longlong),
                ("dW", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("M", ctypes.c_int), ("N", ctypes.c_int)]


WGRAD_GROUP_MAX = 4


def gemm_wgrad_group(problems, *, K: int, splitk: int = 1) -> None:
    """Several weight (+ bias) gradients over ONE token axis in one launch: ``problems`` = iterable of
    (dy [K, M], x [K, N], dw [M, N] fp32 dense, dbias [M] or None, M, N); dW = dY^T . X, dbias = column sums of dY."""
    problems = list(problems)
    if not 1 <= len(problems) <= WGRAD_GROUP_MAX:
        raise ValueError(f"gemm_wgrad_group: 1..{WGRAD_GROUP_MAX} problems, got {len(problems)}")
    arr = (_WgradDesc * len(problems))()
    flops = 0.0
    for d, (dy, x, dw, dbias, M, N) in zip(arr, problems):
        _req(dy, torch.bfloat16, "dy"); _req(x, torch.bfloat16, "x"); _req(dw, torch.float32, "dw")
        if dbias is not None: _req(dbias, torch.float32, "dbias")
        if not dw.is_contiguous() or dw.numel() != M * N:
            raise ValueError("gemm_wgrad_group: dw must be a dense [M, N] tensor")
        d.dY, d.lddy, d.X, d.ldx = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
        d.dW, d.dbias, d.M, d.N = dw.data_ptr(), _ptr(dbias), M, N
        flops += 2.0 * M * N * K
    l = _lib.lib()
    p = ctypes.cast(arr, ctypes.c_void_p)
    ws = workspace(l.sc_gemm_wgrad_group_ws_floats(p, len(problems), K, splitk), problems[0][0].device, "wgrad")
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = l.sc_gemm_wgrad_group(p, len(problems), K, splitk, ws.data_ptr(), _stream())
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append(("gemm_tn", flops, ev))
    check(rc, "sc_gemm_wgrad_group")


# bench.py sets this to a list to bracket every GEMM launch with HIP events on the launch stream
KERNEL_EVENTS = None


# ------------------------------------------------------------------------------------------ workspace
_ws_cache = {}


def workspace(nfloat: int, device, tag: str = "ws", dtype=torch.float32) -> torch.Tensor:
    """Scratch of at least ``nfloat`` elements for the launch being enqueued: one buffer per (device, purpose, dtype, STREAM) --
    see ``_slabs``."""
    key = (device.index, tag, dtype, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nfloat:
        buf = torch.empty(max(int(nfloat), 1024), dtype=dtype, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------------------------------ attention
def attn_fwd(qkv: torch.Tensor, B: int, L: int, H: int, dh: int, causal: bool = False,
             out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None, q_rows: int = 0):
    _req(qkv, torch.bfloat16, "qkv")
    if out is None:
        out = torch.empty((B * L, H * dh), dtype=torch.bfloat16, device=qkv.device)
    if lse is None:
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    check(_lib.lib().sc_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, L, H, dh, int(causal), q_rows,
                                 _stream()), "sc_attn_fwd")
    return out, lse


def attn_bwd(qkv, out, dout, lse, B: int, L: int, H: int, dh: int, causal: bool = False,
             dqkv: Optional[torch.Tensor] = None, delta: Optional[torch.Tensor] = None, q_rows: int = 0):
    for t_, n in ((qkv, "qkv"), (out, "out"), (dout, "dout")):
        _req(t_, torch.bfloat16, n)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    check(_lib.lib().sc_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                 dqkv.data_ptr(), B, L, H, dh, int(causal), q_rows, _stream()), "sc_attn_bwd")
    return dqkv


# ------------------------------------------------------------------------------------------ norms
def _t8_args(t8, rows: int, d: int, what: str):
    """(buffer uint8 [rows, >= d], scale fp32 [1], amax fp32 [64]) -> ctypes arguments of a per-tensor e4m3 second output."""
    buf, scale, amax = t8
    if buf.dtype != torch.uint8 or not buf.is_cuda or buf.stride(-1) != 1 or buf.shape[0] < rows or buf.shape[1] < d:
        raise TypeError(f"{what}: t8 buffer must be a device uint8 matrix of at least [{rows}, {d}]")
    _req(scale, torch.float32, "t8 scale"); _req(amax, torch.float32, "t8 amax")
    if scale.numel() != 1 or amax.numel() != 64:
        raise ValueError(f"{what}: t8 needs a one-element scale and 64 amax slots")
    return buf.data_ptr(), buf.stride(0), scale.data_ptr(), amax.data_ptr()


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows: int, d: int, ldx: Optional[int] = None,
                  ldy: Optional[int] = None, eps: float = 1e-5, q8: Optional[torch.Tensor] = None,
                  q8_scale_inv: Optional[torch.Tensor] = None, t8=None):
    """``q8`` (uint8 [rows, d]) + ``q8_scale_inv`` (fp32 [rows]): also emit the e4m3 copy of the output with its
    per-row power-of-two scale (the fp8 forward GEMM's A operand) from the same kernel.  ``t8`` = (uint8 [rows, d], scale
    [1], amax [64]): a SECOND e4m3 copy with one delayed scale for the whole tensor (the e4m3 weight gradient's X operand;
    bf16 rows + q8 only)."""
    _req(y, torch.bfloat16, "y")
    if q8 is not None:
        if q8.dtype != torch.uint8 or not q8.is_cuda or q8.stride(-1) != 1:
            raise TypeError("layernorm_fwd: q8 must be a device uint8 matrix")
        _req(q8_scale_inv, torch.float32, "q8_scale_inv")
    if t8 is not None and (x.dtype != torch.bfloat16 or q8 is None):
        raise _lib.SpatialClipHipError("layernorm_fwd: the per-tensor e4m3 copy exists for bf16 rows with the per-row copy (q8)")
    if x.dtype == torch.bfloat16 and t8 is not None:
        tp, tld, tsc, tam = _t8_args(t8, rows, d, "layernorm_fwd")
        check(_lib.lib().sc_layernorm_fwd_x16_t8(x.data_ptr(), ldx if ldx is not None else d, gamma.data_ptr(), beta.data_ptr(),
                                                 y.data_ptr(), ldy if ldy is not None else d, q8.data_ptr(), q8.stride(0),
                                                 q8_scale_inv.data_ptr(), tp, tld, tsc, tam, _ptr(mean), _ptr(rstd), rows, d, eps,
                                                 _stream()), "sc_layernorm_fwd_x16_t8")
        return y
    if x.dtype == torch.bfloat16:           # residual stream kept in bf16 (EPI_BF16_BIAS_RES)
        _req(x, torch.bfloat16, "x")
        check(_lib.lib().sc_layernorm_fwd_x16(x.data_ptr(), ldx if ldx is not None else d, gamma.data_ptr(), beta.data_ptr(),
                                              y.data_ptr(), ldy if ldy is not None else d, _ptr(q8),
                                              q8.stride(0) if q8 is not None else 0, _ptr(q8_scale_inv), _ptr(mean),
                                              _ptr(rstd), rows, d, eps, _stream()), "sc_layernorm_fwd_x16")
        return y
    _req(x, torch.float32, "x")
    if q8 is not None:
        check(_lib.lib().sc_layernorm_fwd_q8(x.data_ptr(), ldx if ldx is not None else d, gamma.data_ptr(),
                                             beta.data_ptr(), y.data_ptr(), ldy if ldy is not None else d, q8.data_ptr(),
                                             q8.stride(0), q8_scale_inv.data_ptr(), _ptr(mean), _ptr(rstd), rows, d, eps,
                                             _stream()), "sc_layernorm_fwd_q8")
        return y
    check(_lib.lib().sc_layernorm_fwd(x.data_ptr(), ldx if ldx is not None else d, gamma.data_ptr(), beta.data_ptr(),
                                      y.data_ptr(), ldy if ldy is not None else d, _ptr(mean), _ptr(rstd), rows, d, eps,
                                      _stream()), "sc_layernorm_fwd")
    return y


def gelu_bf16(u: torch.Tensor, h: torch.Tensor, quick: bool = False) -> torch.Tensor:
    """h = gelu(u) (bf16 -> bf16, contiguous), bit-identical to the GELU-pair GEMM epilogue's second output; ``quick``:
    QuickGELU (the SC_EPI_QGELU_PAIR epilogue's)."""
    _req(u, torch.bfloat16, "u"); _req(h, torch.bfloat16, "h")
    if not (u.is_contiguous() and h.is_contiguous()) or u.numel() != h.numel():
        raise ValueError("gelu_bf16: contiguous tensors of equal size required")
    fn = _lib.lib().sc_quick_gelu_bf16 if quick else _lib.lib().sc_gelu_bf16
    check(fn(u.data_ptr(), h.data_ptr(), u.numel(), _stream()), "sc_gelu_bf16")
    return h


def layernorm_bwd(dy, x, mean, rstd, gamma, dres, dres_bf16, dgamma, dbeta, colsum, rows: int, d: int, *,
                  accumulate, lddy=None, ldx=None, lddres=None, lddbf=None, ws=None, defer_reduce: bool = False,
                  q8: Optional[torch.Tensor] = None, q8_scale_inv: Optional[torch.Tensor] = None,
                  g_in: Optional[torch.Tensor] = None, ldgin=None, write_f32: bool = True, g16: bool = False, t8=None):
    """``accumulate``: False / True, or a negative int -P: only rows r % P == 0 of ``dres`` carry an incoming gradient.
    ``defer_reduce``: leave the per-block partial sums of dgamma / dbeta / colsum in ``ws`` (caller-owned, at least
    layernorm_bwd_ws_floats floats) and finish them with layernorm_bwd_reduce -- e.g. on the weight-gradient stream.
    ``g16`` (sc_layernorm_bwd_g16): the residual gradient travels in bf16 -- the incoming one is ``g_in`` (bf16) when
    ``accumulate`` is True, the outgoing one ``dres_bf16``; the fp32 ``dres`` is written only if ``write_f32``."""
    _req(dy, torch.bfloat16, "dy"); _req(dres, torch.float32, "dres")
    xb = x.dtype == torch.bfloat16          # residual stream kept in bf16
    _req(x, torch.bfloat16 if xb else torch.float32, "x")
    l = _lib.lib()
    if ws is None:
        if defer_reduce:
            raise _lib.SpatialClipHipError("layernorm_bwd: defer_reduce needs a caller-owned workspace")
        ws = workspace(l.sc_layernorm_bwd_ws_floats(rows, d), dy.device, "ln")
    legacy = xb and not g16                 # bf16 rows, fp32 gradient buffer read-modify-written as in sc_layernorm_bwd
    if legacy:
        g16, g_in, write_f32 = True, None, True
    if g16:
        if int(accumulate) > 0 and not legacy:
            _req(g_in, torch.bfloat16, "g_in")
        if dres_bf16 is not None or not legacy:
            _req(dres_bf16, torch.bfloat16, "dres_bf16")
        if q8 is not None:
            _req(q8_scale_inv, torch.float32, "q8_scale_inv")
        if t8 is not None:
            if not xb or q8 is None:
                raise _lib.SpatialClipHipError("layernorm_bwd: the per-tensor e4m3 copy exists for bf16 rows with the per-row copy (q8)")
            tp, tld, tsc, tam = _t8_args(t8, rows, d, "layernorm_bwd")
            check(l.sc_layernorm_bwd_x16_t8(dy.data_ptr(), lddy or d, x.data_ptr(), ldx or d, mean.data_ptr(), rstd.data_ptr(),
                                            gamma.data_ptr(), _ptr(g_in), ldgin or d, dres.data_ptr(), lddres or d, int(write_f32),
                                            _ptr(dres_bf16), lddbf or d, q8.data_ptr(), q8.stride(0), q8_scale_inv.data_ptr(),
                                            tp, tld, tsc, tam, int(accumulate), None if defer_reduce else dgamma.data_ptr(),
                                            dbeta.data_ptr(), _ptr(colsum), ws.data_ptr(), rows, d, _stream()),
                  "sc_layernorm_bwd_x16_t8")
            return
        fn = l.sc_layernorm_bwd_x16 if xb else l.sc_layernorm_bwd_g16
        check(fn(dy.data_ptr(), lddy or d, x.data_ptr(), ldx or d, mean.data_ptr(), rstd.data_ptr(),
                                     gamma.data_ptr(), _ptr(g_in), ldgin or d, dres.data_ptr(), lddres or d, int(write_f32),
                                     _ptr(dres_bf16), lddbf or d, _ptr(q8), q8.stride(0) if q8 is not None else 0,
                                     _ptr(q8_scale_inv), int(accumulate),
                                     None if defer_reduce else dgamma.data_ptr(), dbeta.data_ptr(), _ptr(colsum),
                                     ws.data_ptr(), rows, d, _stream()), "sc_layernorm_bwd_x16" if xb else "sc_layernorm_bwd_g16")
        return
    if q8 is not None:          # e4m3 copy of the new residual gradient + per-row 1/scale (A operand of the fp8 dgrad GEMMs)
        if q8.dtype != torch.uint8 or not q8.is_cuda or q8.stride(-1) != 1:
            raise TypeError("layernorm_bwd: q8 must be a device uint8 matrix")
        _req(q8_scale_inv, torch.float32, "q8_scale_inv")
        check(l.sc_layernorm_bwd_q8(dy.data_ptr(), lddy or d, x.data_ptr(), ldx or d, mean.data_ptr(), rstd.data_ptr(),
                                    gamma.data_ptr(), dres.data_ptr(), lddres or d, _ptr(dres_bf16), lddbf or d,
                                    q8.data_ptr(), q8.stride(0), q8_scale_inv.data_ptr(), int(accumulate),
                                    None if defer_reduce else dgamma.data_ptr(), dbeta.data_ptr(), _ptr(colsum),
                                    ws.data_ptr(), rows, d, _stream()), "sc_layernorm_bwd_q8")
        return
    check(l.sc_layernorm_bwd(dy.data_ptr(), lddy or d, x.data_ptr(), ldx or d, mean.data_ptr(), rstd.data_ptr(),
                             gamma.data_ptr(), dres.data_ptr(), lddres or d, _ptr(dres_bf16), lddbf or d,
                             int(accumulate), None if defer_reduce else dgamma.data_ptr(), dbeta.data_ptr(), _ptr(colsum),
                             ws.data_ptr(), rows, d, _stream()), "sc_layernorm_bwd")


def layernorm_bwd_ws_floats(rows: int, d: int) -> int:
    return int(_lib.lib().sc_layernorm_bwd_ws_floats(rows, d))


def layernorm_bwd_reduce(ws, dgamma, dbeta, colsum, rows: int, d: int):
    _req(ws, torch.float32, "ws")
    check(_lib.lib().sc_layernorm_bwd_reduce(ws.data_ptr(), rows, d, dgamma.data_ptr(), dbeta.data_ptr(), _ptr(colsum),
                                             _stream()), "sc_layernorm_bwd_reduce")


def bias_gelu_pair(x, bias, u, h, rows: int, n: int):
    _req(x, torch.float32, "x"); _req(u, torch.bfloat16, "u"); _req(h, torch.bfloat16, "h")
    check(_lib.lib().sc_bias_gelu_pair(x.data_ptr(), bias.data_ptr(), u.data_ptr(), h.data_ptr(), rows, n, _stream()),
          "sc_bias_gelu_pair")


def colsum_bf16(x, rows: int, n: int, out, ld: Optional[int] = None):
    _req(x, torch.bfloat16, "x"); _req(out, torch.float32, "out")
    l = _lib.lib()
    ws = workspace(l.sc_colsum_ws_floats(rows, n), x.device, "colsum")
    check(l.sc_colsum_bf16(x.data_ptr(), ld or x.stride(0), rows, n, out.data_ptr(), ws.data_ptr(), _stream()),
          "sc_colsum_bf16")
    return out


def l2norm_fwd(x, y, y_bf16, inv, rows: int, d: int):
    check(_lib.lib().sc_l2norm_fwd(x.data_ptr(), y.data_ptr(), _ptr(y_bf16), _ptr(inv), rows, d, _stream()),
          "sc_l2norm_fwd")
    return y


def l2norm_bwd(dy, y, inv, dx_bf16, rows: int, d: int):
    check(_lib.lib().sc_l2norm_bwd(dy.data_ptr(), y.data_ptr(), inv.data_ptr(), dx_bf16.data_ptr(), rows, d, _stream()),
          "sc_l2norm_bwd")
    return dx_bf16


def cast_pad_bf16(src, dst, rows: int, cols: int, cols_pad: int, ld_src=None, ld_dst=None):
    _req(src, torch.float32, "src"); _req(dst, torch.bfloat16, "dst")
    check(_lib.lib().sc_cast_pad_bf16(src.data_ptr(), ld_src or cols, dst.data_ptr(), ld_dst or cols_pad, rows, cols,
                                      cols_pad, _stream()), "sc_cast_pad_bf16")
    return dst


def cast_transpose_bf16(src, dst, rows: int, cols: int, ld_dst=None):
    _req(src, torch.float32, "src"); _req(dst, torch.bfloat16, "dst")
    check(_lib.lib().sc_cast_transpose_bf16(src.data_ptr(), dst.data_ptr(), rows, cols, ld_dst or rows, _stream()),
          "sc_cast_transpose_bf16")
    return dst


# ------------------------------------------------------------------------------------------ patch embedding
def im2col(images, patches, P: int, ld_out: Optional[int] = None):
    _req(images, torch.float32, "images"); _req(patches, torch.bfloat16, "patches")
    B, C, H, W = images.shape
    if not images.is_contiguous():
        raise ValueError("images must be contiguous NCHW")
    check(_lib.lib().sc_im2col(images.data_ptr(), patches.data_ptr(), B, C, H, W, P, ld_out or patches.stride(0),
                               _stream()), "sc_im2col")
    return patches


def embed_ln_fwd(patch_out, cls, pos, gamma, beta, x, mean, rstd, B: int, L: int, d: int, eps: float = 1e-5):
    """class token + positional embedding + ln_pre -> the residual stream ``x`` [B*L, d]: fp32, or bf16 (sc_embed_ln_fwd_x16)
    when the stream is kept in bf16."""
    if x.dtype == torch.bfloat16:
        _req(x, torch.bfloat16, "x")
        fn, what = _lib.lib().sc_embed_ln_fwd_x16, "sc_embed_ln_fwd_x16"
    else:
        _req(x, torch.float32, "x")
        fn, what = _lib.lib().sc_embed_ln_fwd, "sc_embed_ln_fwd"
    check(fn(patch_out.data_ptr(), cls.data_ptr(), pos.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.data_ptr(),
             mean.data_ptr(), rstd.data_ptr(), B, L, d, float(eps), _stream()), what)
    return x


def embed_ln_bwd(dres, patch_out, cls, pos, mean, rstd, gamma, dpatch_bf16, dgamma, dbeta, dpos, dcls, B, L, d):
    l = _lib.lib()
    ws = workspace(l.sc_embed_ln_bwd_ws_floats(B, L, d), dres.device, "embed")
    check(l.sc_embed_ln_bwd(dres.data_ptr(), patch_out.data_ptr(), cls.data_ptr(), pos.data_ptr(), mean.data_ptr(),
                            rstd.data_ptr(), gamma.data_ptr(), dpatch_bf16.data_ptr(), dgamma.data_ptr(),
                            dbeta.data_ptr(), dpos.data_ptr(), dcls.data_ptr(), ws.data_ptr(), B, L, d, _stream()),
          "sc_embed_ln_bwd")


# ------------------------------------------------------------------------------------------ contrastive head
def sgemm(a, sam, sak, b, sbn, sbk, c, ldc, M, N, K, accumulate=False):
    check(_lib.lib().sc_sgemm_f32(a.data_ptr(), sam, sak, b.data_ptr(), sbn, sbk, c.data_ptr(), ldc, M, N, K,
                                  int(accumulate), _stream()), "sc_sgemm_f32")
    return c


class _SgemmDesc(ctypes.Structure):
    """``sc_sgemm_desc`` of include/spatial_clip_hip.h (host-side problem descriptor)."""
    _fields_ = [("A", ctypes.c_void_p), ("sam", ctypes.c_longlong), ("sak", ctypes.c_longlong),
                ("B", ctypes.c_void_p), ("sbn", ctypes.c_longlong), ("sbk", ctypes.c_longlong),
                ("C", ctypes.c_void_p), ("ldc", ctypes.c_longlong),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("accumulate", ctypes.c_int)]


SGEMM_MAX_GROUP = 6


def sgemm_grouped(problems) -> None:
    """Several independent fp32 GEMMs in one launch.  ``problems``: iterable of
    (a, sam, sak, b, sbn, sbk, c, ldc, M, N, K[, accumulate]) exactly as for :func:`sgemm`."""
    problems = list(problems)
    if not 1 <= len(problems) <= SGEMM_MAX_GROUP:
        raise ValueError(f"sgemm_grouped: 1..{SGEMM_MAX_GROUP} problems, got {len(problems)}")
    arr = (_SgemmDesc * len(problems))()
    for d, p in zip(arr, problems):
        a, sam, sak, b, sbn, sbk, c, ldc, M, N, K = p[:11]
        for t_, n in ((a, "a"), (b, "b"), (c, "c")):
            if not t_.is_cuda or t_.dtype != torch.float32:
                raise TypeError(f"sgemm_grouped: {n} must be a device fp32 tensor")
        d.A, d.sam, d.sak = a.data_ptr(), sam, sak
        d.B, d.sbn, d.sbk = b.data_ptr(), sbn, sbk
        d.C, d.ldc = c.data_ptr(), ldc
        d.M, d.N, d.K, d.accumulate = M, N, K, int(p[11]) if len(p) > 11 else 0
    check(_lib.lib().sc_sgemm_f32_grouped(ctypes.cast(arr, ctypes.c_void_p), len(problems), _stream()),
          "sc_sgemm_f32_grouped")


def pack_rows(feat: torch.Tensor, ids_a: Optional[torch.Tensor], ids_b: Optional[torch.Tensor],
              out: torch.Tensor) -> torch.Tensor:
    """out[B, D(+4)] = feat | ids_a | ids_b (int64 through two float slots each): all-gather send buffer."""
    _req(feat, torch.float32, "feat"); _req(out, torch.float32, "out")
    if ids_a is not None:
        _req(ids_a, torch.int64, "ids_a"); _req(ids_b, torch.int64, "ids_b")
        if not (ids_a.is_contiguous() and ids_b.is_contiguous()):
            raise ValueError("pack_rows: id vectors must be contiguous")
    B, D = feat.shape
    check(_lib.lib().sc_pack_rows(feat.data_ptr(), feat.stride(0), _ptr(ids_a), _ptr(ids_b), out.data_ptr(),
                                  out.stride(0), B, D, _stream()), "sc_pack_rows")
    return out


def neighbor_join(all_img_ids, all_txt_ids, nbr_ids, nbr_alpha, B, G, K, rank, alpha_scale, lab_col, lab_w):
    for t_, n in ((all_img_ids, "all_image_tile_ids"), (all_txt_ids, "all_text_tile_ids"), (nbr_ids, "neighbor_tile_ids")):
        _req(t_, torch.int64, n)
    _req(nbr_alpha, torch.float32, "neighbor_alphas")
    check(_lib.lib().sc_neighbor_join(all_img_ids.data_ptr(), all_txt_ids.data_ptr(), nbr_ids.data_ptr(),
                                      nbr_alpha.data_ptr(), B, G, K, rank, float(alpha_scale), lab_col.data_ptr(),
                                      lab_w.data_ptr(), _stream()), "sc_neighbor_join")


def onehot_labels(B, rank, lab_col, lab_w):
    check(_lib.lib().sc_onehot_labels(B, rank, lab_col.data_ptr(), lab_w.data_ptr(), _stream()), "sc_onehot_labels")


def contrastive_loss_fwd(z, B, G, scale, cap, bias, lab_col, lab_w, nlab, w, rowstats, loss_out):
    check(_lib.lib().sc_contrastive_loss_fwd(z.data_ptr(), B, G, scale.data_ptr(), float(cap), _ptr(bias),
                                             lab_col.data_ptr(), lab_w.data_ptr(), nlab, float(w),
                                             rowstats.data_ptr(), loss_out.data_ptr(), _stream()),
          "sc_contrastive_loss_fwd")


def contrastive_loss_bwd(z, B, G, scale, cap, bias, lab_col, lab_w, nlab, w, rowstats, loss_out, grad_out, rowgrad,
                         dscale, dbias):
    check(_lib.lib().sc_contrastive_loss_bwd(z.data_ptr(), B, G, scale.data_ptr(), float(cap), _ptr(bias),
                                             lab_col.data_ptr(), lab_w.data_ptr(), nlab, float(w),
                                             rowstats.data_ptr(), loss_out.data_ptr(), _ptr(grad_out),
                                             rowgrad.data_ptr(), _ptr(dscale), _ptr(dbias), _stream()),
          "sc_contrastive_loss_bwd")


def recall_hits(z_img_rows, G, B, col0, hits3):
    check(_lib.lib().sc_recall_hits(z_img_rows.data_ptr(), G, B, col0, hits3.data_ptr(), _stream()), "sc_recall_hits")


def exp_scalar(x, y):
    check(_lib.lib().sc_exp_scalar(x.data_ptr(), y.data_ptr(), _stream()), "sc_exp_scalar")
    return y


def scale_by_scalar(x: torch.Tensor, s: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out = x * s (s: one-element device tensor); contiguous fp32."""
    _req(x, torch.float32, "x"); _req(s, torch.float32, "s"); _req(out, torch.float32, "out")
    if not (x.is_contiguous() and out.is_contiguous()) or x.numel() != out.numel() or s.numel() != 1:
        raise ValueError("scale_by_scalar: contiguous tensors of equal size and a one-element scale required")
    check(_lib.lib().sc_scale_by_scalar(x.data_ptr(), s.data_ptr(), out.data_ptr(), x.numel(), _stream()), "sc_scale_by_scalar")
    return out


def exp_scalar_bwd(y, dy, dx, mult: float = 1.0):
    check(_lib.lib().sc_exp_scalar_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), float(mult), _stream()),
          "sc_exp_scalar_bwd")


# ------------------------------------------------------------------------------------------ optimiser
def grad_norm(grads, n, grad_scale, max_norm, out2):
    ws = workspace(1024, grads.device, "gn", torch.float64)
    check(_lib.lib().sc_grad_norm(grads.data_ptr(), n, float(grad_scale), float(max_norm), ws.data_ptr(),
                                  out2.data_ptr(), _stream()), "sc_grad_norm")
    return out2


def grad_sumsq_partial(grads, n, partial1024):
    """1024 fp64 partial sums of squares of grads[:n] (one contiguous piece of a rank's gradient shard)."""
    _req(grads, torch.float32, "grads"); _req(partial1024, torch.float64, "partial")
    check(_lib.lib().sc_grad_sumsq_partial(grads.data_ptr(), int(n), partial1024.data_ptr(), _stream()), "sc_grad_sumsq_partial")


def grad_norm_final(partial, n_partial, grad_scale, max_norm, out2):
    """out2 = {norm * grad_scale, clip coefficient} from (all-reduced) fp64 partial sums of squares."""
    _req(partial, torch.float64, "partial")
    check(_lib.lib().sc_grad_norm_final(partial.data_ptr(), int(n_partial), float(grad_scale), float(max_norm), out2.data_ptr(),
                                        _stream()), "sc_grad_norm_final")
    return out2


def adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, wd, step, grad_scale, norm_clip, p_bf16=None):
    check(_lib.lib().sc_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, float(lr), float(beta1),
                                   float(beta2), float(eps), float(wd), int(step), float(grad_scale), _ptr(norm_clip),
                                   _ptr(p_bf16), _stream()), "sc_adamw_step")


def adamw_step_dev(p, g, m, v, n, hyper, beta1, beta2, eps, wd, grad_scale, norm_clip, p_bf16=None):
    """AdamW with {lr, 1 - beta1^step, sqrt(1 - beta2^step)} read from the device tensor ``hyper`` (graph-captured steps)."""
    _req(hyper, torch.float32, "hyper")
    check(_lib.lib().sc_adamw_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(), float(beta1),
                                       float(beta2), float(eps), float(wd), float(grad_scale), _ptr(norm_clip), _ptr(p_bf16),
                                       _stream()), "sc_adamw_step_dev")


def cast_transpose_batched(master, desc, tile_prefix, n, total_tiles, mirror_bf16=None):
    check(_lib.lib().sc_cast_transpose_batched(master.data_ptr(), _ptr(mirror_bf16), desc.data_ptr(),
                                               tile_prefix.data_ptr(), n, total_tiles, _stream()),
          "sc_cast_transpose_batched")


# ------------------------------------------------------------------------------------------ text tower glue
def token_embed_fwd(tokens, table, pos, x, B, L, d, V):
    _req(tokens, torch.int64, "tokens")
    check(_lib.lib().sc_token_embed_fwd(tokens.data_ptr(), table.data_ptr(), pos.data_ptr(), x.data_ptr(), B, L, d, V,
                                        _stream()), "sc_token_embed_fwd")
    return x


def token_embed_bwd(tokens, dres, dtable, dpos, B, L, d, V, eot=None, deterministic=True):
    """Gradient of the embedding gather + positional embedding.  ``deterministic`` (default): every table row summed in
    position order by one wave per column slab (bit-reproducible); ``eot`` int32 [B] = pooled positions (rows behind them are
    skipped: exactly zero in the causal tower).  ``deterministic=False``: float atomics (order-dependent last bit)."""
    if deterministic:
        if eot is not None:
            _req(eot, torch.int32, "eot")
        check(_lib.lib().sc_token_embed_bwd_det(tokens.data_ptr(), _ptr(eot), dres.data_ptr(), dtable.data_ptr(), dpos.data_ptr(),
                                                B, L, d, V, _stream()), "sc_token_embed_bwd_det")
        return
    check(_lib.lib().sc_token_embed_bwd(tokens.data_ptr(), dres.data_ptr(), dtable.data_ptr(), dpos.data_ptr(), B, L, d,
                                        V, _stream()), "sc_token_embed_bwd")


def argmax_rows(tokens, out_idx, B, L):
    _req(tokens, torch.int64, "tokens")
    check(_lib.lib().sc_argmax_rows_i64(tokens.data_ptr(), out_idx.data_ptr(), B, L, _stream()), "sc_argmax_rows_i64")
    return out_idx


def gather_rows(src, idx, L, dst, B, d):
    check(_lib.lib().sc_gather_rows_f32(src.data_ptr(), idx.data_ptr(), L, dst.data_ptr(), B, d, _stream()),
          "sc_gather_rows_f32")
    return dst


def scatter_rows(src, idx, L, dst, dst_bf16, B, d):
    check(_lib.lib().sc_scatter_rows_f32(src.data_ptr(), idx.data_ptr(), L, dst.data_ptr(), _ptr(dst_bf16), B, d,
                                         _stream()), "sc_scatter_rows_f32")


def pcc_rows(pred: torch.Tensor, target: torch.Tensor, pcc: Optional[torch.Tensor] = None,
             sum_count: Optional[torch.Tensor] = None) -> None:
    """Row-wise Pearson correlation with the reference metric's guards (src/metrics/zero_shot.py:72-88)."""
    _req(pred, torch.float32, "pred"); _req(target, torch.float32, "target")
    rows, cols = pred.shape
    if tuple(target.shape) != (rows, cols):
        raise _lib.SpatialClipHipError(f"pcc_rows: pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
    check(_lib.lib().sc_pcc_rows(pred.data_ptr(), pred.stride(0), target.data_ptr(), target.stride(0), rows, cols,
                                 _ptr(pcc), _ptr(sum_count), _stream()), "sc_pcc_rows")


# ------------------------------------------------------------------------------------------ input pipeline
def knn_alpha(xy: torch.Tensor, K: int, mode: str = "inverse", sigma: float = 1.0):
    """Spatial neighbours + loss weights of the tiles of one slide: xy fp32 [N,2] -> (nbr_index int32 [N,K], alpha [N,K])."""
    _req(xy, torch.float32, "xy")
    if xy.dim() != 2 or xy.shape[1] != 2 or not xy.is_contiguous():
        raise ValueError("knn_alpha: xy must be a contiguous [N, 2] tensor")
    N = xy.shape[0]
    nbr = torch.empty((N, K), dtype=torch.int32, device=xy.device)
    alpha = torch.empty((N, K), dtype=torch.float32, device=xy.device)
    check(_lib.lib().sc_knn_alpha(xy.data_ptr(), N, K, {"inverse": 0, "gaussian": 1}[mode], float(sigma), nbr.data_ptr(),
                                  alpha.data_ptr(), _stream()), "sc_knn_alpha")
    return nbr, alpha


def png_decode(files: torch.Tensor, offsets: torch.Tensor, H: int, W: int):
    """Concatenated PNG files (device uint8 [total]) + int64 offsets [B + 1] -> (uint8 [B, H, W, 3], int32 status [B])."""
    if not files.is_cuda or files.dtype != torch.uint8 or files.dim() != 1 or not files.is_contiguous():
        raise TypeError("png_decode: files must be a contiguous device uint8 vector")
    _req(offsets, torch.int64, "offsets")
    B = offsets.numel() - 1
    if B < 1:
        raise ValueError("png_decode: offsets must hold B + 1 entries")
    l = _lib.lib()
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=files.device)
    status = torch.full((B,), -1, dtype=torch.int32, device=files.device)
    scratch = workspace(l.sc_png_decode_scratch_bytes(B, H, W), files.device, "png", torch.uint8)
    check(l.sc_png_decode(files.data_ptr(), offsets.data_ptr(), B, out.data_ptr(), H, W, scratch.data_ptr(),
                          status.data_ptr(), _stream()), "sc_png_decode")
    return out, status


def augment_tiles(src_u8: torch.Tensor, params: torch.Tensor, out_size: int, mean, std) -> torch.Tensor:
    """uint8 [B,H,W,3] + params fp32 [B,12] -> normalised fp32 [B,3,S,S] (crop, resize, colour jitter, normalise)."""
    if not src_u8.is_cuda or src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3 \
            or not src_u8.is_contiguous():
        raise TypeError("augment_tiles: src must be a contiguous device uint8 [B,H,W,3] tensor")
    _req(params, torch.float32, "params")
    B, H, W, _ = src_u8.shape
    if tuple(params.shape) != (B, 12) or not params.is_contiguous():
        raise ValueError("augment_tiles: params must be [B, 12]")
    out = torch.empty((B, 3, out_size, out_size), dtype=torch.float32, device=src_u8.device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().sc_augment_tiles(src_u8.data_ptr(), B, H, W, params.data_ptr(), out.data_ptr(), out_size,
                                      ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p), _stream()),
          "sc_augment_tiles")
    return out


# ------------------------------------------------------------------------------------------ fp8 forward path
def quantize_rows_fp8_batched(desc: torch.Tensor, block_prefix: torch.Tensor, n: int, total_blocks: int) -> None:
    """Row-wise e4m3 copies of ``n`` matrices in one launch (params.ParamStore._q8_plan builds the device tables)."""
    check(_lib.lib().sc_quantize_rows_fp8_batched(desc.data_ptr(), block_prefix.data_ptr(), n, total_blocks, _stream()),
          "sc_quantize_rows_fp8_batched")


def quantize_rows_fp8(src: torch.Tensor, dst: Optional[torch.Tensor] = None, scale_inv: Optional[torch.Tensor] = None,
                      fixed_scale: float = 0.0):
    """Row-wise e4m3 quantisation with a power-of-two scale per row: -> (fp8 bytes as uint8 [rows, cols], 1/scale [rows])."""
    if not src.is_cuda or src.dtype not in (torch.float32, torch.bfloat16) or src.dim() != 2 or src.stride(1) != 1:
        raise TypeError("quantize_rows_fp8: src must be a device fp32 / bf16 matrix with unit inner stride")
    rows, cols = src.shape
    if dst is None:
        dst = torch.empty((rows, cols), dtype=torch.uint8, device=src.device)
    if scale_inv is None:
        scale_inv = torch.empty(rows, dtype=torch.float32, device=src.device)
    check(_lib.lib().sc_quantize_rows_fp8(src.data_ptr(), int(src.dtype == torch.float32), src.stride(0), rows, cols,
                                          dst.data_ptr(), dst.stride(0), scale_inv.data_ptr(), float(fixed_scale),
                                          _stream()), "sc_quantize_rows_fp8")
    return dst, scale_inv


def fp8_scale_update(amax_slots: torch.Tensor, scale: torch.Tensor, scale_inv: torch.Tensor, margin_bits: int = 1,
                     hist: Optional[torch.Tensor] = None, slot: int = 0) -> None:
    """Delayed scaling: next step's per-tensor e4m3 scales from the maxima the epilogues recorded ([n, 64] slots, cleared).
    ``hist`` (fp32 [H, n]) + ``slot``: keep the maxima of the last H steps and scale for their maximum."""
    _req(amax_slots, torch.float32, "amax_slots"); _req(scale, torch.float32, "scale"); _req(scale_inv, torch.float32, "scale_inv")
    n = scale.numel()
    if amax_slots.numel() != n * 64 or scale_inv.numel() != n:
        raise ValueError("fp8_scale_update: amax_slots must be [n, 64] for n scales")
    if hist is not None:
        _req(hist, torch.float32, "hist")
        if hist.dim() != 2 or hist.shape[1] != n or not hist.is_contiguous():
            raise ValueError("fp8_scale_update: hist must be a contiguous [H, n] tensor")
        check(_lib.lib().sc_fp8_scale_update_hist(amax_slots.data_ptr(), hist.data_ptr(), hist.shape[0], int(slot) % hist.shape[0],
                                                  scale.data_ptr(), scale_inv.data_ptr(), n, int(margin_bits), _stream()),
              "sc_fp8_scale_update_hist")
        return
    check(_lib.lib().sc_fp8_scale_update(amax_slots.data_ptr(), scale.data_ptr(), scale_inv.data_ptr(), n, int(margin_bits),
                                         _stream()), "sc_fp8_scale_update")


def gemm_wgrad_fp8(dy8: torch.Tensor, dy_scale_inv: torch.Tensor, x8: torch.Tensor, x_scale_inv: torch.Tensor, dw: torch.Tensor,
                   dbias: Optional[torch.Tensor], *, M: int, N: int, K: int, splitk: int = 1) -> None:
    """dW[M,N] = s_dy s_x dY8[K,M]^T . X8[K,N] (fp32) and dbias[M] = s_dy column sums of dY8: e4m3 operands with ONE scale per
    tensor (``*_scale_inv``: one-element device tensors)."""
    for t_, n in ((dy8, "dy8"), (x8, "x8")):
        if not t_.is_cuda or t_.dtype != torch.uint8 or t_.stride(-1) != 1:
            raise TypeError(f"gemm_wgrad_fp8: {n} must be a device uint8 (e4m3 bytes) matrix")
    _req(dy_scale_inv, torch.float32, "dy_scale_inv"); _req(x_scale_inv, torch.float32, "x_scale_inv"); _req(dw, torch.float32, "dw")
    if dbias is not None: _req(dbias, torch.float32, "dbias")
    if not dw.is_contiguous() or dw.numel() != M * N:
        raise ValueError("gemm_wgrad_fp8: dw must be a dense [M, N] tensor")
    l = _lib.lib()
    ws = workspace(l.sc_gemm_wgrad_ws_floats(M, N, K, splitk), dw.device, "wgrad")
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = l.sc_gemm_wgrad_fp8(dy8.data_ptr(), dy8.stride(0), dy_scale_inv.data_ptr(), x8.data_ptr(), x8.stride(0),
                             x_scale_inv.data_ptr(), M, N, K, dw.data_ptr(), N, _ptr(dbias), splitk, ws.data_ptr(), _stream())
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append(("gemm_tn_fp8", 2.0 * M * N * K, ev))
    check(rc, "sc_gemm_wgrad_fp8")


def gemm_fp8(epi: int, a8: torch.Tensor, a_scale_inv: torch.Tensor, b8: torch.Tensor, b_scale_inv: torch.Tensor,
             out: torch.Tensor, *, M: int, N: int, K: int, out2: Optional[torch.Tensor] = None,
             bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
             aux: Optional[torch.Tensor] = None, a_scale_scalar: bool = False, q8_out: Optional[torch.Tensor] = None,
             q8_scale: Optional[torch.Tensor] = None, q8_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[M,N] = dequant(A8[M,K] . B8[N,K]^T) with the bf16 GEMM's epilogues (see sc_gemm_fp8); ``aux`` = the pre-GELU
    tensor of EPI_BF16_DGELU (the c_proj data-gradient GEMM)."""
    for t_, n in ((a8, "a8"), (b8, "b8")):
        if not t_.is_cuda or t_.dtype != torch.uint8 or t_.stride(-1) != 1:
            raise TypeError(f"gemm_fp8: {n} must be a device uint8 (e4m3 bytes) matrix")
    _req(a_scale_inv, torch.float32, "a_scale_inv"); _req(b_scale_inv, torch.float32, "b_scale_inv")
    want = torch.float32 if epi in (EPI_F32, EPI_F32_BIAS_RES) else torch.bfloat16
    _req(out, want, "out")
    if res is not None: _req(res, torch.bfloat16 if epi == EPI_BF16_BIAS_RES else torch.float32, "res")
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if q8_out is not None and (q8_out.dtype != torch.uint8 or not q8_out.is_cuda or q8_out.stride(-1) != 1):
        raise TypeError("gemm_fp8: q8_out must be a device uint8 matrix")
    rc = _lib.lib().sc_gemm_fp8_q(epi, a8.data_ptr(), a8.stride(0), a_scale_inv.data_ptr(), int(a_scale_scalar),
                                  b8.data_ptr(), b8.stride(0), b_scale_inv.data_ptr(), M, N, K, out.data_ptr(),
                                  out.stride(0), _ptr(out2), out2.stride(0) if out2 is not None else 0, _ptr(bias),
                                  _ptr(res), res.stride(0) if res is not None else 0, _ptr(aux),
                                  aux.stride(0) if aux is not None else 0, _ptr(q8_out),
                                  q8_out.stride(0) if q8_out is not None else 0, _ptr(q8_scale), _ptr(q8_amax), _stream())
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append(("gemm_nt_fp8", 2.0 * M * N * K, ev))
    check(rc, "sc_gemm_fp8")
    return out
