#!/usr/bin/env python3
"""fp8 (e4m3) vs bf16 forward GEMMs on the ViT-L/14 shapes of BASELINE configs[4] (B = 256: M = 65792 tokens), and the
row quantiser that feeds them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

M = int(os.environ.get("M", 65792))
n = int(os.environ.get("N", 10))


def timeit(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in (("qkv", 3072, 1024, "bias"), ("out_proj", 1024, 1024, "res"), ("c_fc", 4096, 1024, "gelu"), ("c_proj", 1024, 4096, "res")):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    if epi == "res":
        res = torch.randn(M, N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        fb = lambda: ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, a, w, out, M=M, N=N, K=K, bias=bias, res=res)
        f8 = lambda: ops.gemm_fp8(ops.EPI_F32_BIAS_RES, a8, sa, w8, sw, out, M=M, N=N, K=K, bias=bias, res=res)
    elif epi == "gelu":
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out2 = torch.empty_like(out)
        fb = lambda: ops.gemm(ops.NT, ops.EPI_GELU_PAIR, a, w, out, M=M, N=N, K=K, bias=bias, out2=out2)
        f8 = lambda: ops.gemm_fp8(ops.EPI_GELU_PAIR, a8, sa, w8, sw, out, M=M, N=N, K=K, bias=bias, out2=out2)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fb = lambda: ops.gemm(ops.NT, ops.EPI_BF16_BIAS, a, w, out, M=M, N=N, K=K, bias=bias)
        f8 = lambda: ops.gemm_fp8(ops.EPI_BF16_BIAS, a8, sa, w8, sw, out, M=M, N=N, K=K, bias=bias)
    tq = timeit(lambda: ops.quantize_rows_fp8(a, a8, sa))
    tb, t8 = timeit(fb), timeit(f8)
    fl = 2.0 * M * N * K
    print(f"{name:9s} N={N:5d} K={K:5d}  bf16 {tb:7.1f} us {fl / tb / 1e6:7.0f} TF   fp8 {t8:7.1f} us {fl / t8 / 1e6:7.0f} TF   x{tb / t8:4.2f}   "
          f"quantise A {tq:6.1f} us ({(a.numel() * 3) / tq / 1e6:4.2f} TB/s)   fp8 + quantise x{tb / (t8 + tq):4.2f}", flush=True)
