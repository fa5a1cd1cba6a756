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

# ---- round 3: what the delayed-scaling second outputs cost, and what their consumers return
print("--- e4m3 second outputs of the GELU / GELU' epilogues (delayed per-tensor scale) and their consumers")
d, mlp = 1024, 4096
a = torch.randn(M, d, device="cuda", generator=g).bfloat16()
wfc = (torch.randn(mlp, d, device="cuda", generator=g) * 0.03).bfloat16()
bias = torch.randn(mlp, device="cuda", generator=g)
a8, sa = ops.quantize_rows_fp8(a)
w8, sw = ops.quantize_rows_fp8(wfc)
u = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16); h = torch.empty_like(u)
h8 = torch.zeros((M, mlp), dtype=torch.uint8, device="cuda")
sc = torch.full((2,), 16.0, device="cuda"); sci = 1.0 / sc; am = torch.zeros((2, 64), device="cuda")
t0 = timeit(lambda: ops.gemm_fp8(ops.EPI_GELU_PAIR, a8, sa, w8, sw, u, M=M, N=mlp, K=d, bias=bias, out2=h))
t1 = timeit(lambda: ops.gemm_fp8(ops.EPI_GELU_PAIR, a8, sa, w8, sw, u, M=M, N=mlp, K=d, bias=bias, out2=h, q8_out=h8,
                                 q8_scale=sc[0:1], q8_amax=am[0]))
print(f"c_fc fp8 GELU pair: {t0:7.1f} us;  + e4m3(h) output: {t1:7.1f} us  (+{t1 - t0:5.1f} us for {M * mlp / 1e6:.0f} MB)")
wpj = (torch.randn(d, mlp, device="cuda", generator=g) * 0.02).bfloat16()
wp8, swp = ops.quantize_rows_fp8(wpj)
res = torch.randn(M, d, device="cuda", generator=g); xo = torch.empty(M, d, device="cuda"); bd = torch.randn(d, device="cuda", generator=g)
tb = timeit(lambda: ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, h, wpj, xo, M=M, N=d, K=mlp, bias=bd, res=res))
t8 = timeit(lambda: ops.gemm_fp8(ops.EPI_F32_BIAS_RES, h8, sci[0:1], wp8, swp, xo, M=M, N=d, K=mlp, bias=bd, res=res, a_scale_scalar=True))
print(f"c_proj fwd: bf16 {tb:7.1f} us, fp8 (A = e4m3(h)) {t8:7.1f} us  (-{tb - t8:5.1f} us);  pair net {tb - t8 - (t1 - t0):+6.1f} us")
gq, gs = ops.quantize_rows_fp8(torch.randn(M, d, device="cuda", generator=g).bfloat16())
wb8, swb = ops.quantize_rows_fp8(wpj.t().contiguous())            # wb of c_proj: [mlp, d]
dU = torch.empty(M, mlp, device="cuda", dtype=torch.bfloat16); dU8 = torch.zeros((M, mlp), dtype=torch.uint8, device="cuda")
t0 = timeit(lambda: ops.gemm_fp8(ops.EPI_BF16_DGELU, gq, gs, wb8, swb, dU, M=M, N=mlp, K=d, aux=u))
t1 = timeit(lambda: ops.gemm_fp8(ops.EPI_BF16_DGELU, gq, gs, wb8, swb, dU, M=M, N=mlp, K=d, aux=u, q8_out=dU8, q8_scale=sc[1:2], q8_amax=am[1]))
print(f"c_proj dgrad fp8 GELU': {t0:7.1f} us;  + e4m3(dU) output: {t1:7.1f} us  (+{t1 - t0:5.1f} us)")
wfb = wfc.t().contiguous()                                        # wb of c_fc: [d, mlp]
wfb8, swfb = ops.quantize_rows_fp8(wfb)
dA = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
tb = timeit(lambda: ops.gemm(ops.NT, ops.EPI_BF16, dU, wfb, dA, M=M, N=d, K=mlp))
t8 = timeit(lambda: ops.gemm_fp8(ops.EPI_BF16, dU8, sci[1:2], wfb8, swfb, dA, M=M, N=d, K=mlp, a_scale_scalar=True))
print(f"c_fc dgrad: bf16 {tb:7.1f} us, fp8 (A = e4m3(dU)) {t8:7.1f} us  (-{tb - t8:5.1f} us);  pair net {tb - t8 - (t1 - t0):+6.1f} us")
