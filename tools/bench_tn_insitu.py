#!/usr/bin/env python3
"""Weight-gradient GEMMs as the training step issues them: the four shapes of a ViT-B/16 block in sequence (two of them
with the fused bias gradient), operands rotating over three layers' worth of buffers so that nothing stays cached, 8-wave
vs 4-wave kernel (SC_GEMM_TN4W)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

M, d, mlp = 256 * 197, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
L = 3
bufs = [dict(g0=torch.randn(M, d, device="cuda", generator=g).bfloat16(), h=torch.randn(M, mlp, device="cuda", generator=g).bfloat16(),
             dU=torch.randn(M, mlp, device="cuda", generator=g).bfloat16(), a2=torch.randn(M, d, device="cuda", generator=g).bfloat16(),
             o=torch.randn(M, d, device="cuda", generator=g).bfloat16(), dqkv=torch.randn(M, 3 * d, device="cuda", generator=g).bfloat16())
        for _ in range(L)]
w = dict(proj=torch.empty(d, mlp, device="cuda"), fc=torch.empty(mlp, d, device="cuda"), out=torch.empty(d, d, device="cuda"),
         qkv=torch.empty(3 * d, d, device="cuda"), bfc=torch.empty(mlp, device="cuda"), bqkv=torch.empty(3 * d, device="cuda"))


def block(b):
    ops.gemm(ops.TN, ops.EPI_F32, b["g0"], b["h"], w["proj"], M=d, N=mlp, K=M, splitk=7)
    ops.gemm_wgrad_bias(b["dU"], b["a2"], w["fc"], w["bfc"], M=mlp, N=d, K=M, splitk=7)
    ops.gemm(ops.TN, ops.EPI_F32, b["g0"], b["o"], w["out"], M=d, N=d, K=M, splitk=28)
    ops.gemm_wgrad_bias(b["dqkv"], b["a2"], w["qkv"], w["bqkv"], M=3 * d, N=d, K=M, splitk=9)


for four in ("0", "1", "0", "1"):
    os.environ["SC_GEMM_TN4W"] = four
    for i in range(3):
        block(bufs[i % L])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 12
    for i in range(n):
        block(bufs[i % L])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"four_wave={four}: {us:7.1f} us per block (4 wgrad GEMMs + slab reductions), {2.0 * M * (2 * d * mlp + d * d + 3 * d * d) / us / 1e6:6.0f} TF/s", flush=True)
