#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernels on the ViT-B/16 (B=256) shapes.  SC_GEMM_FORCE=128 forces the general kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import ops

M = int(os.environ.get("M", 256 * 197))
dev = "cuda"


def run(name, mode, epi, Mo, No, K, **kw):
    g = torch.Generator(device=dev).manual_seed(0)
    if mode == ops.NT:
        a = torch.randn(Mo, K, device=dev, generator=g).bfloat16()
        b = (torch.randn(No, K, device=dev, generator=g) * 0.05).bfloat16()
    else:
        a = torch.randn(K, Mo, device=dev, generator=g).bfloat16()
        b = torch.randn(K, No, device=dev, generator=g).bfloat16()
    f32 = epi in (ops.EPI_F32, ops.EPI_F32_BIAS_RES)
    out = torch.empty(Mo, No, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    extra = {}
    if epi in (ops.EPI_BF16_BIAS, ops.EPI_GELU_PAIR, ops.EPI_F32_BIAS_RES):
        extra["bias"] = torch.randn(No, device=dev)
    if epi == ops.EPI_F32_BIAS_RES:
        extra["res"] = torch.randn(Mo, No, device=dev)
    if epi == ops.EPI_GELU_PAIR:
        extra["out2"] = torch.empty_like(out)
    if epi == ops.EPI_BF16_DGELU:
        extra["aux"] = torch.randn(Mo, No, device=dev).bfloat16()
    extra.update(kw)
    for _ in range(3):
        ops.gemm(mode, epi, a, b, out, M=Mo, N=No, K=K, **extra)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(mode, epi, a, b, out, M=Mo, N=No, K=K, **extra)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:28s} M={Mo:6d} N={No:5d} K={K:6d}  {us:8.1f} us  {2.0 * Mo * No * K / us / 1e6:7.1f} TF/s", flush=True)


def main():
    from spatial_clip_amd.towers import _splitk_for
    d, mlp = 768, 3072
    run("qkv fwd (bias)", ops.NT, ops.EPI_BF16_BIAS, M, 3 * d, d)
    run("out_proj fwd (res f32)", ops.NT, ops.EPI_F32_BIAS_RES, M, d, d)
    run("c_fc fwd (gelu pair)", ops.NT, ops.EPI_GELU_PAIR, M, mlp, d)
    run("c_proj fwd (res f32)", ops.NT, ops.EPI_F32_BIAS_RES, M, d, mlp)
    run("c_proj dgrad (dgelu)", ops.NT, ops.EPI_BF16_DGELU, M, mlp, d)
    run("c_fc dgrad", ops.NT, ops.EPI_BF16, M, d, mlp)
    run("out_proj dgrad", ops.NT, ops.EPI_BF16, M, d, d)
    run("qkv dgrad", ops.NT, ops.EPI_BF16, M, d, 3 * d)
    run("plain 4096^3", ops.NT, ops.EPI_BF16, 4096, 4096, 4096)
    run("c_proj wgrad", ops.TN, ops.EPI_F32, d, mlp, M, splitk=_splitk_for(d, mlp, M))
    run("c_fc wgrad", ops.TN, ops.EPI_F32, mlp, d, M, splitk=_splitk_for(mlp, d, M))
    run("out_proj wgrad", ops.TN, ops.EPI_F32, d, d, M, splitk=_splitk_for(d, d, M))
    run("qkv wgrad", ops.TN, ops.EPI_F32, 3 * d, d, M, splitk=_splitk_for(3 * d, d, M))
    

if __name__ == "__main__":
    main()
