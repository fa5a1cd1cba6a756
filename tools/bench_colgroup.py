#!/usr/bin/env python3
"""Round 5: the column-group tile walk of gemm8p_kernel (SC_GEMM_COLGROUP="<epi>:<Gc>") per launch class, alone on the chip.
For each shape: row-major walk (Gc = 0) against groups of Gc column tiles inside per-XCD row bands, interleaved, three rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import ops

dev = "cuda"


def make(epi, M, N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
    f32 = epi in (ops.EPI_F32, ops.EPI_F32_BIAS_RES)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BF16_BIAS, ops.EPI_GELU_GRAD_PAIR, ops.EPI_F32_BIAS_RES, ops.EPI_BF16_BIAS_RES):
        kw["bias"] = torch.randn(N, device=dev)
    if epi == ops.EPI_F32_BIAS_RES:
        kw["res"] = torch.randn(M, N, device=dev)
    if epi == ops.EPI_BF16_BIAS_RES:
        kw["res"] = torch.randn(M, N, device=dev).bfloat16()
    if epi == ops.EPI_GELU_GRAD_PAIR:
        kw["out2"] = torch.empty_like(out)
    if epi == ops.EPI_BF16_MUL_AUX:
        kw["aux"] = torch.randn(M, N, device=dev).bfloat16()
    return a, b, out, kw


def time_one(epi, M, N, K, gc, sets, n=12):
    os.environ["SC_GEMM_COLGROUP"] = f"{epi}:{gc}"
    for a, b, out, kw in sets:
        ops.gemm(ops.NT, epi, a, b, out, M=M, N=N, K=K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        a, b, out, kw = sets[i % len(sets)]
        ops.gemm(ops.NT, epi, a, b, out, M=M, N=N, K=K, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    shapes = [("ViT-B c_fc fwd (gelu'|gelu pair)", ops.EPI_GELU_GRAD_PAIR, 256 * 197, 3072, 768, (0, 2, 3, 4, 6)),
              ("ViT-B c_proj dgrad (x gelu')", ops.EPI_BF16_MUL_AUX, 256 * 197, 3072, 768, (0, 2, 3, 4, 6)),
              ("ViT-B c_fc dgrad (plain, N=768)", ops.EPI_BF16, 256 * 197, 768, 3072, (0, 1, 2)),
              ("ViT-B c_proj fwd (bf16 res, N=768)", ops.EPI_BF16_BIAS_RES, 256 * 197, 768, 3072, (0, 1, 2)),
              ("ViT-B qkv dgrad (plain, N=768)", ops.EPI_BF16, 256 * 197, 768, 2304, (0, 1, 2)),
              ("ViT-L c_fc fwd (pair)", ops.EPI_GELU_GRAD_PAIR, 256 * 257, 4096, 1024, (0, 2, 4, 8)),
              ("ViT-L c_proj dgrad (x gelu')", ops.EPI_BF16_MUL_AUX, 256 * 257, 4096, 1024, (0, 2, 4, 8))]
    for name, epi, M, N, K, gcs in shapes:
        sets = [make(epi, M, N, K, s) for s in range(2)]
        res = {gc: [] for gc in gcs}
        for _ in range(3):
            for gc in gcs:
                res[gc].append(time_one(epi, M, N, K, gc, sets))
        base = min(res[gcs[0]])
        print(f"{name:36s} M={M} N={N} K={K}: " + "  ".join(
            f"Gc={gc}: {min(v):6.1f} us ({2.0 * M * N * K / min(v) / 1e6:5.0f} TF/s, x{min(v) / base:.3f})" for gc, v in res.items()), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
