#!/usr/bin/env python3
"""Timing of the contrastive head (2 similarity GEMMs + label join + row passes + 4 gradient GEMMs) at the per-rank
sizes of BASELINE configs [2]/[3] (B=256, G=2048, D=512) and [4] (B=1024, G=8192, D=768), plus the bare GEMM rates
of the exact-fp32 MFMA kernel.  Run on the GPU box:  python tools/bench_head.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import contrastive as C, ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def head(B, W, D, K=8, mode="spatial"):
    G = B * W
    g = torch.Generator(device="cuda").manual_seed(0)
    img = torch.nn.functional.normalize(torch.randn(G, D, device="cuda", generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img + 0.7 * torch.randn(G, D, device="cuda", generator=g), dim=-1)
    ids = 10_000 + torch.randperm(G, device="cuda", generator=g)
    r = W // 2
    sl = slice(r * B, (r + 1) * B)
    nb = ids[torch.randint(0, G, (B, K), device="cuda", generator=g)]
    al = torch.rand(B, K, device="cuda", generator=g)
    s = torch.tensor(14.2857, device="cuda")
    kw = dict(mode=mode, all_image=img, all_text=txt, rank=r)
    if mode == "spatial":
        kw.update(image_tile_ids=ids[sl], text_tile_ids=ids[sl], all_image_tile_ids=ids, all_text_tile_ids=ids,
                  neighbor_tile_ids=nb, neighbor_alphas=al, cap_logit_scale=40.0, temp_reg_weight=0.05,
                  neighbor_alpha_scale=0.5)
    fi, ft = img[sl].contiguous(), txt[sl].contiguous()
    ms = timeit(lambda: C.contrastive_forward_backward(fi, ft, s, **kw))
    flops = 2.0 * B * G * D * 6
    print(f"head {mode:8s} B={B} G={G} D={D}: {ms:.3f} ms  ({flops / ms / 1e9:.1f} TFLOP/s over the six products, wall)")
    # bare GEMM groups
    z = torch.empty(2, B, G, device="cuda")
    ms_f = timeit(lambda: ops.sgemm_grouped([(fi, D, 1, txt, D, 1, z[0], G, B, G, D), (ft, D, 1, img, D, 1, z[1], G, B, G, D)]))
    d1, d2 = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda")
    da = torch.empty(G, 2 * D, device="cuda")
    ms_b = timeit(lambda: ops.sgemm_grouped([
        (z[0], G, 1, txt, 1, D, d1, D, B, D, G), (z[1], G, 1, img, 1, D, d2, D, B, D, G),
        (z[0], 1, G, fi, 1, D, da[:, D:], 2 * D, G, D, B), (z[1], 1, G, ft, 1, D, da[:, :D], 2 * D, G, D, B)]))
    print(f"   forward pair {ms_f * 1e3:.1f} us ({4.0 * B * G * D / ms_f / 1e9:.1f} TFLOP/s), "
          f"backward four {ms_b * 1e3:.1f} us ({8.0 * B * G * D / ms_b / 1e9:.1f} TFLOP/s)")


if __name__ == "__main__":
    for mode in ("clip", "spatial"):
        head(256, 8, 512, mode=mode)
    head(1024, 8, 768, mode="spatial")
    for n in (2048, 4096):
        a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda"); c = torch.empty(n, n, device="cuda")
        ms = timeit(lambda: ops.sgemm(a, n, 1, b, n, 1, c, n, n, n, n))
        print(f"sgemm {n}^3 (NT): {ms * 1e3:.0f} us = {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s of 157 (fp32 MFMA peak)")
