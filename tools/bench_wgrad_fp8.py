#!/usr/bin/env python3
"""e4m3 weight-gradient GEMM (sc_gemm_wgrad_fp8) against the bf16 TN kernel on the ViT-L/14 and ViT-B/16 MLP shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, sk in (("ViT-L c_fc  (4096 x 1024, 65792 tokens)", 4096, 1024, 65792, 4), ("ViT-L c_proj (1024 x 4096)", 1024, 4096, 65792, 4),
                          ("ViT-B c_fc  (3072 x 768, 50432 tokens)", 3072, 768, 50432, 7), ("ViT-B c_proj (768 x 3072)", 768, 3072, 50432, 7)):
    dy = torch.randn(K, M, device="cuda", generator=g)
    x = torch.randn(K, N, device="cuda", generator=g)
    dyb, xb = dy.bfloat16(), x.bfloat16()
    dy8, x8 = (dy * 16).to(torch.float8_e4m3fn).view(torch.uint8), (x * 16).to(torch.float8_e4m3fn).view(torch.uint8)
    s = torch.tensor([1.0 / 16], device="cuda")
    dw = torch.empty(M, N, device="cuda"); db = torch.empty(M, device="cuda")
    tb = timeit(lambda: ops.gemm_wgrad_bias(dyb, xb, dw, db, M=M, N=N, K=K, splitk=sk))
    ref = dw.clone()
    t8 = timeit(lambda: ops.gemm_wgrad_fp8(dy8, s, x8, s, dw, db, M=M, N=N, K=K, splitk=sk))
    rel = float((dw - ref).norm() / ref.norm())
    fl = 2.0 * M * N * K
    print(f"{name:42s} bf16 {tb:7.1f} us ({fl / tb / 1e6:6.0f} TFLOP/s)   e4m3 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TFLOP/s)   x{tb / t8:.2f}   rel L2 diff {rel:.3f}", flush=True)
