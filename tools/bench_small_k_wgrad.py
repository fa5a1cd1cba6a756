#!/usr/bin/env python3
"""Split-K of the weight gradients at SMALL token counts (ViT-B-32 + CLIP text tower at the reference's batch 32: K = 1600 vision
tokens = 25 K tiles, 2464 text tokens): how many splits before the fp32 slabs cost more than the idle CUs?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401,E402
from spatial_clip_amd import ops  # noqa: E402
from spatial_clip_amd.towers import _splitk_for  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)


def t(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, tokens, M, N in (("vision c_fc   (3072 x 768)", 1600, 3072, 768), ("vision c_proj (768 x 3072)", 1600, 768, 3072),
                           ("vision qkv    (2304 x 768)", 1600, 2304, 768), ("vision out    (768 x 768)", 1600, 768, 768),
                           ("text c_fc     (2048 x 512)", 2464, 2048, 512), ("text qkv      (1536 x 512)", 2464, 1536, 512),
                           ("text out      (512 x 512)", 2464, 512, 512)):
    dy = torch.randn(tokens, M, device="cuda", generator=g).bfloat16()
    x = torch.randn(tokens, N, device="cuda", generator=g).bfloat16()
    dw = torch.empty(M, N, device="cuda")
    db = torch.empty(M, device="cuda")
    line = f"{name} K = {tokens}: default split-K {_splitk_for(M, N, tokens)};"
    for sk in (1, 2, 3, 4, 6, 9):
        us = t(lambda: ops.gemm_wgrad_bias(dy, x, dw, db, M=M, N=N, K=tokens, splitk=sk))
        line += f"  sk={sk}: {us:6.1f} us"
    print(line, flush=True)
