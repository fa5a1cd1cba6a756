#!/usr/bin/env python3
"""LayerNorm forward / backward kernels alone: time and bytes per launch.  Default = the headline shape (50 432 rows x 768);
``python tools/bench_ln.py 65792 1024`` = ViT-L/14 at 256 tiles of 257 tokens."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import ops

M, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256 * 197, 768)
print(f"rows {M} x d {d}  lib {os.environ.get('SC_HIP_LIB', 'default')}")
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, d, device=dev, generator=g)
x16 = x.to(torch.bfloat16)
gamma, beta = torch.randn(d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
y = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
dy = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
gin = torch.randn(M, d, device=dev, generator=g).to(torch.bfloat16)
gout = torch.empty_like(gin)
dres = torch.randn(M, d, device=dev, generator=g)
dg, db, cs = (torch.empty(d, device=dev) for _ in range(3))
big = torch.zeros(512 * 1024 * 1024 // 4, device=dev)           # READ between launches: clean lines push the operands out of the 256-MB Infinity Cache


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        big.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


E = M * d
cases = [
    ("ln_fwd  fp32 rows", lambda: ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, M, d), 6 * E),
    ("ln_fwd  bf16 rows", lambda: ops.layernorm_fwd(x16, gamma, beta, y, mean, rstd, M, d), 4 * E),
    ("ln_bwd  fp32 buffer (rounds 1-2)", lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, gout, dg, db, cs, M, d, accumulate=True), 16 * E),
    ("ln_bwd  bf16 gradient stream", lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, gout, dg, db, cs, M, d, accumulate=True, g16=True, g_in=gin, write_f32=False), 10 * E),
    ("ln_bwd  bf16 stream + bf16 rows", lambda: ops.layernorm_bwd(dy, x16, mean, rstd, gamma, dres, gout, dg, db, cs, M, d, accumulate=True, g16=True, g_in=gin, write_f32=False), 8 * E),
]
q8 = torch.empty(M, d, dtype=torch.uint8, device=dev)
sinv = torch.empty(M, device=dev)
t8 = (torch.empty(M, d, dtype=torch.uint8, device=dev), torch.full((1,), 16.0, device=dev), torch.zeros(64, device=dev))
cases += [
    ("ln_bwd  bf16 + per-row e4m3 copy", lambda: ops.layernorm_bwd(dy, x16, mean, rstd, gamma, dres, gout, dg, db, cs, M, d, accumulate=True, g16=True, g_in=gin, write_f32=False, q8=q8, q8_scale_inv=sinv), 9 * E),
    ("ln_bwd  bf16 + both e4m3 copies", lambda: ops.layernorm_bwd(dy, x16, mean, rstd, gamma, dres, gout, dg, db, cs, M, d, accumulate=True, g16=True, g_in=gin, write_f32=False, q8=q8, q8_scale_inv=sinv, t8=t8), 10 * E),
    ("ln_fwd  bf16 rows + per-row e4m3 copy", lambda: ops.layernorm_fwd(x16, gamma, beta, y, mean, rstd, M, d, q8=q8, q8_scale_inv=sinv), 5 * E),
]
ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, M, d)
for name, fn, nbytes in cases:
    us = t(fn)
    print(f"{name:36s} {us:7.1f} us   {nbytes / 1e6:6.0f} MB   {nbytes / us / 1e6:5.2f} TB/s")
