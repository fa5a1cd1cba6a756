#!/usr/bin/env python3
"""Attention forward / backward alone on the ViT-L/14 shape (B=256, L=257, H=16, dh=64): the persistent 225..288-token kernels
of round 5 (sc_attention_p2.hip, ...) against the per-head kernels (SC_ATTN_PERSIST2=0 / SC_ATTN_BWD4=0), interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

B, L, H, dh = int(os.environ.get("B", 256)), int(os.environ.get("L", 257)), int(os.environ.get("H", 16)), 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * L, 3 * d, device="cuda", generator=g).bfloat16()
out = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dout = (torch.randn(B * L, d, device="cuda", generator=g) * 0.05).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, L, device="cuda")
n = int(os.environ.get("N", 20))


def timeit(fn, name):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us", flush=True)


for rep in range(int(os.environ.get("REPS", 2))):
    for sw, tag in (("1", "persistent (round 5)"), ("0", "per-head kernels")):
        os.environ["SC_ATTN_PERSIST2"] = sw
        os.environ["SC_ATTN_BWD4"] = sw
        timeit(lambda: ops.attn_fwd(qkv, B, L, H, dh, False, out=out, lse=lse), f"fwd  {tag}")
        timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=dqkv, delta=delta), f"bwd  {tag}")
