#!/usr/bin/env python3
"""Attention forward / backward alone on the ViT-B/16 shape (B=256, L=197, H=12, dh=64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

B, L, H, dh = int(os.environ.get("B", 256)), int(os.environ.get("L", 197)), int(os.environ.get("H", 12)), 64      # B=3072 H=1: the same bytes head-major
d = H * dh
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * L, 3 * d, device="cuda", generator=g).bfloat16()
out = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dout = torch.randn(B * L, d, device="cuda", generator=g).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, L, device="cuda")
n = int(os.environ.get("N", 20))


def timeit(fn, name, bytes_):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:16s} {us:8.1f} us   {bytes_ / us / 1e6:6.2f} TB/s algorithmic", flush=True)


qb = qkv.numel() * 2
ob = out.numel() * 2
timeit(lambda: ops.attn_fwd(qkv, B, L, H, dh, False, out=out, lse=lse), "fwd", qb + ob)
for rep in range(int(os.environ.get("REPS", 2))):
    for name, b3, b2, b1 in (("bwd ring (r4)", "1", "0", "0"), ("bwd single-pass", "0", "0", "1"), ("bwd persistent", "0", "1", "0"),
                             ("bwd per-head", "0", "0", "0")):
        os.environ["SC_ATTN_BWD3"], os.environ["SC_ATTN_BWD2"], os.environ["SC_ATTN_BWD1"] = b3, b2, b1
        timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=dqkv, delta=delta), name, 2 * qb + 2 * ob)       # qkv read, dqkv written, out and dout read
os.environ.pop("SC_ATTN_BWD3"); os.environ.pop("SC_ATTN_BWD2"); os.environ.pop("SC_ATTN_BWD1")
