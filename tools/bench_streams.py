#!/usr/bin/env python3
"""Does running two independent half-batch GEMM chains on two HIP streams overlap epilogue traffic with MFMA work?
Compares: (a) one stream, full batch (M=50432) per launch; (b) two streams, half batch each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

dev = "cuda"
M = 256 * 197
d, mlp = 768, 3072


def mk(Mr):
    g = torch.Generator(device=dev).manual_seed(0)
    t = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    return dict(a=t(Mr, d), wqkv=t(3 * d, d) * 0.05, wo=t(d, d) * 0.05, wfc=t(mlp, d) * 0.05, wpr=t(d, mlp) * 0.05,
                qkv=torch.empty(Mr, 3 * d, device=dev, dtype=torch.bfloat16), x=torch.randn(Mr, d, device=dev),
                xo=torch.empty(Mr, d, device=dev), u=torch.empty(Mr, mlp, device=dev, dtype=torch.bfloat16),
                h=torch.empty(Mr, mlp, device=dev, dtype=torch.bfloat16), bq=torch.randn(3 * d, device=dev),
                bo=torch.randn(d, device=dev), bf=torch.randn(mlp, device=dev))


def layer(b, Mr):
    ops.gemm(ops.NT, ops.EPI_BF16_BIAS, b["a"], b["wqkv"], b["qkv"], M=Mr, N=3 * d, K=d, bias=b["bq"])
    ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, b["a"], b["wo"], b["xo"], M=Mr, N=d, K=d, bias=b["bo"], res=b["x"])
    ops.gemm(ops.NT, ops.EPI_GELU_PAIR, b["a"], b["wfc"], b["u"], M=Mr, N=mlp, K=d, bias=b["bf"], out2=b["h"])
    ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, b["h"], b["wpr"], b["xo"], M=Mr, N=d, K=mlp, bias=b["bo"], res=b["x"])


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


full = mk(M)
t1 = timeit(lambda: layer(full, M))
print(f"one stream, full batch: {t1:8.1f} us per layer-forward GEMM set ({os.environ.get('SC_GEMM_FORCE', 'default')})")
halves = [mk(M // 2), mk(M // 2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def two():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            layer(halves[k], M // 2)
    for s in streams:
        cur.wait_stream(s)


t2 = timeit(two)
print(f"two streams, half batch each: {t2:8.1f} us")
