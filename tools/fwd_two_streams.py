#!/usr/bin/env python3
"""Would two half-batches on two HIP streams fill the forward's idle CUs (partly filled last GEMM rounds, HBM-bound LayerNorm /
attention phases)?  Image-tower forward of ViT-B/16: one B=256 pass against two B=128 passes, back to back and concurrent."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import net

B = int(os.environ.get("B", 256))
model = os.environ.get("MODEL", "ViT-B-16-gene")
full = net.SpatialClipNet(model, None, n_genes=20000, seed=0)
halves = [net.SpatialClipNet(model, None, n_genes=20000, seed=0) for _ in range(2)]
size = full.cfg.vision.image_size
img = torch.randn(B, 3, size, size, device="cuda")
h = [img[:B // 2].contiguous(), img[B // 2:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
n = int(os.environ.get("N", 10))


def timeit(fn, name):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / n:8.3f} ms", flush=True)


def one():
    full.vision.forward(img)


def seq():
    halves[0].vision.forward(h[0])
    halves[1].vision.forward(h[1])


def par():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for k in range(2):
        streams[k].wait_event(ev)
        with torch.cuda.stream(streams[k]):
            halves[k].vision.forward(h[k])
            done = torch.cuda.Event()
            done.record(streams[k])
        main.wait_event(done)


with torch.no_grad():
    for rep in range(2):
        timeit(one, f"one pass, B = {B}")
        timeit(seq, f"two passes of B = {B // 2}, one stream")
        timeit(par, f"two passes of B = {B // 2}, two streams")
