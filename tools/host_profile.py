#!/usr/bin/env python3
"""cProfile of the host side of training steps (ViT-B/16 + gene-MLP, B = 256, resident synthetic batches): where the Python
time per step goes (the step is kernel-bound at ~34 ms with ~26 ms of host enqueue work)."""
import cProfile
import functools
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import comm, data, losses, module, net, optim, streams

n = net.SpatialClipNet("ViT-B-16-gene", None, n_genes=20000, seed=0)
m = module.SpatialClipLitModule(
    n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True),
    functools.partial(optim.FusedAdamW, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
    functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=2000))


class _T:
    max_steps, max_epochs, estimated_stepping_batches = 1_000_000, None, 1_000_000


m.trainer = _T()
oc = m.configure_optimizers()
opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
reducer = comm.GradBucketReducer(n.store.grad)
rates = data.make_gene_rates(20000)
batches = [{k: v.cuda() for k, v in data.synthetic_batch(256, 224, 20000, 8, s, 0, 1, rates).items()} for s in range(2)]


def step(i):
    with streams.chain_stream():
        loss = m.training_step(batches[i % 2], i)
        loss.backward()
        reducer.finish()
        opt.step(grad_scale=1.0, max_norm=1.0)
        sched.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for i in range(K):
    step(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host enqueue {t_host / K * 1e3:.2f} ms/step, total {(time.perf_counter() - t0) / K * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
