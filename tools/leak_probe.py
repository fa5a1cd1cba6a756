#!/usr/bin/env python3
"""Does a net's device memory come back after the last reference is dropped?  Build a model, run one training step, drop it,
collect, and list what still holds device tensors (and who refers to them)."""
import functools, gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import data, losses, module, net, optim


def run():
    n = net.SpatialClipNet("ViT-Ti-16-gene", None, n_genes=512, seed=0)
    m = module.SpatialClipLitModule(
        n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True),
        functools.partial(optim.FusedAdamW, lr=1e-3), functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 10, None, 10
    m.trainer = T()
    oc = m.configure_optimizers()
    b = {k: v.cuda() for k, v in data.synthetic_batch(16, 224, 512, K=4).items()}
    loss = m.training_step(b, 0)
    loss.backward()
    oc["optimizer"].step(grad_scale=1.0, max_norm=1.0)
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated()


base = torch.cuda.memory_allocated()
peak = run()
print("allocated while the model lives:", (peak - base) >> 20, "MiB")
print("after return, before gc:", (torch.cuda.memory_allocated() - base) >> 20, "MiB")
gc.collect()
left = torch.cuda.memory_allocated() - base
print("after gc.collect():", left >> 20, "MiB")
if left > (8 << 20):
    big = sorted((o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() > (1 << 20)),
                 key=lambda t: -t.numel() * t.element_size())
    print(len(big), "large device tensors still alive")
    for t in big[:4]:
        print("tensor", tuple(t.shape), t.dtype)
        seen = set()
        frontier = [t]
        for depth in range(5):
            nxt = []
            for o in frontier:
                for r in gc.get_referrers(o):
                    if id(r) in seen or r is frontier or r is big:
                        continue
                    seen.add(id(r))
                    desc = type(r).__name__
                    if isinstance(r, dict):
                        desc += " keys=" + str(list(r.keys())[:6])
                    print("  " * (depth + 1) + "<-", desc[:160])
                    nxt.append(r)
            frontier = nxt[:3]
