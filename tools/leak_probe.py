#!/usr/bin/env python3
"""Does a net's device memory come back after the last reference is dropped?  Build a model, run one training step, drop it,
collect, and list which of its objects are still alive and who refers to them."""
import functools, gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import data, losses, module, net, optim, params, towers

STEP = os.environ.get("PROBE_STEP", "1") == "1"
BWD = os.environ.get("PROBE_BWD", "1") == "1"
OPT = os.environ.get("PROBE_OPT", "1") == "1"


def run():
    n = net.SpatialClipNet("ViT-Ti-16-gene", None, n_genes=512, seed=0)
    m = module.SpatialClipLitModule(
        n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True),
        functools.partial(optim.FusedAdamW, lr=1e-3), functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 10, None, 10
    m.trainer = T()
    oc = m.configure_optimizers() if OPT else None
    if STEP:
        b = {k: v.cuda() for k, v in data.synthetic_batch(16, 224, 512, K=4).items()}
        loss = m.training_step(b, 0)
        if BWD:
            loss.backward()
            if OPT:
                oc["optimizer"].step(grad_scale=1.0, max_norm=1.0)
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated()


base = torch.cuda.memory_allocated()
peak = run()
print(f"STEP={STEP} BWD={BWD} OPT={OPT}: allocated while the model lives:", (peak - base) >> 20, "MiB")
gc.collect()
left = torch.cuda.memory_allocated() - base
print("after gc.collect():", left >> 20, "MiB")
kinds = (net.SpatialClipNet, module.SpatialClipLitModule, params.ParamStore, towers.TransformerStack, optim.FusedAdamW, towers._Bufs)
alive = [o for o in gc.get_objects() if isinstance(o, kinds)]
print("alive:", [type(o).__name__ for o in alive])
me = sys._getframe()
for o in alive[:3]:
    print("==", type(o).__name__)
    for r in gc.get_referrers(o):
        if r is alive or r is me or isinstance(r, type(me)):
            continue
        d = type(r).__name__
        if isinstance(r, dict):
            d += " keys=" + str(list(r.keys())[:8])
            owners = [type(x).__name__ for x in gc.get_referrers(r) if not isinstance(x, type(me))][:4]
            d += " owned by " + str(owners)
        elif isinstance(r, torch.Tensor):
            d += f" shape={tuple(r.shape)} requires_grad={r.requires_grad} grad_fn={r.grad_fn}"
        print("   <-", d[:220])
