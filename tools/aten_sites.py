"""Which Python lines of the training step still launch ATen kernels or runtime copies?  Runs a few steps of the bench
model (small batch: the launch COUNT does not depend on it) under torch.profiler with Python stacks and prints, per
step, every aten:: operator that reached the device together with the innermost frame inside this repository.

    python tools/aten_sites.py [--model ViT-B-16-gene] [--batch 32] [--steps 3]"""
import argparse
import collections
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B-16-gene")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--loss", default="clip")
    args = ap.parse_args()
    import torch
    from torch.profiler import ProfilerActivity, profile
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, module, net, optim, streams
    n = net.SpatialClipNet(args.model, None, n_genes=20000, seed=0)
    cfg = n.cfg
    if args.loss == "clip":
        loss_fn = losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
    else:
        loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                                     neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=2000))

    class _T:
        max_steps, max_epochs, estimated_stepping_batches = 1_000_000, None, 1_000_000
    m.trainer = _T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.GradBucketReducer(n.store.grad)
    rates = data.make_gene_rates(20000)
    batches = [{k: v.cuda() for k, v in data.synthetic_batch(args.batch, cfg.vision.image_size, 20000, 8, s, 0, 1, rates).items()}
               for s in range(2)]

    def step(i):
        with streams.chain_stream():
            loss = m.training_step(batches[i % 2], i)
            loss.backward(m.root_gradient(loss))
            reducer.finish()
            opt.step(grad_scale=1.0, max_norm=1.0)
            sched.step()

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(args.steps):
            step(3 + i)
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sites = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
            continue
        if any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
            continue                                    # count the innermost operator that owns the device work
        frame = next((f for f in (ev.stack or []) if root in f and "tools/aten_sites" not in f), "(no repo frame)")
        sites[(ev.name, frame.replace(root + "/", ""))] += 1
    print(f"aten operators with device work per step ({args.steps} steps profiled):")
    tot = 0
    for (name, frame), cnt in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"  {cnt / args.steps:6.2f}  {name:28s} {frame}")
        tot += cnt
    print(f"  total {tot / args.steps:.1f} per step")


if __name__ == "__main__":
    main()
