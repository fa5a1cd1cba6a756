"""Two data-parallel ranks on ONE GPU (gloo backend with device tensors) against a single process on the concatenated
batch: exercises the product's whole multi-GPU code path (packed feature all-gather, reduce-scatter of the gather
gradient, bucketed gradient all-reduce launched from backward, 1/world_size folded into AdamW) on real kernels.
RCCL itself cannot be exercised on a one-GPU box; the collectives differ only in the backend string."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, steps, B, ret):
    import sys
    sys.path.insert(0, ROOT)
    import functools
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, model_configs as mc, module, net, optim
    cfg = mc.ModelCfg(embed_dim=64, vision=mc.VisionCfg(32, 8, 64, 2, 32), text=None, gene=mc.GeneCfg(200, 64))
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=7)
    loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.0,
                                 neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 10, None, 10
    m.trainer = T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.GradBucketReducer(n.store.grad, bucket_floats=20000)
    n.grad_bucket_hook = reducer.bucket_ready if world > 1 else None
    losses_out = []
    for s in range(steps):
        if world > 1:
            b = data.synthetic_batch(B, 32, 200, 4, s, rank, world)
        else:       # the concatenation of what the two ranks see
            parts = [data.synthetic_batch(B // 2, 32, 200, 4, s, r, 2) for r in range(2)]
            b = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        loss = m.training_step({k: v.cuda() for k, v in b.items()}, s)
        loss.backward()
        reducer.finish()
        opt.step(grad_scale=1.0 / world, max_norm=1.0)
        sched.step()
        losses_out.append(float(loss.detach()))
    torch.cuda.synchronize()
    ret[(world, rank)] = {"loss": losses_out, "proj": n.store.p("visual.proj").cpu(),
                          "fc2": n.store.p("gene.fc2.weight").cpu(), "qkv": n.store.p(
                              "visual.transformer.resblocks.0.attn.in_proj_weight").cpu()}
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_match_single_process():
    mp.set_start_method("spawn", force=True)
    steps, B = 3, 8
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_run, args=(1, 0, steps, 2 * B, ret), nprocs=1, join=True)
        mp.spawn(_run, args=(2, 29713, steps, B, ret), nprocs=2, join=True)
        res = dict(ret)
    one, r0, r1 = res[(1, 0)], res[(2, 0)], res[(2, 1)]
    for k in ("proj", "fc2", "qkv"):
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"          # same reduced gradients -> same weights
        assert float((r0[k] - one[k]).abs().max()) < 3e-3, k                  # and they follow the single-process run
    for s in range(steps):
        mean2 = 0.5 * (r0["loss"][s] + r1["loss"][s])                         # mean of the per-rank local losses
        assert abs(mean2 - one["loss"][s]) < (4e-3 if s < 2 else 2e-2), (s, mean2, one["loss"][s])
