"""BASELINE.json full-size configuration (ViT-B/16 + gene-MLP 20000->512->512, local batch 256) checked through
size-independent properties -- the fp32 CPU oracle cannot run this size in seconds:
  * determinism: two runs from the same seed give bit-identical losses and gradients (no float atomics on this path);
  * the device grad-norm equals the norm of the flat gradient buffer; clipping scales the AdamW update accordingly;
  * ClipLoss is invariant under a permutation of the batch (rows and columns permuted together);
  * ClipLoss == SpatialLoss with no neighbours, no cap, no regulariser (the multi-positive loss degenerates);
  * a few optimisation steps on one fixed batch decrease the loss (the whole fwd/bwd/AdamW chain has the right sign)."""
import functools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pkg():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import data, losses, module, net, optim
    return data, losses, module, net, optim


def _module(net_mod, module_mod, losses_mod, optim_mod, loss_fn, seed=0, lr=3e-4):
    n = net_mod.SpatialClipNet("ViT-B-16-gene", None, n_genes=20000, seed=seed)
    m = module_mod.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim_mod.FusedAdamW, lr=lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim_mod.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 100, None, 100
    m.trainer = T()
    return n, m


def test_vitb16_b256_properties():
    data, losses, module, net, optim = _pkg()
    B = 256
    batch = {k: v.cuda() for k, v in data.synthetic_batch(B, 224, 20000, K=8).items()}
    clip = losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
    n, m = _module(net, module, losses, optim, clip)
    out = m.model_step(batch)
    out["loss"].backward()
    torch.cuda.synchronize()
    loss1 = float(out["loss"].detach())
    g1 = n.store.grad.clone()
    assert torch.isfinite(g1).all() and loss1 == loss1
    # unit-norm features
    assert (out["image_features"].norm(dim=1) - 1).abs().max() < 1e-5
    assert (out["text_features"].norm(dim=1) - 1).abs().max() < 1e-5

    # determinism (same weights, same batch, second pass)
    out2 = m.model_step(batch)
    out2["loss"].backward()
    torch.cuda.synchronize()
    assert float(out2["loss"].detach()) == loss1
    assert torch.equal(n.store.grad, g1)

    # grad-norm kernel vs the flat buffer
    from spatial_clip_amd import ops
    nc = torch.empty(2, device="cuda")
    ops.grad_norm(n.store.grad, n.store.total, 1.0, 1.0, nc)
    ref_norm = float(n.store.grad.double().norm())
    assert abs(float(nc[0]) - ref_norm) < 1e-4 * ref_norm
    assert abs(float(nc[1]) - min(1.0, 1.0 / (ref_norm + 1e-6))) < 1e-5

    # permutation invariance of the symmetric InfoNCE loss
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    pb = {k: v[perm] for k, v in batch.items()}
    with torch.no_grad():
        lp = float(m.model_step(pb)["loss"])
    assert abs(lp - loss1) < 2e-4, (lp, loss1)

    # SpatialLoss degenerates to ClipLoss without neighbours / cap / regulariser
    sp = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=None, temp_reg_weight=0.0,
                            neighbor_alpha_scale=0.0, float32_logits=True)
    m2 = module.SpatialClipLitModule(n, sp, None, None)
    with torch.no_grad():
        ls = float(m2.model_step(batch)["loss"])
    assert abs(ls - loss1) < 1e-5, (ls, loss1)


def test_vitb16_b256_overfits_one_batch():
    data, losses, module, net, optim = _pkg()
    B = 256
    batch = {k: v.cuda() for k, v in data.synthetic_batch(B, 224, 20000, K=8).items()}
    sp = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                            neighbor_alpha_scale=0.5, float32_logits=True)
    n, m = _module(net, module, losses, optim, sp, seed=1)
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    ls = []
    for step in range(8):
        loss = m.training_step(batch, step)
        loss.backward()
        opt.step(grad_scale=1.0, max_norm=1.0)
        sched.step()
        ls.append(float(loss.detach()))
    assert all(l == l for l in ls)
    assert ls[-1] < ls[1] - 0.05, ls          # step 0 runs with lr = 0 (LambdaLR), so compare from step 1
    r = m.train_metrics.compute()
    assert 0.0 <= r["train/R@1"] <= r["train/R@10"] <= 1.0
