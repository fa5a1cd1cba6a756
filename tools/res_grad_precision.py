#!/usr/bin/env python3
"""Gradient error against the fp32 oracle (CPU autograd) with the residual gradient carried in fp32 (SC_RES_GRAD=fp32) and in
bf16 (default): a 12-block ViT (width 256, 65 tokens) + gene-MLP, SpatialLoss, B = 48, identical weights and batch.
Per parameter tensor: relative L2 error of the HIP gradient; printed: median / worst over the tensors, and the blocks' first
and last in_proj weights (the deepest tensor sees the most bf16 roundings of the stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import data, losses, model_configs as mc, module, net
from oracle import spatial_clip_oracle as O

LAYERS = 12
cfg = mc.ModelCfg(embed_dim=128, vision=mc.VisionCfg(64, 8, 256, LAYERS, 64), text=None, gene=mc.GeneCfg(2000, 256))
ocfg = O.ModelCfg(embed_dim=128, vision=O.VisionCfg(64, 8, 256, LAYERS, 64), text=None, gene=O.GeneCfg(2000, 256))
n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=3)
loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                             neighbor_alpha_scale=0.5, float32_logits=True)
m = module.SpatialClipLitModule(n, loss_fn, None, None)
batch = data.synthetic_batch(48, 64, 2000, K=4)
p = {k: v.cpu().requires_grad_(True) for k, v in n.state_dict().items()}
f = O.net_forward(batch["images"], batch["texts"], p, ocfg)
ref = O.spatial_loss(f["image_features"], f["text_features"], f["logit_scale"], batch["image_tile_ids"],
                     batch["text_tile_ids"], batch["neighbor_tile_ids"], batch["neighbor_alphas"])
ref.backward()
gb = {k: v.cuda() for k, v in batch.items()}
res = {}
for mode in ("fp32", "bf16"):
    os.environ["SC_RES_GRAD"] = mode
    n.store.grad.zero_()
    out = m.model_step(gb)
    out["loss"].backward()
    torch.cuda.synchronize()
    res[mode] = {k: n.store.g(k).detach().cpu().double().clone() for k in n.store.by_name}
    errs = {}
    for k, g in res[mode].items():
        gr = p[k].grad.double()
        if float(gr.norm()) > 1e-7:
            errs[k] = float((g - gr).norm() / gr.norm())
    v = np.array(list(errs.values()))
    first = errs["visual.transformer.resblocks.0.attn.in_proj_weight"]
    lastb = errs[f"visual.transformer.resblocks.{LAYERS - 1}.attn.in_proj_weight"]
    print(f"residual gradient in {mode}: loss {float(out['loss'].detach()):.6f} (oracle {float(ref.detach()):.6f}); relative L2 gradient "
          f"error vs the fp32 oracle over {len(v)} tensors: median {np.median(v):.4f}, worst {v.max():.4f} "
          f"({max(errs, key=errs.get)}); in_proj of block 0 {first:.4f}, of block {LAYERS - 1} {lastb:.4f}")
d = np.array([float((res["bf16"][k] - res["fp32"][k]).norm() / res["fp32"][k].norm().clamp_min(1e-12)) for k in res["fp32"]])
print(f"bf16 against fp32 stream: median relative L2 difference {np.median(d):.4f}, worst {d.max():.4f}")
