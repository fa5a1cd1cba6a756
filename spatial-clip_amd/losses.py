"""ClipLoss / SpatialLoss with the reference's constructor kwargs, forward kwargs and return contract
(``{"contrastive_loss": 0-d tensor}``), computed by the fused device-side contrastive head.

Mirrors ``src/models/components/losses.py:11-141`` (SpatialLoss; ClipLoss wrapper over open_clip's ClipLoss,
``src/open_clip/loss.py:68-155``).  Distributed behaviour is the INTENDED one: rank / world size
are read from the live process group at call time, the global batch is formed with one packed all-gather and the
gradient of the gather (``gather_with_grad=True``) is one reduce-scatter."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import comm, ops
from .contrastive import contrastive_forward_backward


class _ContrastiveFn(torch.autograd.Function):
    """Forward computes the loss AND its feature / scale gradients (for an upstream gradient of 1); backward
    scales them by the actual upstream gradient."""

    @staticmethod
    def forward(ctx, image_features, text_features, logit_scale, owner, kw):
        rank, W = comm.world()
        dist_on = comm.is_dist()
        ids_i, ids_t = kw.get("image_tile_ids"), kw.get("text_tile_ids")
        img = image_features.detach().contiguous().float()
        txt = text_features.detach().contiguous().float()
        B, D = img.shape
        all_i = all_t = all_ids_i = all_ids_t = late = None
        fg = owner.prefetched
        need_ids = owner.mode == "spatial"
        if dist_on and fg is not None and fg.has("text", txt) and fg.has("image", img) \
                and (not need_ids or fg.with_ids("text")):
            # the gathers were launched from inside the net's forward on the communication stream; the image
            # features are only waited for after the first similarity GEMM has been enqueued
            all_t, all_ids_i, all_ids_t = fg.take("text")
            late = lambda: fg.take("image")[0]
        elif dist_on:
            all_i, all_t, all_ids_i, all_ids_t = comm.gather_packed(img, txt, ids_i if need_ids else None,
                                                                     ids_t if need_ids else None)
        common = dict(mode=owner.mode, image_tile_ids=ids_i, text_tile_ids=ids_t, all_image_tile_ids=all_ids_i,
                      all_text_tile_ids=all_ids_t, neighbor_tile_ids=kw.get("neighbor_tile_ids"),
                      neighbor_alphas=kw.get("neighbor_alphas"), cap_logit_scale=owner.cap_logit_scale,
                      temp_reg_weight=owner.temp_reg_weight, neighbor_alpha_scale=owner.neighbor_alpha_scale,
                      logit_bias=kw.get("logit_bias"), recall_hits=owner.recall_hits, want_recall=False)
        if dist_on and owner.mode == "clip" and not owner.local_loss:
            # loss.py:119-121: every rank forms the full [G,G] logits of the global batch (labels arange(G))
            if late is not None:
                all_i = late()
            res = contrastive_forward_backward(all_i, all_t, logit_scale.detach(), rank=0,
                                               recall_rows=(rank * B, B), **common)
            tot = torch.cat([res["d_image"] + res["d_all_image"], res["d_text"] + res["d_all_text"]], dim=1)
            if owner.gather_with_grad:
                both = comm.reduce_scatter_sum(tot)                 # every rank contributes its (identical) full-loss term
            else:
                both = tot[rank * B:(rank + 1) * B]                  # loss.py:58-61: only the spliced local shard has grad
            d_img, d_txt = both[:, :D].contiguous(), both[:, D:].contiguous()
        else:
            res = contrastive_forward_backward(img, txt, logit_scale.detach(), all_image=all_i, all_text=all_t,
                                               rank=rank if dist_on else 0, late_all_image=late, join_local=not dist_on,
                                               **common)
            if dist_on and owner.gather_with_grad:
                # autograd of torch.distributed.nn.all_gather (loss.py:50-52): SUM over ranks, keep own rows
                both = comm.reduce_scatter_sum(res["d_all"])
                d_img = res["d_image"] + both[:, :D]
                d_txt = res["d_text"] + both[:, D:]
            elif dist_on:
                d_img, d_txt = res["d_image"], res["d_text"]     # loss.py:54-63 with local_loss: remote shards detached
            else:           # single process: the head's GEMMs accumulated the gathered-feature terms onto the direct ones
                d_img, d_txt = res["d_image"], res["d_text"]
        ctx.flat = res["grads"] if d_img is res["d_image"] and d_txt is res["d_text"] else None
        ctx.save_for_backward(d_img, d_txt, res["d_scale"])
        owner.last = res
        return res["loss"]

    @staticmethod
    def backward(ctx, g):
        d_img, d_txt, d_s = ctx.saved_tensors
        g = g.detach().reshape(1).float()
        if ctx.flat is not None and g.is_cuda:     # d_image | d_text | d_scale are one buffer: one launch applies the upstream gradient
            out = ops.scale_by_scalar(ctx.flat, g, torch.empty_like(ctx.flat))
            n = d_img.numel()
            return out[:n].view_as(d_img), out[n:2 * n].view_as(d_txt), out[2 * n], None, None
        return d_img * g, d_txt * g, d_s * g, None, None


class _LossBase(torch.nn.Module):
    mode = "clip"
    cap_logit_scale: Optional[float] = None
    temp_reg_weight: float = 0.0
    neighbor_alpha_scale: float = 1.0

    def _init_common(self, local_loss, gather_with_grad, rank, world_size, use_horovod):
        if use_horovod:
            raise NotImplementedError("Horovod is out of scope (SURVEY.md 2.1); use torch.distributed / RCCL")
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.use_horovod = use_horovod
        # rank / world_size ctor kwargs are accepted for signature parity (loss.py:71-78) but the LIVE process group
        # decides at call time: under Lightning the reference constructs its losses before the group exists and
        # therefore never gathers ("distributed quirk"); the intended behaviour is built here
        self.recall_hits: Optional[torch.Tensor] = None
        self.prefetched: Optional[comm.FeatureGather] = None      # set by the module: gathers launched inside the net
        self.last: Dict[str, torch.Tensor] = {}

    @property
    def rank(self) -> int:
        return comm.world()[0]

    @property
    def world_size(self) -> int:
        return comm.world()[1]


class ClipLoss(_LossBase):
    """``ClipLoss(local_loss, gather_with_grad, cache_labels, rank, world_size, use_horovod)``.
    ``local_loss=True``: each rank scores its B rows against the global batch ([B,G], loss.py:116-118);
    ``local_loss=False``: each rank forms the full [G,G] logits (loss.py:119-121) -- W times the work for the same
    optimisation objective, kept for parity with the constructor default.  ``gather_with_grad=False`` detaches the
    remote shards (loss.py:54-63)."""
    mode = "clip"

    def __init__(self, local_loss: bool = False, gather_with_grad: bool = False, cache_labels: bool = False,
                 rank: int = 0, world_size: int = 1, use_horovod: bool = False):
        super().__init__()
        self._init_common(local_loss, gather_with_grad, rank, world_size, use_horovod)
        self.cache_labels = cache_labels

    def forward(self, image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: torch.Tensor,
                logit_bias: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        loss = _ContrastiveFn.apply(image_features, text_features, logit_scale, self, {"logit_bias": logit_bias})
        return {"contrastive_loss": loss}


class SpatialLoss(_LossBase):
    """Multi-positive spatial-neighbour loss: soft labels from the tile-id join, STE-capped temperature,
    temperature regulariser (losses.py:16-42 ctor, :44-124 forward)."""
    mode = "spatial"

    def __init__(self, local_loss: bool = False, gather_with_grad: bool = False, rank: int = 0, world_size: int = 1,
                 use_horovod: bool = False, cap_logit_scale: Optional[float] = None, temp_reg_weight: float = 0.0,
                 float32_logits: bool = False, neighbor_alpha_scale: float = 1.0):
        super().__init__()
        self._init_common(local_loss, gather_with_grad, rank, world_size, use_horovod)
        self.cap_logit_scale = cap_logit_scale
        self.temp_reg_weight = temp_reg_weight
        self.float32_logits = float32_logits        # logits are always fp32 on this path
        self.neighbor_alpha_scale = neighbor_alpha_scale

    def forward(self, image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: torch.Tensor,
                image_tile_ids: torch.Tensor, text_tile_ids: torch.Tensor, neighbor_tile_ids: torch.Tensor,
                neighbor_alphas: torch.Tensor, logit_bias: Optional[torch.Tensor] = None,
                output_dict: bool = True) -> Dict[str, torch.Tensor]:
        kw = {"image_tile_ids": image_tile_ids, "text_tile_ids": text_tile_ids,
              "neighbor_tile_ids": neighbor_tile_ids, "neighbor_alphas": neighbor_alphas, "logit_bias": logit_bias}
        loss = _ContrastiveFn.apply(image_features, text_features, logit_scale, self, kw)
        return {"contrastive_loss": loss}
