"""Device-side contrastive head: similarity GEMMs -> sparse soft labels -> fused CE (+ temperature regulariser)
forward AND backward in one pass (the loss is always differentiated during training, so the gradients w.r.t. the
features are produced while the similarity matrix is still hot).

Reference semantics: open_clip ClipLoss (src/open_clip/loss.py:91-155, local_loss layout) and SpatialLoss
(src/models/components/losses.py:44-124).  Nothing here synchronises with the host."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def contrastive_forward_backward(
        image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: torch.Tensor, *, mode: str = "clip",
        all_image: Optional[torch.Tensor] = None, all_text: Optional[torch.Tensor] = None, rank: int = 0,
        image_tile_ids: Optional[torch.Tensor] = None, text_tile_ids: Optional[torch.Tensor] = None,
        all_image_tile_ids: Optional[torch.Tensor] = None, all_text_tile_ids: Optional[torch.Tensor] = None,
        neighbor_tile_ids: Optional[torch.Tensor] = None, neighbor_alphas: Optional[torch.Tensor] = None,
        cap_logit_scale: Optional[float] = None, temp_reg_weight: float = 0.0, neighbor_alpha_scale: float = 1.0,
        logit_bias: Optional[torch.Tensor] = None, recall_hits: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Returns loss (0-d), d_image/d_text [B,D] (direct terms), d_all_image/d_all_text [G,D] (this rank's
    contribution to EVERY rank's features = the operand of the reduce-scatter that is the autograd of
    torch.distributed.nn.all_gather, loss.py:50-52), d_scale, d_bias, recall_hits (R@1/5/10 hit counters)."""
    f_i = image_features.contiguous().float()
    f_t = text_features.contiguous().float()
    a_i = f_i if all_image is None else all_image.contiguous().float()
    a_t = f_t if all_text is None else all_text.contiguous().float()
    dev = f_i.device
    B, D = f_i.shape
    G = a_i.shape[0]
    if a_t.shape[0] != G or f_t.shape != f_i.shape:
        raise ValueError("feature shapes disagree")
    if (rank + 1) * B > G:
        raise ValueError(f"rank {rank} with local batch {B} does not fit global batch {G}")
    scale = logit_scale.detach().reshape(1).float()
    z = torch.empty((2, B, G), dtype=torch.float32, device=dev)
    ops.sgemm(f_i, D, 1, a_t, D, 1, z[0], G, B, G, D)
    ops.sgemm(f_t, D, 1, a_i, D, 1, z[1], G, B, G, D)

    if mode == "clip":
        nlab = 1
        lab_col = torch.empty((2, B, 1), dtype=torch.int32, device=dev)
        lab_w = torch.empty((2, B, 1), dtype=torch.float32, device=dev)
        ops.onehot_labels(B, rank, lab_col, lab_w)
        cap, w = 0.0, 0.0
    elif mode == "spatial":
        if neighbor_tile_ids is None or neighbor_alphas is None or image_tile_ids is None or text_tile_ids is None:
            raise ValueError("spatial mode needs tile ids, neighbor_tile_ids and neighbor_alphas")
        K = neighbor_tile_ids.shape[1]
        nlab = K + 1
        ids_i = image_tile_ids if all_image_tile_ids is None else all_image_tile_ids
        ids_t = text_tile_ids if all_text_tile_ids is None else all_text_tile_ids
        lab_col = torch.empty((2, B, nlab), dtype=torch.int32, device=dev)
        lab_w = torch.empty((2, B, nlab), dtype=torch.float32, device=dev)
        ops.neighbor_join(ids_i.contiguous(), ids_t.contiguous(), neighbor_tile_ids.contiguous(),
                          neighbor_alphas.contiguous().float(), B, G, K, rank, neighbor_alpha_scale, lab_col, lab_w)
        cap = float(cap_logit_scale) if cap_logit_scale is not None else 0.0
        w = float(temp_reg_weight)
    else:
        raise ValueError(f"unknown contrastive mode {mode!r}")

    bias = None if logit_bias is None else logit_bias.detach().reshape(1).float()
    rowstats = torch.empty((2 * B, 4), dtype=torch.float32, device=dev)
    loss_out = torch.empty(4, dtype=torch.float32, device=dev)
    ops.contrastive_loss_fwd(z, B, G, scale, cap, bias, lab_col, lab_w, nlab, w, rowstats, loss_out)
    if recall_hits is None:
        recall_hits = torch.zeros(3, dtype=torch.int32, device=dev)
    ops.recall_hits(z[0], G, B, rank * B, recall_hits)
    rowgrad = torch.empty((2 * B, 2), dtype=torch.float32, device=dev)
    d_scale = torch.empty(1, dtype=torch.float32, device=dev)
    d_bias = torch.empty(1, dtype=torch.float32, device=dev)
    ops.contrastive_loss_bwd(z, B, G, scale, cap, bias, lab_col, lab_w, nlab, w, rowstats, loss_out, None, rowgrad,
                             d_scale, d_bias)
    d_image = torch.empty((B, D), dtype=torch.float32, device=dev)
    d_text = torch.empty((B, D), dtype=torch.float32, device=dev)
    d_all_image = torch.empty((G, D), dtype=torch.float32, device=dev)
    d_all_text = torch.empty((G, D), dtype=torch.float32, device=dev)
    ops.sgemm(z[0], G, 1, a_t, 1, D, d_image, D, B, D, G)         # dz_it . all_text
    ops.sgemm(z[0], 1, G, f_i, 1, D, d_all_text, D, G, D, B)      # dz_it^T . image
    ops.sgemm(z[1], G, 1, a_i, 1, D, d_text, D, B, D, G)          # dz_ti . all_image
    ops.sgemm(z[1], 1, G, f_t, 1, D, d_all_image, D, G, D, B)     # dz_ti^T . text
    return {"loss": loss_out[0], "gap": loss_out[1], "d_image": d_image, "d_text": d_text,
            "d_all_image": d_all_image, "d_all_text": d_all_text, "d_scale": d_scale[0], "d_bias": d_bias[0],
            "recall_hits": recall_hits}
