"""Device-side contrastive head: similarity GEMMs -> sparse soft labels -> fused CE (+ temperature regulariser)
forward AND backward in one pass (the loss is always differentiated during training, so the gradients w.r.t. the
features are produced while the similarity matrix is still hot).

Reference semantics: open_clip ClipLoss (src/open_clip/loss.py:91-155, local_loss layout) and SpatialLoss
(src/models/components/losses.py:44-124).  Nothing here synchronises with the host.

The six matrix products (two similarity matrices, four gradient products) are exact-fp32 MFMA GEMMs
(``sc_sgemm_f32_grouped``): one launch for the forward pair, one for the backward four."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import ops


def _rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, D] with unit inner stride (the row stride may exceed D: gathered features sit in the packed
    receive buffer of the all-gather and are read in place)."""
    t = t.float()
    return t if t.stride(1) == 1 else t.contiguous()


def contrastive_forward_backward(
        image_features: torch.Tensor, text_features: torch.Tensor, logit_scale: torch.Tensor, *, mode: str = "clip",
        all_image: Optional[torch.Tensor] = None, all_text: Optional[torch.Tensor] = None, rank: int = 0,
        image_tile_ids: Optional[torch.Tensor] = None, text_tile_ids: Optional[torch.Tensor] = None,
        all_image_tile_ids: Optional[torch.Tensor] = None, all_text_tile_ids: Optional[torch.Tensor] = None,
        neighbor_tile_ids: Optional[torch.Tensor] = None, neighbor_alphas: Optional[torch.Tensor] = None,
        cap_logit_scale: Optional[float] = None, temp_reg_weight: float = 0.0, neighbor_alpha_scale: float = 1.0,
        logit_bias: Optional[torch.Tensor] = None, recall_hits: Optional[torch.Tensor] = None,
        recall_rows: Optional[tuple] = None,
        late_all_image: Optional[Callable[[], torch.Tensor]] = None, join_local: bool = False,
        want_recall: bool = True) -> Dict[str, torch.Tensor]:
    """Returns loss (0-d), d_image/d_text [B,D] (direct terms), d_all [G,2D] = d_all_image | d_all_text (this rank's
    contribution to EVERY rank's features = the operand of the reduce-scatter that is the autograd of
    torch.distributed.nn.all_gather, loss.py:50-52), d_scale, d_bias, recall_hits (R@1/5/10 hit counters).

    ``late_all_image``: callable returning ``all_image`` -- invoked only after the first similarity GEMM
    (image . all_text^T, which does not need it) has been enqueued, so a still-running all-gather of the image
    features overlaps with that GEMM.  ``recall_rows = (row0, n)``: the rows of z[0] whose diagonal block feeds R@k
    (default: all B rows, diagonal at column rank*B).  ``join_local`` (single process, G == B): the gathered-feature
    terms are accumulated straight onto the direct terms by the GEMMs (``d_image`` / ``d_text`` then hold the complete
    feature gradients and no ``d_all`` is formed).  ``want_recall=False`` skips the R@k counters.
    ``grads``: one flat fp32 buffer [2 B D + 1] = d_image | d_text | d_scale (one launch scales all three in backward)."""
    f_i = _rows(image_features)
    f_t = _rows(text_features)
    a_t = f_t if all_text is None else _rows(all_text)
    dev = f_i.device
    B, D = f_i.shape
    G = a_t.shape[0]
    if f_t.shape != f_i.shape:
        raise ValueError("feature shapes disagree")
    if (rank + 1) * B > G:
        raise ValueError(f"rank {rank} with local batch {B} does not fit global batch {G}")
    scale = logit_scale.detach().reshape(1).float()
    z = torch.empty((2, B, G), dtype=torch.float32, device=dev)
    p_it = (f_i, f_i.stride(0), 1, a_t, a_t.stride(0), 1, z[0], G, B, G, D)
    if late_all_image is not None:
        ops.sgemm_grouped([p_it])
        a_i = _rows(late_all_image())
        ops.sgemm_grouped([(f_t, f_t.stride(0), 1, a_i, a_i.stride(0), 1, z[1], G, B, G, D)])
    else:
        a_i = f_i if all_image is None else _rows(all_image)
        ops.sgemm_grouped([p_it, (f_t, f_t.stride(0), 1, a_i, a_i.stride(0), 1, z[1], G, B, G, D)])
    if a_i.shape[0] != G:
        raise ValueError("gathered image / text batches disagree")

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
    if want_recall or recall_hits is not None:
        if recall_hits is None:
            recall_hits = torch.zeros(3, dtype=torch.int32, device=dev)
        if recall_rows is None:
            ops.recall_hits(z[0], G, B, rank * B, recall_hits)
        else:           # a window of a [G,G] matrix: rows row0.. whose diagonal sits at the same column offset
            row0, nrow = recall_rows
            ops.recall_hits(z[0][row0:], G, nrow, row0, recall_hits)
    rowgrad = torch.empty((2 * B, 2), dtype=torch.float32, device=dev)
    grads = torch.empty(2 * B * D + 1, dtype=torch.float32, device=dev)     # d_image | d_text | d_scale
    d_image, d_text = grads[:B * D].view(B, D), grads[B * D:2 * B * D].view(B, D)
    d_scale = grads[2 * B * D:]
    d_bias = torch.empty(1, dtype=torch.float32, device=dev)
    ops.contrastive_loss_bwd(z, B, G, scale, cap, bias, lab_col, lab_w, nlab, w, rowstats, loss_out, None, rowgrad,
                             d_scale, d_bias)
    direct = [(z[0], G, 1, a_t, 1, a_t.stride(0), d_image, D, B, D, G),     # dz_it . all_text        (K = G: first)
              (z[1], G, 1, a_i, 1, a_i.stride(0), d_text, D, B, D, G)]      # dz_ti . all_image
    out = {"loss": loss_out[0], "gap": loss_out[1], "d_image": d_image, "d_text": d_text, "d_scale": d_scale[0],
           "d_bias": d_bias[0], "recall_hits": recall_hits, "grads": grads}
    if join_local:
        if G != B or rank != 0:
            raise ValueError("join_local: only for the single-process head (G == B)")
        ops.sgemm_grouped(direct)
        ops.sgemm_grouped([(z[0], 1, G, f_i, 1, f_i.stride(0), d_text, D, G, D, B, 1),     # += dz_it^T . image = d all_text
                           (z[1], 1, G, f_t, 1, f_t.stride(0), d_image, D, G, D, B, 1)])   # += dz_ti^T . text  = d all_image
        return out
    d_all = torch.empty((G, 2 * D), dtype=torch.float32, device=dev)       # d_all_image | d_all_text
    ops.sgemm_grouped(direct + [
        (z[0], 1, G, f_i, 1, f_i.stride(0), d_all[:, D:], 2 * D, G, D, B),   # dz_it^T . image  -> d all_text
        (z[1], 1, G, f_t, 1, f_t.stride(0), d_all[:, :D], 2 * D, G, D, B),   # dz_ti^T . text   -> d all_image
    ])
    out.update(d_all=d_all, d_all_image=d_all[:, :D], d_all_text=d_all[:, D:])
    return out
