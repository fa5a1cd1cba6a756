"""Data-parallel exchange steps of the training step over ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for the world_size-2 correctness tests).  One process per GPU.

Collectives of the path:
  C1+C3  gather_packed     ONE all-gather of image_features | text_features | tile ids packed per row
                           (reference: 2x torch.distributed.nn.all_gather + 2x dist.all_gather,
                           src/open_clip/loss.py:50-52, src/models/components/losses.py:63-68)
  C1'    reduce_scatter_sum  autograd of the feature all-gather: sum over ranks of d(all_features), keep own rows
  C4     GradBucketReducer   bucketed SUM all-reduce of the flat fp32 gradient buffer, launched per layer while
                           backward is still running (RCCL runs on its own stream; the 1/world_size of DDP's mean
                           is folded into the optimiser's grad_scale)
All functions are device-agnostic (CUDA/HIP or CPU tensors) and degrade to no-ops at world_size 1."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_packed(image_features: torch.Tensor, text_features: torch.Tensor,
                  image_tile_ids: Optional[torch.Tensor] = None, text_tile_ids: Optional[torch.Tensor] = None):
    """Rank-major concatenation of every rank's rows: returns (all_image, all_text, all_image_ids, all_text_ids).
    Equal local batch on every rank is assumed, as by the reference (loss.py:96, losses.py:94)."""
    if not is_dist():
        return image_features, text_features, image_tile_ids, text_tile_ids
    _, W = world()
    B, D = image_features.shape
    with_ids = image_tile_ids is not None
    cols = 2 * D + (4 if with_ids else 0)
    packed = torch.empty((B, cols), dtype=torch.float32, device=image_features.device)
    packed[:, :D] = image_features
    packed[:, D:2 * D] = text_features
    if with_ids:
        packed[:, 2 * D:2 * D + 2].view(torch.int64).copy_(image_tile_ids.view(B, 1))
        packed[:, 2 * D + 2:2 * D + 4].view(torch.int64).copy_(text_tile_ids.view(B, 1))
    out = torch.empty((W * B, cols), dtype=torch.float32, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    all_i = out[:, :D].contiguous()
    all_t = out[:, D:2 * D].contiguous()
    if with_ids:
        ids_i = out[:, 2 * D:2 * D + 2].contiguous().view(torch.int64).view(-1)
        ids_t = out[:, 2 * D + 2:2 * D + 4].contiguous().view(torch.int64).view(-1)
        return all_i, all_t, ids_i, ids_t
    return all_i, all_t, None, None


def reduce_scatter_sum(full: torch.Tensor) -> torch.Tensor:
    """full: [W*B, C] per-rank contribution to every rank's rows -> this rank's [B, C] summed over ranks."""
    if not is_dist():
        return full
    rank, W = world()
    B = full.shape[0] // W
    full = full.contiguous()
    if dist.get_backend() == "gloo":          # gloo has no reduce_scatter: all-reduce and slice
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        return full[rank * B:(rank + 1) * B].clone()
    out = torch.empty((B,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM)
    return out


class GradBucketReducer:
    """Overlapped gradient reduction.  ``bucket_ready(lo, hi)`` is called by backward as soon as the flat-gradient
    range [lo, hi) is final; ranges are coalesced until ``bucket_floats`` is reached and then all-reduced
    asynchronously.  ``finish()`` flushes and waits (on the compute stream, no host sync for nccl)."""

    def __init__(self, flat_grad: torch.Tensor, bucket_floats: int = 16 * 1024 * 1024):
        self.flat = flat_grad
        self.bucket_floats = bucket_floats
        self.pending: Optional[Tuple[int, int]] = None
        self.works: List = []
        self.launched: List[Tuple[int, int]] = []

    def bucket_ready(self, lo: int, hi: int) -> None:
        if not is_dist():
            return
        if self.pending is None:
            self.pending = (lo, hi)
        else:
            plo, phi = self.pending
            if hi == plo or lo == phi or (lo <= phi and hi >= plo):      # adjacent / overlapping: merge
                self.pending = (min(lo, plo), max(hi, phi))
            else:
                self._launch(*self.pending)
                self.pending = (lo, hi)
        plo, phi = self.pending
        if phi - plo >= self.bucket_floats:
            self._launch(plo, phi)
            self.pending = None

    def _launch(self, lo: int, hi: int) -> None:
        self.launched.append((lo, hi))
        self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> None:
        if not is_dist():
            return
        if self.pending is not None:
            self._launch(*self.pending)
            self.pending = None
        covered = sum(hi - lo for lo, hi in self.launched)
        if covered < self.flat.numel():
            # anything backward did not announce (e.g. logit_scale): reduce the complement in one go
            done = sorted(self.launched)
            pos = 0
            for lo, hi in done + [(self.flat.numel(), self.flat.numel())]:
                if lo > pos:
                    self.works.append(dist.all_reduce(self.flat[pos:lo], op=dist.ReduceOp.SUM, async_op=True))
                pos = max(pos, hi)
        for w in self.works:
            w.wait()
        self.works, self.launched = [], []
