"""Optimiser and LR schedule with the reference's config surface.

``FusedAdamW(params, lr, betas, eps, weight_decay)`` takes the kwargs of ``torch.optim.AdamW``
(configs/optimizer/adamw.yaml) and applies one fused kernel over the flat fp32 buffers of the ParamStore: global
grad-norm clip (Lightning ``gradient_clip_val``), DDP's 1/world_size mean, decoupled weight decay on ALL
parameters (the reference passes ``self.parameters()`` un-grouped, src/models/spatial_clip_module.py:139) and the
bf16 compute-copy refresh.  ``get_cosine_schedule_with_warmup`` reproduces transformers' LambdaLR schedule
(configs/scheduler/cosine.yaml): the first optimiser step runs with lr = 0."""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from . import ops


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        params = list(params)
        stores = {id(getattr(p, "_sc_store", None)): getattr(p, "_sc_store", None) for p in params}
        if len(stores) != 1 or None in stores.values():
            raise ValueError("FusedAdamW needs the parameters of exactly one SpatialClipNet (flat ParamStore)")
        self.store = next(iter(stores.values()))
        if len(params) != len(self.store.specs):
            raise ValueError("FusedAdamW updates the whole flat buffer: pass all parameters (reference passes "
                             "self.parameters() un-grouped)")
        self.param_groups = [{"params": params, "lr": lr, "initial_lr": lr, "betas": tuple(betas), "eps": eps,
                              "weight_decay": weight_decay}]
        n = self.store.total
        dev = self.store.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.norm_clip = torch.zeros(2, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.exchange = None        # comm.ShardedGradExchange when this rank owns 1/W of every gradient bucket

    def attach_exchange(self, exchange) -> None:
        """Data-parallel runs: hand the optimiser the gradient exchange of the process group.  With a
        ``comm.ShardedGradExchange`` the optimiser state shrinks to this rank's 1/W of every bucket (Adam moments: 8 bytes
        per parameter / W) and ``step`` becomes reduce-scattered gradients -> grad-norm (one tiny all-reduce) -> AdamW on the
        shard -> all-gather of the updated masters behind the next forward.  Anything else (None, the all-reduce reducer)
        keeps the replicated optimiser.  Call before ``load_state_dict``."""
        from . import comm
        if not isinstance(exchange, comm.ShardedGradExchange):
            self.exchange = None
            return
        self.exchange = exchange
        n = exchange.shard_floats()
        dev = self.store.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._partials = torch.zeros(1024 * len(exchange.buckets), dtype=torch.float64, device=dev)

    def _step_sharded(self, grad_scale: float, max_norm: Optional[float]) -> torch.Tensor:
        g = self.param_groups[0]
        st, ex = self.store, self.exchange
        st.wait_all()               # a step without a forward in between (tests): the previous gathers must have landed
        clip = None
        if max_norm is not None and max_norm > 0:
            for k in range(len(ex.buckets)):
                a, b = ex.piece(k)
                ops.grad_sumsq_partial(st.grad[a:b], b - a, self._partials[1024 * k:1024 * (k + 1)])
            ex.all_reduce_partials(self._partials)
            ops.grad_norm_final(self._partials, self._partials.numel(), grad_scale, max_norm, self.norm_clip)
            clip = self.norm_clip
        off = {}
        pos = 0
        for k in range(len(ex.buckets)):
            a, b = ex.piece(k)
            off[k] = pos
            pos += b - a
        for k in ex.order:          # in the order the next forward consumes the buckets
            a, b = ex.piece(k)
            m, v = self.exp_avg[off[k]:off[k] + (b - a)], self.exp_avg_sq[off[k]:off[k] + (b - a)]
            ops.adamw_step(st.master[a:b], st.grad[a:b], m, v, b - a, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                           g["weight_decay"], self.step_count, grad_scale, clip, st.master_bf16[a:b])
            ex.gather_bucket(k)
        return self.norm_clip

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Gradients are overwritten by every backward; nothing to clear."""

    def step(self, grad_scale: float = 1.0, max_norm: Optional[float] = None) -> torch.Tensor:
        g = self.param_groups[0]
        st = self.store
        self.step_count += 1
        if self.exchange is not None:
            return self._step_sharded(grad_scale, max_norm)
        clip = None
        if max_norm is not None and max_norm > 0:
            ops.grad_norm(st.grad, st.total, grad_scale, max_norm, self.norm_clip)
            clip = self.norm_clip
        ops.adamw_step(st.master, st.grad, self.exp_avg, self.exp_avg_sq, st.total, g["lr"], g["betas"][0],
                       g["betas"][1], g["eps"], g["weight_decay"], self.step_count, grad_scale, clip, st.master_bf16)
        st.refresh_compute_copies(mirror_is_fresh=True)
        return self.norm_clip

    def state_dict(self):
        """Full flat-layout moments whatever the exchange (a checkpoint written by W ranks resumes on any W').  With the
        sharded exchange this is a COLLECTIVE: every rank must call it (the trainer does, before rank 0 writes the file)."""
        if self.exchange is not None:
            m, v = self.exchange.gather_moments(self.exp_avg), self.exchange.gather_moments(self.exp_avg_sq)
        else:
            m, v = self.exp_avg.clone(), self.exp_avg_sq.clone()
        return {"step": self.step_count, "exp_avg": m, "exp_avg_sq": v,
                "param_groups": [{k: v_ for k, v_ in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd) -> None:
        self.step_count = int(sd["step"])
        n = self.store.total
        m, v = sd["exp_avg"], sd["exp_avg_sq"]
        if m.numel() != n:          # a file written with another padding (other world size): the real parameters come first
            keep = min(m.numel(), n)
            m2, v2 = torch.zeros(n, dtype=m.dtype, device=m.device), torch.zeros(n, dtype=v.dtype, device=v.device)
            m2[:keep], v2[:keep] = m[:keep], v[:keep]
            m, v = m2, v2
        if self.exchange is not None:
            self.exchange.scatter_moments(m.to(self.exp_avg.device), self.exp_avg)
            self.exchange.scatter_moments(v.to(self.exp_avg.device), self.exp_avg_sq)
        else:
            self.exp_avg.copy_(m)
            self.exp_avg_sq.copy_(v)


def cosine_warmup_lambda(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR semantics: lr = initial_lr * f(number of scheduler steps so far)."""

    def __init__(self, optimizer, lr_lambda):
        self.optimizer, self.lr_lambda = optimizer, lr_lambda
        self.last_epoch = 0
        self._apply()

    def _apply(self) -> None:
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self.lr_lambda(self.last_epoch)

    def step(self) -> None:
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                    num_cycles: float = 0.5, last_epoch: int = -1) -> LambdaLR:
    return LambdaLR(optimizer, lambda s: cosine_warmup_lambda(s, num_warmup_steps, num_training_steps, num_cycles))
