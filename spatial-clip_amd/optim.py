"""Optimiser and LR schedule with the reference's config surface.

``FusedAdamW(params, lr, betas, eps, weight_decay)`` takes the kwargs of ``torch.optim.AdamW``
(configs/optimizer/adamw.yaml) and applies one fused kernel over the flat fp32 buffers of the ParamStore: global
grad-norm clip (Lightning ``gradient_clip_val``), DDP's 1/world_size mean, decoupled weight decay on ALL
parameters (the reference passes ``self.parameters()`` un-grouped, src/models/spatial_clip_module.py:139) and the
bf16 compute-copy refresh.  ``get_cosine_schedule_with_warmup`` reproduces transformers' LambdaLR schedule
(configs/scheduler/cosine.yaml): the first optimiser step runs with lr = 0."""
from __future__ import annotations

import math
import os
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
        self._behind = None         # bucket plan of the replicated optimiser's overlapped form (_step_behind_forward)
        # graph-captured steps (graph.GraphedTrainStep): the step-dependent scalars live in device memory
        self.hyper = None           # fp32 [3] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)}
        self._hyper_host = None     # pinned staging of the same

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

    # ---- replicated optimiser, update hidden behind the next forward (round 5)
    def _behind_plan(self, bucket_floats: int = 16 * 1024 * 1024):
        """Static buckets of the flat buffers in the order the forward consumes them (the second tower runs first and sits at
        the END of the buffers), each with the derived weight copies that are complete once it is."""
        if self._behind is None:
            st = self.store
            size = max(64, (int(bucket_floats) + 63) // 64 * 64)
            edges = list(range(0, st.total, size)) + [st.total]
            if len(edges) > 2 and edges[-1] - edges[-2] < size // 2:
                edges.pop(-2)
            buckets = [(edges[k], edges[k + 1]) for k in range(len(edges) - 1)]
            first_second = min((sp.offset for sp in st.specs if not sp.name.startswith("visual.")), default=0)
            k0 = next(k for k, (lo, hi) in enumerate(buckets) if lo <= first_second < hi)
            order = list(range(k0, len(buckets))) + list(range(0, k0))
            pos = {k: i for i, k in enumerate(order)}
            copies = [[] for _ in buckets]
            for c in st.copies.values():
                sp = st.by_name[c.name]
                touched = [k for k, (lo, hi) in enumerate(buckets) if lo < sp.offset + sp.numel and sp.offset < hi]
                copies[max(touched, key=lambda k: pos[k])].append(c)
            self._behind = (buckets, order, copies, [st._transpose_plan(cs) for cs in copies])
        return self._behind

    def _step_behind_forward(self, grad_scale: float, max_norm: Optional[float]) -> torch.Tensor:
        """The replicated update (one process, or the all-reduce route) bucket by bucket on the communication stream, in the order
        the next forward consumes the buckets: the forward waits per bucket at the first use of its parameters
        (``ParamStore.wait_names``, the mechanism of the sharded optimiser), so all but the first bucket of the HBM-bound update
        (30 bytes per parameter) runs under the matrix-bound start of the next step.  Same kernels on the same values: weights
        bit-identical to the one-launch form (``SC_ADAMW_BEHIND=0``)."""
        from . import streams
        g = self.param_groups[0]
        st = self.store
        st.wait_all()               # a step without a forward in between: the previous update must be complete
        clip = None
        if max_norm is not None and max_norm > 0:
            ops.grad_norm(st.grad, st.total, grad_scale, max_norm, self.norm_clip)
            clip = self.norm_clip
        buckets, order, copies, plans = self._behind_plan(int(os.environ.get("SC_ADAMW_BUCKET", 16 * 1024 * 1024)))
        cur = torch.cuda.current_stream(st.device)
        side = streams.comm_stream(st.device)       # (a low-priority stream of its own measured the same on ViT-B/16, worse on ViT-L/14)
        ready = torch.cuda.Event()
        ready.record(cur)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            for k in order:
                lo, hi = buckets[k]
                ops.adamw_step(st.master[lo:hi], st.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], hi - lo, g["lr"],
                               g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, grad_scale, clip,
                               st.master_bf16[lo:hi])
                st.refresh_range(lo, hi, copies[k], plans[k], fresh=(lo, hi))
                done = torch.cuda.Event()
                done.record(side)
                st.add_pending(lo, hi, done)
        return self.norm_clip

    # ---- graph-captured steps: identical launches every step, step-dependent scalars refreshed from the host
    def refresh_hyper(self) -> None:
        """Write {lr, 1 - beta1^step, sqrt(1 - beta2^step)} for the step about to run (``step_count`` already advanced) into
        the device triple ``self.hyper``: one 12-byte async H2D copy on the current stream, enqueued in front of the graph
        replay that reads it."""
        g = self.param_groups[0]
        if self.hyper is None:
            self.hyper = torch.zeros(3, dtype=torch.float32, device=self.store.device)
            self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory()
        # formed by the library itself, with the expressions of sc_adamw_step: same bits as the eager launch
        from . import _lib
        _lib.check(_lib.lib().sc_adamw_hyper_host(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), int(self.step_count),
                                                  self._hyper_host.data_ptr()), "sc_adamw_hyper_host")
        self.hyper.copy_(self._hyper_host, non_blocking=True)

    def step_captured(self, grad_scale: float = 1.0, max_norm: Optional[float] = None) -> torch.Tensor:
        """The replicated one-launch update in the form a hipGraph can replay: clip coefficient and the step-dependent
        scalars from device memory (``refresh_hyper``), no host-side step counter.  Single process only."""
        if self.exchange is not None:
            raise RuntimeError("step_captured: the sharded optimiser issues collectives; graph capture is single-process")
        g = self.param_groups[0]
        st = self.store
        st.wait_all()
        clip = None
        if max_norm is not None and max_norm > 0:
            ops.grad_norm(st.grad, st.total, grad_scale, max_norm, self.norm_clip)
            clip = self.norm_clip
        ops.adamw_step_dev(st.master, st.grad, self.exp_avg, self.exp_avg_sq, st.total, self.hyper, g["betas"][0],
                           g["betas"][1], g["eps"], g["weight_decay"], grad_scale, clip, st.master_bf16)
        st.refresh_compute_copies(mirror_is_fresh=True)
        return self.norm_clip

    def zero_grad(self, set_to_none: bool = False) -> None:
        """Gradients are overwritten by every backward; nothing to clear."""

    def step(self, grad_scale: float = 1.0, max_norm: Optional[float] = None) -> torch.Tensor:
        g = self.param_groups[0]
        st = self.store
        self.step_count += 1
        if self.exchange is not None:
            return self._step_sharded(grad_scale, max_norm)
        # Default: behind the forward for models up to 200 M parameters.  Measured (same box, interleaved): ViT-B/16 + gene-MLP
        # (97 M) -0.12 ms per step; ViT-L/14 + gene transformer (428 M) +0.3 ms whatever the bucket size or stream priority -- its
        # longer update and its forward contend for HBM longer than the overlap returns.  SC_ADAMW_BEHIND=1 / 0 forces either form.
        behind = os.environ.get("SC_ADAMW_BEHIND", "auto")
        if st.master.is_cuda and (behind == "1" or (behind not in ("0", "1") and st.total <= 200_000_000)):
            return self._step_behind_forward(grad_scale, max_norm)
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
        self.store.wait_all()           # an update may still be running behind the forward (communication stream)
        if self.exchange is not None:
            m, v = self.exchange.gather_moments(self.exp_avg), self.exchange.gather_moments(self.exp_avg_sq)
        else:
            m, v = self.exp_avg.clone(), self.exp_avg_sq.clone()
        return {"step": self.step_count, "exp_avg": m, "exp_avg_sq": v,
                "param_groups": [{k: v_ for k, v_ in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd) -> None:
        self.store.wait_all()
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
