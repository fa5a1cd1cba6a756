"""One training step as ONE hipGraph (round 6).

Why: a step of this build is 360 (ViT-B/16 + gene-MLP) to ~750 (ViT-B-32 + CLIP text tower) kernel launches enqueued from
Python at ~25-45 us each.  At BASELINE's headline size the GPU takes longer than the host (31 ms against 17 ms), but the
reference's OWN experiments run ViT-B-32 at batch 32 (configs/experiment/medium_*.yaml, configs/model/spatial_clip.yaml:10):
there the kernels of a step finish in ~3 ms and the step takes the 11.6 ms the host needs to enqueue them
(profiles/r06_bench_*): launch-bound.  The reference pays the same price per ATen call and has no remedy short of a tracing
compiler; here every launch of the step already goes to explicit HIP streams with event hand-offs and no host
synchronisation, so the whole step -- forward, loss, backward (data-gradient chain + weight-gradient side stream as a
fork / join inside the graph), grad-norm clip, AdamW, weight-copy refresh -- is captured once per batch shape and replayed.

What makes the launches identical from step to step:
  * inputs are copied into static device buffers in front of the replay (one D2D copy per batch tensor);
  * the learning rate of the LambdaLR schedule and Adam's bias corrections are read by the AdamW kernel from a device triple
    the host refreshes before each replay (``FusedAdamW.refresh_hyper`` / ``sc_adamw_step_dev``); the clip coefficient was
    device-side already;
  * R@k hit counters and the loss are device tensors; the host-side row counter is advanced by the wrapper.
Same kernels on the same values in the same order: weights are BIT-IDENTICAL to the eager step's
(tests/test_gpu_graph.py).  Not captured: multi-rank steps (RCCL collectives stay eager), the e4m3 path (its amax-history slot
is a host-side step counter), batches of another shape (the ragged last batch of an epoch runs eagerly).

Reference: what Lightning runs per batch -- training_step, backward, clip_grad_norm_, optimizer.step, scheduler.step
(src/models/spatial_clip_module.py:103-108,138-158; configs/trainer/default.yaml:19)."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch

from . import comm, streams


def graph_mode() -> str:
    """SC_GRAPH = auto (default) | 1 | 0.  ``auto``: Trainer.fit / bench.py capture the step once its schedule decisions are
    made and keep the graph when replaying is faster than enqueueing (bench.py measures both; Trainer.fit uses the host-bound
    test of ``GraphedTrainStep.worthwhile``)."""
    return os.environ.get("SC_GRAPH", "auto")


def _signature(batch: Dict[str, Any]):
    return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in batch.items() if isinstance(v, torch.Tensor)))


class GraphedTrainStep:
    """``step = GraphedTrainStep(module, optimizer, max_norm); loss = step(batch)`` -- captured on the first call whose
    preconditions hold, replayed afterwards; ``step.eager(batch)`` is the uncaptured form with the same contract (used for
    batches of another shape and before the schedule decisions are made).  The caller keeps calling ``scheduler.step()``."""

    HOST_BOUND_RATIO = 0.8          # policy "host_bound": capture when enqueueing a step takes >= this share of running it

    def __init__(self, module, optimizer, max_norm: Optional[float] = 1.0, grad_scale: float = 1.0, policy: str = "always"):
        self.m, self.opt = module, optimizer
        self.max_norm, self.grad_scale = max_norm, grad_scale
        # "always": capture as soon as the step can be captured (bench.py times both forms itself; SC_GRAPH=1);
        # "host_bound" (Trainer.fit under SC_GRAPH=auto): only where the eager steps before the capture show the host as the
        # bottleneck -- a GPU-bound step gains nothing from replay and would lose the optimiser's overlap with the next forward
        self.policy = policy
        self._samples = []              # (host enqueue ms, start event, end event) of eager steps
        self._host_ms, self._dev_ms = [], []
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.sig = None
        self._warm_sig = None
        self.static: Dict[str, torch.Tensor] = {}
        self.extra: Dict[str, Any] = {}
        self.loss: Optional[torch.Tensor] = None
        self.rows = 0
        self.replays = 0
        self.failed: Optional[str] = None
        self._capture_stream = None

    # ------------------------------------------------------------------ preconditions
    def capturable(self) -> Optional[str]:
        """None if a step of this module can be captured now, else the reason it cannot."""
        net = self.m.net
        if comm.is_dist():
            return "multi-rank step (RCCL collectives stay eager)"
        if getattr(net, "precision", "bf16") != "bf16":
            return "e4m3 path (amax-history slot is a host-side counter)"
        if getattr(self.opt, "exchange", None) is not None:
            return "sharded optimiser"
        if os.environ.get("SC_OVERLAP", "auto") not in ("0", "1"):
            for name, stack in net._stacks():
                if getattr(stack, "no_side_stream", False):       # runs beside the other tower on its own stream: nothing to decide
                    continue
                st = getattr(stack, "_ov_auto", None)
                if not st or any(s["choice"] is None for s in st.values()):
                    return f"side-stream schedule of the {name} stack not decided yet"
        return None

    # ------------------------------------------------------------------ the two forms of the step
    def eager(self, batch: Dict[str, Any], probe: bool = False) -> torch.Tensor:
        """The enqueued step.  ``probe`` (set by ``__call__`` once the step could be captured, i.e. in steady state: the first
        steps of a process load code objects and run the schedule trials, their device times say nothing): time the host
        enqueue and, with two events, the device side of this step for the host-bound test."""
        import time
        probe = probe and self.policy == "host_bound" and self.graph is None and len(self._host_ms) < 8
        if probe:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
        with streams.chain_stream():
            loss = self.m.training_step(batch, 0)
            loss.backward(self.m.root_gradient(loss))
            self.opt.step(grad_scale=self.grad_scale, max_norm=self.max_norm)
        if self.graph is None:
            self._warm_sig = _signature(batch)      # this shape has now run eagerly (see __call__)
        if probe:
            e1.record()
            self._samples.append(((time.perf_counter() - t0) * 1e3, e0, e1))
            for rec in list(self._samples):         # device times of earlier steps, once they have finished (no host wait)
                if rec[2].query():
                    self._host_ms.append(rec[0])
                    self._dev_ms.append(rec[1].elapsed_time(rec[2]))
                    self._samples.remove(rec)
        return loss

    def worthwhile(self) -> Optional[bool]:
        """policy "host_bound": True / False once >= 3 eager steps have been timed (median host enqueue time against median
        device time of the same steps), None before."""
        if self.policy != "host_bound":
            return True
        if len(self._host_ms) < 3:
            return None
        h, d = sorted(self._host_ms), sorted(self._dev_ms)
        return h[len(h) // 2] >= self.HOST_BOUND_RATIO * d[len(d) // 2]

    def _capture(self, batch: Dict[str, Any]) -> None:
        m, opt = self.m, self.opt
        self.static = {k: v.clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        self.extra = {k: v for k, v in batch.items() if not isinstance(v, torch.Tensor)}
        self.rows = int(self.static["images"].shape[0])
        m.net.store.wait_all()
        opt.step_count += 1
        opt.refresh_hyper()                       # allocates the device triple; the capture only records its address
        opt.step_count -= 1
        m.root_gradient(torch.zeros((), device=m.device))       # resident 1.0 (an ATen fill on first use: outside the capture)
        totals = (m.train_metrics.total,)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # the capture stream stays alive with the graph: scratch buffers are keyed by stream handle (ops.workspace), and a
        # destroyed stream's handle could be handed to a later stream
        pool_stream = self._capture_stream = torch.cuda.Stream(priority=-1)
        pool_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=pool_stream, capture_error_mode=os.environ.get("SC_GRAPH_CAPTURE_MODE", "thread_local")):
            loss = m.training_step({**self.static, **self.extra}, 0)
            loss.backward(m.root_gradient(loss))
            opt.step_captured(grad_scale=self.grad_scale, max_norm=self.max_norm)
        torch.cuda.current_stream().wait_stream(pool_stream)
        (m.train_metrics.total,) = totals         # the capture ran the host side of training_step once without running a kernel
        self.graph, self.loss, self.sig = g, loss.detach(), _signature(batch)

    def __call__(self, batch: Dict[str, Any]) -> torch.Tensor:
        if self.failed is not None:
            return self.eager(batch)
        if self.graph is None:
            why = self.capturable()
            if why is not None:
                return self.eager(batch)
            worth = self.worthwhile()
            if worth is None:
                return self.eager(batch, probe=True)
            if not worth:
                self.failed = "not captured: the eager step is GPU-bound (host enqueue below %.0f %% of the device time)" % (100 * self.HOST_BOUND_RATIO)
                return self.eager(batch)
            if _signature(batch) != self._warm_sig:
                # a shape is captured only right after it has run eagerly: everything a first step creates lazily (activation
                # buffers, workspaces, the R@k counters and their zero fill, side streams) exists before the capture -- an
                # allocation + initialisation recorded INTO the graph would be repeated by every replay
                return self.eager(batch)
            try:
                self._capture(batch)
            except Exception as e:              # a runtime that cannot capture this step: say so once, stay eager
                self.failed = f"{type(e).__name__}: {e}"
                self.graph = None
                import sys
                sys.stderr.write(f"[spatial_clip_amd.graph] capture failed, the step stays eager: {self.failed}\n")
                torch.cuda.synchronize()
                return self.eager(batch)
        elif _signature(batch) != self.sig:
            return self.eager(batch)            # e.g. the ragged last batch of an epoch
        for k, dst in self.static.items():
            src = batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.opt.step_count += 1
        self.opt.refresh_hyper()
        self.graph.replay()
        self.m.train_metrics.total += self.rows
        self.m.logged["train/loss"] = self.loss
        self.replays += 1
        return self.loss
