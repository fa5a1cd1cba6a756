"""In-batch retrieval metrics with the reference's names: ``ContrastiveMetrics(prefix)`` -> R@1 / R@5 / R@10
(``src/models/components/metrics.py:8-52``).  Hit counting runs on the device (rank of the diagonal among the row)
and is accumulated in int32 counters; ``compute()`` is the only host synchronisation."""
from __future__ import annotations

from typing import Dict

import torch

from . import comm, ops


class ContrastiveMetrics:
    KS = (1, 5, 10)

    def __init__(self, prefix: str):
        self.prefix = prefix
        self.hits = None
        self.total = 0

    def _ensure(self, device) -> None:
        if self.hits is None:
            self.hits = torch.zeros(3, dtype=torch.int32, device=device)

    def update(self, logits: torch.Tensor, target: torch.Tensor = None) -> None:
        """logits [B, B'] with the positive of row i at column i (the module always passes target = arange)."""
        B, G = logits.shape
        self._ensure(logits.device)
        ops.recall_hits(logits.contiguous().float(), G, B, 0, self.hits)
        self.total += B

    def add_hits(self, n: int) -> None:
        """Hits were accumulated into ``self.hits`` by the fused loss; only count the rows."""
        self.total += n

    __call__ = update

    def compute(self) -> Dict[str, float]:
        if self.hits is None or self.total == 0:
            return {f"{self.prefix}R@{k}": float("nan") for k in self.KS}
        hits = self.hits.to(torch.float64)
        total = torch.tensor(float(self.total), dtype=torch.float64, device=hits.device)
        if comm.is_dist():            # dist_reduce_fx="sum" of the reference metric states
            both = torch.cat([hits, total.view(1)])
            torch.distributed.all_reduce(both)
            hits, total = both[:3], both[3]
        vals = (hits / total).tolist()
        return {f"{self.prefix}R@{k}": v for k, v in zip(self.KS, vals)}

    def reset(self) -> None:
        if self.hits is not None:
            self.hits.zero_()
        self.total = 0
