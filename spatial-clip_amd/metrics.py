"""In-batch retrieval metrics with the reference's names: ``ContrastiveMetrics(prefix)`` -> R@1 / R@5 / R@10
(``src/models/components/metrics.py:8-52``).  Hit counting runs on the device (rank of the diagonal among the row)
and is accumulated in int32 counters; ``compute()`` is the only host synchronisation."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
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


class ZeroShotGeneExpressionMetric:
    """``src/metrics/zero_shot.py`` with the same constructor, ``update(preds_logits, captions)`` and ``compute()``.

    The caption -> rank-weighted target vector step is string work and stays on the host (one [B, n_genes] float
    matrix per batch, copied once); the sample-wise Pearson correlation and the two metric states (sum of PCC, count)
    live on the device (``sc_pcc_rows``), ``compute()`` is the only host synchronisation and sums the states over
    ranks like the reference's ``dist_reduce_fx="sum"``."""

    def __init__(self, global_hvg_path: Optional[str] = None, dist_sync_on_step: bool = False):
        self.gene_to_idx: Dict[str, int] = {}
        self.num_global_genes = 0
        if global_hvg_path and os.path.exists(global_hvg_path):
            with open(global_hvg_path, "r") as f:
                genes = [line.strip() for line in f if line.strip()]
            self.gene_to_idx = {gene: i for i, gene in enumerate(genes)}
            self.num_global_genes = len(genes)
        self.state = None                      # device float32 [2]: sum_pcc, total_count

    def _compute_rank_weighted_vector(self, caption_list: List[str], device) -> torch.Tensor:
        t = np.zeros((len(caption_list), self.num_global_genes), dtype=np.float32)
        for i, caption in enumerate(caption_list):
            names = caption.split()
            n = len(names)
            for rank, gene in enumerate(names):
                j = self.gene_to_idx.get(gene)
                if j is not None:
                    t[i, j] = 1.0 - (0.8 * rank / max(n, 1))
        return torch.from_numpy(t).to(device, non_blocking=True)

    def update(self, preds_logits: torch.Tensor, captions: List[str]) -> None:
        if self.num_global_genes == 0:
            return
        if preds_logits.shape != (len(captions), self.num_global_genes):
            raise ValueError(f"preds_logits {tuple(preds_logits.shape)} vs {len(captions)} captions x "
                             f"{self.num_global_genes} genes")
        if self.state is None:
            self.state = torch.zeros(2, dtype=torch.float32, device=preds_logits.device)
        targets = self._compute_rank_weighted_vector(captions, preds_logits.device)
        ops.pcc_rows(preds_logits.float().contiguous(), targets, None, self.state)

    __call__ = update

    def compute(self) -> float:
        if self.state is None:
            return 0.0
        st = self.state.to(torch.float64)
        if comm.is_dist():
            torch.distributed.all_reduce(st)
        s, n = st.tolist()
        return s / n if n > 0 else 0.0

    def reset(self) -> None:
        if self.state is not None:
            self.state.zero_()
