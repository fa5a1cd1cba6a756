"""Neighbour-aware batch sampler.

Spec: ``SpatialBucketBatchSampler`` and ``dataset_build_fast_indices`` of the reference's
``notebooks/test1_loss_test.ipynb`` (cells 5-7): the multi-positive loss only has positives when an anchor's spatial
neighbours sit in the same global batch, so a batch is grown from a few *centre* tiles of one slide plus (a budget of)
their graph neighbours from the same slide, then topped up with further tiles of that slide.  Only index lists are
produced (the dataset's ``__getitem__`` is untouched, no extra IO).  Under data parallelism every rank scans a disjoint
set of slides and all ranks yield the same number of batches per epoch.

Differences from the notebook, both deliberate:
  * slides are dealt to ranks round-robin over the sorted slide ids (stable on every rank, balanced) -- the notebook
    uses Python's per-process salted ``hash(str(sample_id))``, which disagrees between ranks unless PYTHONHASHSEED is
    pinned and can leave a rank without slides;
  * the shuffles really happen (the notebook shuffles a temporary ``order.tolist()`` copy, leaving the order unchanged)."""
from __future__ import annotations

from typing import Dict, Hashable, Iterable, Iterator, List, Mapping, Optional, Sequence

import numpy as np


def build_fast_indices(tile_ids: Sequence[int], sample_ids: Sequence[Hashable],
                       edges_map: Mapping[int, Sequence[int]], k_neighbors: int = 6):
    """-> (id2idx, sample_to_indices, nbr_index[N, K]) : tile id -> dataset index, slide -> its dataset indices, and the
    neighbour table as dataset indices (-1 = missing / neighbour not in the dataset); at most K neighbours per tile, in
    the order the edge list gives them (notebook cell 5)."""
    tile_ids = np.asarray(tile_ids)
    N, K = len(tile_ids), int(k_neighbors)
    id2idx = {int(t): i for i, t in enumerate(tile_ids)}
    buckets: Dict[Hashable, List[int]] = {}
    for i, sid in enumerate(sample_ids):
        buckets.setdefault(sid, []).append(i)
    sample_to_indices = {sid: np.asarray(ix, dtype=np.int64) for sid, ix in buckets.items()}
    nbr_index = np.full((N, K), -1, dtype=np.int64)
    for i in range(N):
        nbrs = list(edges_map.get(int(tile_ids[i]), ()))[:K]
        for k, t in enumerate(nbrs):
            nbr_index[i, k] = id2idx.get(int(t), -1)
    return id2idx, sample_to_indices, nbr_index


class SpatialBucketBatchSampler:
    """``batch_sampler=`` object for a DataLoader (or any loop): iterating yields lists of dataset indices.

    Constructor kwargs as in the notebook: ``batch_size, world_size, rank, centers_per_batch,
    max_neighbors_per_center, same_sample_only, drop_last, seed``.  The dataset must expose ``sample_ids`` [N] plus
    either (``sample_to_indices``, ``nbr_index``) from :func:`build_fast_indices` or (``tile_ids``, ``edges_map``)."""

    def __init__(self, dataset, batch_size: int, world_size: int = 1, rank: int = 0, centers_per_batch: int = 16,
                 max_neighbors_per_center: int = 4, same_sample_only: bool = True, drop_last: bool = True,
                 seed: int = 2025, k_neighbors: Optional[int] = None):
        if not hasattr(dataset, "nbr_index") or not hasattr(dataset, "sample_to_indices"):
            if not (hasattr(dataset, "tile_ids") and hasattr(dataset, "edges_map")):
                raise ValueError("dataset needs (sample_to_indices, nbr_index) or (tile_ids, edges_map)")
            k = k_neighbors or max((len(v) for v in dataset.edges_map.values()), default=1)
            dataset.id2idx, dataset.sample_to_indices, dataset.nbr_index = build_fast_indices(
                dataset.tile_ids, dataset.sample_ids, dataset.edges_map, k)
        self.sample_ids = np.asarray(dataset.sample_ids)
        self.nbr_index = np.asarray(dataset.nbr_index)
        self.buckets: Dict[Hashable, np.ndarray] = dict(dataset.sample_to_indices)
        self.N = len(self.sample_ids)
        self.batch_size, self.world_size, self.rank = int(batch_size), int(world_size), int(rank)
        self.centers_per_batch = int(centers_per_batch)
        self.max_neighbors_per_center = int(max_neighbors_per_center)
        self.same_sample_only, self.drop_last, self.seed = bool(same_sample_only), bool(drop_last), int(seed)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"rank {rank} outside world_size {world_size}")
        ordered = sorted(self.buckets, key=str)
        self.assigned = ordered[self.rank::self.world_size]
        if not self.assigned:
            raise ValueError(f"rank {self.rank} of {self.world_size} owns no slide: fewer slides than ranks?")
        # one global step count so that every rank runs the same number of optimisation steps
        self.batches_per_epoch = max(1, self.N // (self.batch_size * self.world_size))
        self.set_epoch(0)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)
        self._rng = np.random.default_rng([self.seed, self.epoch, self.rank])
        self._order = {s: self._rng.permutation(self.buckets[s]) for s in self.assigned}
        self._ptr = {s: 0 for s in self.assigned}
        self._cycle = [self.assigned[i] for i in self._rng.permutation(len(self.assigned))]

    def __len__(self) -> int:
        return self.batches_per_epoch

    def _next_from(self, s) -> int:
        """Next unused index of slide ``s`` (the slide's order is reshuffled and reused when exhausted)."""
        if self._ptr[s] >= len(self._order[s]):
            self._order[s] = self._rng.permutation(self.buckets[s])
            self._ptr[s] = 0
        i = int(self._order[s][self._ptr[s]])
        self._ptr[s] += 1
        return i

    def __iter__(self) -> Iterator[List[int]]:
        B = self.batch_size
        produced, cursor, guard = 0, 0, 0
        while produced < self.batches_per_epoch:
            s = self._cycle[cursor % len(self._cycle)]
            cursor += 1
            batch: List[int] = []
            used = set()
            n_slide = len(self.buckets[s])
            # 1) centres: the next unused tiles of this slide
            centres = []
            for _ in range(min(self.centers_per_batch, n_slide, B)):
                c = self._next_from(s)
                if c not in used:
                    centres.append(c)
                    used.add(c)
                    batch.append(c)
            # 2) their graph neighbours, in random order, within a total budget of max_neighbors_per_center per centre
            budget = self.max_neighbors_per_center * len(centres)
            taken = 0
            for c in centres:
                if taken >= budget or len(batch) >= B:
                    break
                nbrs = self.nbr_index[c]
                nbrs = nbrs[nbrs >= 0]
                if self.same_sample_only and len(nbrs):
                    nbrs = nbrs[self.sample_ids[nbrs] == s]
                for nb in self._rng.permutation(nbrs):
                    if taken >= budget or len(batch) >= B:
                        break
                    nb = int(nb)
                    if nb not in used:
                        used.add(nb)
                        batch.append(nb)
                        taken += 1
            # 3) top up with further tiles of the same slide (not necessarily neighbours)
            tries = 0
            while len(batch) < B and len(used) < n_slide and tries < 4 * n_slide + B:
                f = self._next_from(s)
                tries += 1
                if f not in used:
                    used.add(f)
                    batch.append(f)
            if len(batch) < B and self.drop_last:
                guard += 1
                if guard > 8 * len(self._cycle) + 8:
                    raise RuntimeError(f"no slide of rank {self.rank} holds batch_size={B} tiles (drop_last=True)")
                continue
            produced += 1
            yield batch


def in_batch_neighbor_rate(batch: Iterable[int], nbr_index: np.ndarray) -> float:
    """Fraction of the valid neighbour entries of a batch's anchors that are themselves in the batch -- the quantity this
    sampler exists to raise (the multi-positive labels only see in-batch neighbours: losses.py:102-108)."""
    idx = np.fromiter(batch, dtype=np.int64)
    nb = nbr_index[idx]
    valid = nb >= 0
    if not valid.any():
        return 0.0
    return float(np.isin(nb[valid], idx).mean())
