"""Synthetic datamodule with the reference's batch contract (``SpatialClipDataModule._collate_fn``,
``src/data/spatial_datamodule.py:110-137``): ``images f32 [B,3,S,S]``, ``texts`` (float gene matrix [B,n_genes] for
the gene tower), ``image_tile_ids`` / ``text_tile_ids`` int64 [B], ``neighbor_tile_ids`` int64 [B,K] (pad -1),
``neighbor_alphas`` f32 [B,K] (pad 0).  Generation follows SURVEY.md section 8d: tiles on a 64-row grid, Moore
neighbourhood, Gaussian distance weights normalised per row; gene counts log1p(Poisson) with ~85-90 % zeros."""
from __future__ import annotations

import math
from typing import Any, Dict, Iterator, Optional

import torch


def make_gene_rates(n_genes: int, seed: int = 7) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.exp(torch.randn(n_genes, generator=g) * 1.5 - 2.0)


def synthetic_batch(B: int, image_size: int, n_genes: int, K: int = 8, step: int = 0, rank: int = 0,
                    world_size: int = 1, gene_rates: Optional[torch.Tensor] = None, seed: int = 1234) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed + 1000 * step + rank)
    images = torch.randn(B, 3, image_size, image_size, generator=g)
    lam = gene_rates if gene_rates is not None else make_gene_rates(n_genes)
    sb = torch.rand(B, 1, generator=g) * 1.5 + 0.5
    genes = torch.log1p(torch.poisson(lam.unsqueeze(0) * sb, generator=g))
    G = B * world_size
    rows = 64 if G >= 64 else max(1, int(math.sqrt(G)))
    cols = (G + rows - 1) // rows
    idx = torch.arange(rank * B, (rank + 1) * B)
    ids = 10_000 + idx
    r, c = idx // cols, idx % cols
    nb = torch.full((B, K), -1, dtype=torch.long)
    al = torch.zeros(B, K)
    offs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)][:K]
    sigma = 1.0
    for k, (dr, dc) in enumerate(offs):
        rr, cc = r + dr, c + dc
        ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols) & (rr * cols + cc < G)
        nb[ok, k] = 10_000 + (rr * cols + cc)[ok]
        al[ok, k] = math.exp(-(dr * dr + dc * dc) / (2 * sigma * sigma))
    al = al / al.sum(1, keepdim=True).clamp_min(1e-12)
    return {"images": images, "texts": genes, "image_tile_ids": ids.clone(), "text_tile_ids": ids.clone(),
            "neighbor_tile_ids": nb, "neighbor_alphas": al}


def synthetic_captions(B: int, context_length: int = 77, vocab_size: int = 49408, seed: int = 0) -> torch.Tensor:
    """int64 [B, context_length] token ids shaped like the reference tokenizer's output (src/open_clip/tokenizer.py:
    <start_of_text> = vocab - 2, words, <end_of_text> = vocab - 1 -- the largest id, which CLIP.encode_text pools at by argmax --
    zero padding).  Ragged lengths: every other row fills the context (a 50-gene sentence of the data pipeline usually does)."""
    g = torch.Generator().manual_seed(seed)
    sot, eot = vocab_size - 2, vocab_size - 1
    t = torch.zeros(B, context_length, dtype=torch.int64)
    for b in range(B):
        n = context_length - 2 if b % 2 == 0 else int(torch.randint(3, context_length - 2, (1,), generator=g))
        t[b, 0] = sot
        t[b, 1:1 + n] = torch.randint(1, sot, (n,), generator=g)
        t[b, 1 + n] = eot
    return t


class SyntheticSpatialDataModule:
    """Constructor kwargs of ``SpatialClipDataModule`` (spatial_datamodule.py:21-31) plus the synthetic knobs."""

    def __init__(self, data_dir: str = "", k_neighbors: int = 8, batch_size: int = 8, num_workers: int = 0,
                 pin_memory: bool = False, dataset_format: str = "synthetic", dataset_format_kwargs: Optional[Dict] = None,
                 splits: Optional[Dict[str, Any]] = None, image_size: int = 224, n_genes: int = 20000,
                 steps_per_epoch: int = 8, val_steps: int = 2):
        self.k_neighbors, self.batch_size = k_neighbors, batch_size
        self.image_size, self.n_genes = image_size, n_genes
        self.steps_per_epoch, self.val_steps = steps_per_epoch, val_steps
        self.preprocess_fn = None
        self.tokenizer = None
        self._rates = None
        self._text_cfg = None       # TextCfg of the net behind ``tokenizer`` when its second tower is the reference's CLIP text tower

    def setup(self, stage: Optional[str] = None) -> None:
        if self.preprocess_fn is None or self.tokenizer is None:       # spatial_datamodule.py:79-80
            raise ValueError("preprocess_fn and tokenizer must be set before setup()")
        self._rates = make_gene_rates(self.n_genes)
        # The ``texts`` slot follows the model's second tower, as the reference's datamodule does by calling the model's own
        # tokenizer (spatial_datamodule.py:79-80,120): BPE-shaped token ids [B, context_length] for the reference's CLIP text
        # tower (model_name ViT-B-32 / ViT-B-16 / ...: configs/model/spatial_clip.yaml:10), the gene matrix for the gene towers.
        net = getattr(self.tokenizer, "__self__", None)
        self._text_cfg = getattr(getattr(net, "cfg", None), "text", None)

    def _loader(self, n: int, offset: int) -> Iterator[Dict[str, torch.Tensor]]:
        import torch.distributed as dist
        rank, W = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        t = self._text_cfg
        for s in range(n):
            b = synthetic_batch(self.batch_size, self.image_size, self.n_genes if t is None else 64, self.k_neighbors, offset + s,
                                rank, W, self._rates if t is None else None)
            if t is not None:
                b["texts"] = synthetic_captions(self.batch_size, t.context_length, t.vocab_size, seed=4321 + 1000 * (offset + s) + rank)
            yield b

    def set_epoch(self, epoch: int) -> None:
        """Trainer.fit calls this before every epoch: each epoch draws new synthetic batches (seed offset)."""
        self._epoch = int(epoch)

    def train_dataloader(self):
        return _Loader(lambda: self._loader(self.steps_per_epoch, getattr(self, "_epoch", 0) * self.steps_per_epoch),
                       self.steps_per_epoch)

    def val_dataloader(self):
        return _Loader(lambda: self._loader(self.val_steps, 10_000), self.val_steps)

    test_dataloader = val_dataloader


class _Loader:
    def __init__(self, factory, n):
        self.factory, self.n = factory, n

    def __iter__(self):
        return self.factory()

    def __len__(self):
        return self.n


def SpatialClipDataModule(*args, **kwargs):
    """Factory behind the reference's ``_target_: src.data.spatial_datamodule.SpatialClipDataModule``: the
    ``dataset_format`` key picks the backend like the reference's ``create_spatial_dataset`` (spatial_datamodule.py:143):
    ``shards_v1`` -> device-side shard pipeline (shards.py), ``synthetic`` -> HEST-shaped synthetic batches."""
    fmt = kwargs.get("dataset_format", "synthetic")
    if fmt == "shards_v1":
        from .shards import ShardedSpatialDataModule
        return ShardedSpatialDataModule(*args, **kwargs)
    if fmt == "synthetic":
        return SyntheticSpatialDataModule(*args, **kwargs)
    raise ValueError(f"dataset_format {fmt!r}: supported are 'shards_v1' and 'synthetic' "
                     "('parquet_v1' is the reference's legacy layout; convert it to shards)")
