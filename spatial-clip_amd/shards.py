"""``shards_v1`` input pipeline with the data-dependent work on the device.

Format (reference: docs/data_pipeline.md:19, the fixture of tests/test_spatial_datasets.py:57-75): one directory per
slide under ``data_dir``, holding ``*.tar`` files whose members come in triples ``<base>.png`` (RGB tile), ``<base>.txt``
(gene sentence) and ``<base>.json`` (``{"sample_id", "x", "y", ...}``: tile centroid in slide pixels).

What runs where
  host    tar indexing, PNG inflate (PIL), string handling (gene sentence -> tokens through ``tokenizer``, or ->
          a rank-weighted gene vector for the gene towers), the random draws of the augmentation parameters
  device  K-nearest-neighbour search per slide and the loss weights alpha (``sc_knn_alpha``), RandomResizedCrop +
          PIL-exact antialiased bicubic resize + flip + 8-bit ColorJitter + Normalize of the whole batch in one launch
          (``sc_augment_tiles``; byte-identical to the PIL pipeline the reference runs, tests/golden/augment_pil.npz)
The batch dict is the reference's ``_collate_fn`` contract (src/data/spatial_datamodule.py:110-137): ``images``,
``texts``, ``image_tile_ids`` = ``text_tile_ids``, ``neighbor_tile_ids`` (pad -1), ``neighbor_alphas`` (pad 0), ``raw_text``.
Tile ids are global int64 row indices over (sorted slide ids, member order) like the reference's
(notebooks/d1_dataset_construct_cw.ipynb).  Batches are drawn by the neighbour-aware sampler (sampler.py)."""
from __future__ import annotations

import contextlib
import io
import itertools
import json
import math
import os
import queue
import tarfile
import threading
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import comm, ops, streams
from .sampler import SpatialBucketBatchSampler, build_fast_indices

OPENAI_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_STD = (0.26862954, 0.26130258, 0.27577711)


class ShardIndex:
    """Byte offsets of every (png, txt, json) triple of the selected slides; nothing is decoded at construction."""

    def __init__(self, data_dir: str, sample_ids: Optional[Sequence[str]] = None):
        if not os.path.isdir(data_dir):
            raise FileNotFoundError(f"shards_v1 data_dir {data_dir!r} does not exist")     # spatial_datamodule.py:65-71
        slides = sorted(d for d in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, d)) and not d.startswith("."))
        if sample_ids is not None:
            missing = [s for s in sample_ids if s not in slides]
            if missing:
                raise FileNotFoundError(f"slides {missing} not found under {data_dir}")
            slides = sorted(sample_ids)
        self.entries: List[Dict[str, Any]] = []          # {tar, png:(off,size), txt:(off,size), sample_id, x, y}
        for sid in slides:
            sdir = os.path.join(data_dir, sid)
            for tname in sorted(f for f in os.listdir(sdir) if f.endswith(".tar")):
                tpath = os.path.join(sdir, tname)
                groups: Dict[str, Dict[str, Any]] = {}
                with tarfile.open(tpath, "r") as tar:
                    for m in tar:
                        if not m.isfile():
                            continue
                        base, _, ext = m.name.rpartition(".")
                        g = groups.setdefault(base, {})
                        if ext == "json":
                            g["meta"] = json.loads(tar.extractfile(m).read().decode("utf-8"))
                        elif ext in ("png", "txt"):
                            g[ext] = (m.offset_data, m.size)
                for base in sorted(groups):
                    g = groups[base]
                    if "png" not in g or "txt" not in g or "meta" not in g:
                        continue                         # incomplete triple: skipped, like a failed sample in the reference
                    meta = g["meta"]
                    self.entries.append({"tar": tpath, "png": g["png"], "txt": g["txt"],
                                         "sample_id": str(meta.get("sample_id", sid)),
                                         "x": float(meta["x"]), "y": float(meta["y"])})
        if not self.entries:
            raise FileNotFoundError(f"no complete (png, txt, json) triples under {data_dir}")
        self.sample_ids = np.array([e["sample_id"] for e in self.entries])
        self.tile_ids = np.arange(len(self.entries), dtype=np.int64)
        self.xy = np.array([[e["x"], e["y"]] for e in self.entries], dtype=np.float32)

    def __len__(self) -> int:
        return len(self.entries)

    def read(self, i: int):
        e = self.entries[i]
        fd = self._fd(e["tar"])
        return os.pread(fd, e["png"][1], e["png"][0]), self.text(i)

    # ---- the batch producer's forms: one descriptor per tar for the life of the index, PNG bytes read straight into the
    # (pinned) staging buffer, sentences read once
    # At most MAX_OPEN_TARS descriptors are kept (least recently used closed first): a corpus with more tars than the
    # process's RLIMIT_NOFILE (1024 by default; train / val / test indices each hold their own set) must not die with
    # EMFILE in the middle of an epoch.  The sentence cache is bounded the same way (MAX_CACHED_TEXTS entries).
    MAX_OPEN_TARS = 128
    MAX_CACHED_TEXTS = 1 << 18

    def _fd(self, tar: str) -> int:
        fds = self.__dict__.get("_fds")
        if fds is None:
            import collections
            fds = self.__dict__["_fds"] = collections.OrderedDict()
        fd = fds.get(tar)
        if fd is None:
            while len(fds) >= self.MAX_OPEN_TARS:
                _, old = fds.popitem(last=False)
                os.close(old)
            fd = fds[tar] = os.open(tar, os.O_RDONLY)
        else:
            fds.move_to_end(tar)
        return fd

    def png_size(self, i: int) -> int:
        return self.entries[i]["png"][1]

    def read_png_into(self, i: int, dst: memoryview) -> None:
        e = self.entries[i]
        got = os.preadv(self._fd(e["tar"]), [dst], e["png"][0])
        if got != e["png"][1]:
            raise IOError(f"{e['tar']}: short read of tile {i} ({got} of {e['png'][1]} bytes)")

    def text(self, i: int) -> str:
        cache = self.__dict__.setdefault("_txt", {})
        t = cache.get(i)
        if t is None:
            e = self.entries[i]
            t = os.pread(self._fd(e["tar"]), e["txt"][1], e["txt"][0]).decode("utf-8")
            if len(cache) >= self.MAX_CACHED_TEXTS:
                cache.clear()                  # bounded: a corpus larger than the cap is simply re-read
            cache[i] = t
        return t

    def close(self) -> None:
        for fd in self.__dict__.pop("_fds", {}).values():
            os.close(fd)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_png(png: bytes, size: Optional[int] = None) -> np.ndarray:
    """RGB uint8 [H, W, 3]; PIL does the inflate (host).  ``size``: nearest-exact pre-resize of tiles that were not
    written at the working resolution, so that a batch is one dense [B, H, W, 3] array."""
    from PIL import Image
    im = Image.open(io.BytesIO(png)).convert("RGB")
    if size is not None and im.size != (size, size):
        im = im.resize((size, size), Image.BILINEAR)
    return np.array(im, dtype=np.uint8)


def decode_png_batch(pngs: Sequence[bytes], size: int, device=None) -> torch.Tensor:
    """PNG files -> device uint8 [B, size, size, 3].  The compressed bytes go to the GPU as they are and ``sc_png_decode``
    inflates / unfilters them there (one wave per tile); tiles it declines (status != 0: another size, palette / gray /
    16-bit / interlaced files, a damaged stream) are decoded by PIL on the host and patched in."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    lens = np.fromiter((len(p) for p in pngs), dtype=np.int64, count=len(pngs))
    offsets = np.zeros(len(pngs) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    blob = torch.frombuffer(bytearray(b"".join(pngs)), dtype=torch.uint8)
    tiles, status = ops.png_decode(blob.to(device), torch.from_numpy(offsets).to(device), size, size)
    bad = torch.nonzero(status != 0).flatten().cpu().tolist()
    for b in bad:
        tiles[b] = torch.from_numpy(decode_png(pngs[b], size)).to(device)
    return tiles


def neighbor_tables(index: ShardIndex, k_neighbors: int, mode: str = "inverse", device=None):
    """Per-slide KNN + alpha on the device -> (neighbor_tile_ids int64 [N,K] pad -1, neighbor_alphas f32 [N,K] pad 0) in
    GLOBAL tile ids, and the edges map for the sampler."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    N, K = len(index), int(k_neighbors)
    nbr_ids = torch.full((N, K), -1, dtype=torch.int64)
    alphas = torch.zeros((N, K), dtype=torch.float32)
    for sid in np.unique(index.sample_ids):
        rows = np.nonzero(index.sample_ids == sid)[0]
        xy = torch.from_numpy(index.xy[rows]).to(device).contiguous()
        loc, al = ops.knn_alpha(xy, K, mode)
        loc, al = loc.cpu().long(), al.cpu()
        glob = torch.where(loc >= 0, torch.from_numpy(index.tile_ids[rows])[loc.clamp_min(0)], torch.full_like(loc, -1))
        nbr_ids[rows] = glob
        alphas[rows] = al
    edges_map = {int(t): [int(v) for v in nbr_ids[i].tolist() if v >= 0] for i, t in enumerate(index.tile_ids)}
    return nbr_ids, alphas, edges_map


def draw_aug_params(B: int, H: int, W: int, aug_cfg: Optional[Dict[str, Any]], rng: np.random.Generator,
                    train: bool = True) -> torch.Tensor:
    """One parameter row per sample for ``sc_augment_tiles``.  Training: torchvision RandomResizedCrop semantics
    (area fraction ~ U(scale), log-uniform aspect ratio, 10 attempts then centre crop) and ColorJitter factors
    ~ U(1 - j, 1 + j) in a random order (timm's ``color_jitter`` scalar -> brightness = contrast = saturation = j, no
    hue).  Evaluation: the full tile, no jitter."""
    P = np.zeros((B, 12), dtype=np.float32)
    P[:, 4:7] = 1.0
    P[:, 2], P[:, 3] = W, H
    if not train or not aug_cfg:
        return torch.from_numpy(P)
    scale = tuple(aug_cfg.get("scale", (0.9, 1.0)))
    ratio = tuple(aug_cfg.get("ratio", (0.75, 1.3333)))
    j = aug_cfg.get("color_jitter", 0.0) or 0.0
    j = float(j[0]) if isinstance(j, (list, tuple)) else float(j)
    # timm's create_transform flips with its default hflip = 0.5 (the reference passes no hflip, transform.py:186-204);
    # the torchvision fallback branch (use_timm false) has no flip
    flip_p = float(aug_cfg.get("hflip", 0.5 if aug_cfg.get("use_timm") else 0.0))
    for b in range(B):
        cw, ch, x0, y0 = W, H, 0.0, 0.0
        for _ in range(10):
            area = H * W * rng.uniform(*scale)
            ar = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
            w_, h_ = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
            if 0 < w_ <= W and 0 < h_ <= H:
                cw, ch = w_, h_
                x0, y0 = float(rng.integers(0, W - w_ + 1)), float(rng.integers(0, H - h_ + 1))
                break
        else:
            x0, y0 = (W - cw) / 2, (H - ch) / 2
        P[b, 0:4] = (x0, y0, cw, ch)
        if flip_p > 0:
            P[b, 8] = float(rng.uniform() < flip_p)
        if j > 0:
            P[b, 4:7] = rng.uniform(max(0.0, 1 - j), 1 + j, size=3)
            P[b, 7] = float(rng.integers(0, 6))
    return torch.from_numpy(P)


def rank_weighted_vector(sentence: str, gene_to_idx: Dict[str, int], n_genes: int) -> np.ndarray:
    """Gene sentence ("top-N gene symbols, most expressed first") -> dense [n_genes] vector with weight
    1 - rank / N at each listed gene: the target construction of src/metrics/zero_shot.py:41-60, reused as the gene
    towers' input when only sentences are stored."""
    v = np.zeros(n_genes, dtype=np.float32)
    genes = [g for g in sentence.split() if g in gene_to_idx]
    for r, g in enumerate(genes):
        v[gene_to_idx[g]] = max(v[gene_to_idx[g]], 1.0 - r / max(len(genes), 1))
    return v


def rank_weighted_sparse(sentence: str, gene_to_idx: Dict[str, int]):
    """The non-zeros of ``rank_weighted_vector``: (gene indices int64 [n], weights f32 [n]), each gene once (its first =
    heaviest occurrence)."""
    genes = [g for g in sentence.split() if g in gene_to_idx]
    first: Dict[int, float] = {}
    for r, g in enumerate(genes):
        first.setdefault(gene_to_idx[g], 1.0 - r / max(len(genes), 1))
    return (np.fromiter(first.keys(), dtype=np.int64, count=len(first)),
            np.fromiter(first.values(), dtype=np.float32, count=len(first)))


class _PinnedRing:
    """Two pinned staging buffers used in turn; a buffer is handed out again only after the copy that last read it is done."""

    def __init__(self):
        self.buf = [None, None]
        self.done = [None, None]
        self.k = 0

    def take(self, nbytes: int) -> torch.Tensor:
        self.k ^= 1
        if self.done[self.k] is not None:
            self.done[self.k].synchronize()
        b = self.buf[self.k]
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8)
            if torch.cuda.is_available():
                b = b.pin_memory()
            self.buf[self.k] = b
        return b

    def copied(self, stream) -> None:
        ev = torch.cuda.Event()
        ev.record(stream)
        self.done[self.k] = ev


_END = object()
_POOL_LOCK = threading.Lock()


class ShardedSpatialDataModule:
    """``SpatialClipDataModule`` constructor kwargs (spatial_datamodule.py:21-31) with ``dataset_format="shards_v1"``;
    ``splits`` maps "train" / "val" / "test" to lists of slide ids."""

    def __init__(self, data_dir: str = "", k_neighbors: int = 8, batch_size: int = 8, num_workers: int = 0,
                 pin_memory: bool = False, dataset_format: str = "shards_v1",
                 dataset_format_kwargs: Optional[Dict[str, Any]] = None, splits: Optional[Dict[str, Any]] = None,
                 image_size: int = 224, n_genes: Optional[int] = None, gene_vocab: Optional[Sequence[str]] = None,
                 aug_cfg: Optional[Dict[str, Any]] = None, alpha_mode: str = "inverse", seed: int = 2025,
                 centers_per_batch: int = 16, max_neighbors_per_center: int = 4, decode_ahead: int = 8):
        if dataset_format != "shards_v1":
            raise ValueError(f"dataset_format {dataset_format!r}: this module reads 'shards_v1' "
                             "(synthetic batches: data.SyntheticSpatialDataModule)")
        self.data_dir, self.k_neighbors, self.batch_size = data_dir, int(k_neighbors), int(batch_size)
        self.splits = dict(splits or {})
        self.image_size, self.aug_cfg, self.alpha_mode, self.seed = int(image_size), aug_cfg, alpha_mode, int(seed)
        self.n_genes = n_genes
        self.gene_to_idx = {g: i for i, g in enumerate(gene_vocab)} if gene_vocab else None
        self.centers_per_batch, self.max_neighbors_per_center = centers_per_batch, max_neighbors_per_center
        # batches whose PNG files are inflated by ONE sc_png_decode launch: a tile is one wave of dependent scalar steps, so
        # the cost per tile falls 7x between 256 and 8192 tiles per launch (profiles/r03_png_batch_scaling.txt)
        self.decode_ahead = max(1, int(decode_ahead))
        self.preprocess_fn: Optional[Callable] = None
        self.tokenizer: Optional[Callable] = None
        self._sets: Dict[str, Dict[str, Any]] = {}

    def setup(self, stage: Optional[str] = None) -> None:
        if self.preprocess_fn is None or self.tokenizer is None:       # spatial_datamodule.py:79-80
            raise ValueError("preprocess_fn and tokenizer must be set before setup()")
        for name in ("train", "val", "test"):
            if name in self._sets or (self.splits and name not in self.splits):
                continue
            index = ShardIndex(self.data_dir, self.splits.get(name))
            nbr, al, edges = neighbor_tables(index, self.k_neighbors, self.alpha_mode)
            index.edges_map = edges
            index.id2idx, index.sample_to_indices, index.nbr_index = build_fast_indices(
                index.tile_ids, index.sample_ids, edges, self.k_neighbors)
            self._sets[name] = {"index": index, "nbr": nbr, "alpha": al}

    def _texts(self, sentences: List[str]) -> torch.Tensor:
        if self.gene_to_idx is not None:        # gene towers: float [B, n_genes]
            n = self.n_genes or len(self.gene_to_idx)
            return torch.from_numpy(np.stack([rank_weighted_vector(s, self.gene_to_idx, n) for s in sentences]))
        return self.tokenizer(sentences)          # reference text tower: int64 [B, 77]

    def set_epoch(self, epoch: int) -> None:
        """Called by ``Trainer.fit`` before each epoch's iteration: reshuffles the bucket sampler and draws fresh crop /
        colour-jitter parameters (the reference: ``DataLoader(shuffle=True)`` + per-sample random transforms,
        src/data/spatial_datamodule.py:91-101).  Deterministic per (seed, epoch, rank)."""
        self._epoch = int(epoch)

    def _eval_index_batches(self, n: int, bs: int, rank: int, W: int) -> List[List[int]]:
        """Evaluation splits are read sequentially, every sample exactly once (``shuffle=False``,
        spatial_datamodule.py:103-108), the last partial batch kept.  With W ranks: what Lightning's DistributedSampler
        does for ``shuffle=False`` -- the index list is padded by wrap-around to a multiple of W and rank r takes
        elements r, r + W, ...: every rank gets the same number of batches of the same sizes, which the gathered loss
        needs."""
        order = list(range(n))
        if W > 1:
            total = ((n + W - 1) // W) * W
            order = (order + order[:total - n])[rank::W] if total > n else order[rank::W]
        return [order[i:i + bs] for i in range(0, len(order), bs)]

    def _texts_of(self, st: Dict[str, Any], flat: Sequence[int], dev) -> torch.Tensor:
        """The second tower's input for tiles ``flat``, built on the device.  A tile's sentence never changes, so it is read
        and tokenised once (gene towers: the non-zeros of its rank-weighted vector; text tower: its BPE ids) -- the dense
        [B, n_genes] array of a batch is a scatter of ~50 values per row, not 20 000-float rows built in Python."""
        index: ShardIndex = st["index"]
        cache = st.setdefault("tok", {})
        if self.gene_to_idx is not None:
            n = self.n_genes or len(self.gene_to_idx)
            rows, cols, ws = [], [], []
            for r, i in enumerate(flat):
                t = cache.get(i)
                if t is None:
                    t = cache[i] = rank_weighted_sparse(index.text(i), self.gene_to_idx)
                rows.append(np.full(len(t[0]), r, dtype=np.int64))
                cols.append(t[0])
                ws.append(t[1])
            out = torch.zeros((len(flat), n), dtype=torch.float32, device=dev)
            if rows:
                out.index_put_((torch.from_numpy(np.concatenate(rows)).to(dev), torch.from_numpy(np.concatenate(cols)).to(dev)),
                               torch.from_numpy(np.concatenate(ws)).to(dev))
            return out
        toks = []
        for i in flat:
            t = cache.get(i)
            if t is None:
                t = cache[i] = self.tokenizer([index.text(i)])[0]
            toks.append(t)
        return torch.stack(toks).to(dev)

    def _produce(self, name: str, train: bool, dev, stream) -> Iterator[Any]:
        """(batch, event) pairs in training order.  ``decode_ahead`` batches form a group: their PNG files are read straight
        into a pinned buffer, copied to HBM in one piece and inflated by ONE ``sc_png_decode`` launch; augmentation and the
        second tower's input follow per batch.  All device work is enqueued on ``stream`` (None: the caller's), and the event
        marks the point where the batch is complete."""
        st = self._sets[name]
        index: ShardIndex = st["index"]
        rank, W = comm.world()
        epoch = getattr(self, "_epoch", 0)
        if train:
            bs = min(self.batch_size, max(1, len(index) // max(W, 1)))
            sampler = SpatialBucketBatchSampler(index, bs, W, rank, self.centers_per_batch, self.max_neighbors_per_center,
                                                drop_last=True, seed=self.seed)
            sampler.set_epoch(epoch)
        else:
            sampler = self._eval_index_batches(len(index), self.batch_size, rank, W)
        rng = np.random.default_rng([self.seed, epoch, rank, 0 if train else 1])
        host_decode = os.environ.get("SC_PNG_HOST", "0") == "1"      # A/B: PIL on the host, as the reference's workers do
        # staging buffers: one ring per producer that is alive (a validation loop inside an epoch runs while the training
        # producer is parked on its full queue); rings are recycled, pinned memory is not allocated per epoch
        pool = self.__dict__.setdefault("_ring_pool", [])
        with _POOL_LOCK:
            ring = pool.pop() if pool else _PinnedRing()
        try:
            yield from self._produce_groups(st, index, sampler, rng, train, dev, stream, ring, host_decode)
        finally:
            with _POOL_LOCK:
                pool.append(ring)

    def _produce_groups(self, st, index, sampler, rng, train, dev, stream, ring, host_decode) -> Iterator[Any]:
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        it = iter(sampler)
        while True:
            group = list(itertools.islice(it, self.decode_ahead))
            if not group:
                return
            flat = [i for idx in group for i in idx]
            sizes = np.fromiter((index.png_size(i) for i in flat), dtype=np.int64, count=len(flat))
            offsets = np.zeros(len(flat) + 1, dtype=np.int64)
            np.cumsum(sizes, out=offsets[1:])
            total = int(offsets[-1])
            host = ring.take(total)
            mv = memoryview(host.numpy())
            for j, i in enumerate(flat):
                index.read_png_into(i, mv[offsets[j]:offsets[j + 1]])
            with ctx:
                cur = torch.cuda.current_stream()
                if host_decode:
                    tiles_all = torch.from_numpy(np.stack([decode_png(bytes(mv[offsets[j]:offsets[j + 1]]), self.image_size)
                                                           for j in range(len(flat))])).to(dev)
                else:
                    blob = host[:total].to(dev, non_blocking=True)
                    ring.copied(cur)
                    tiles_all, status = ops.png_decode(blob, torch.from_numpy(offsets).to(dev), self.image_size, self.image_size)
                    # tiles the kernel declined go to PIL before the augmentation reads them: one readback per GROUP, on the
                    # producer thread and its own stream (the training thread never waits for it)
                    for j in torch.nonzero(status != 0).flatten().cpu().tolist():
                        tiles_all[j] = torch.from_numpy(decode_png(bytes(mv[offsets[j]:offsets[j + 1]]), self.image_size)).to(dev)
                at = 0
                for idx in group:
                    rows = flat[at:at + len(idx)]
                    tiles = tiles_all[at:at + len(idx)]
                    at += len(idx)
                    params = draw_aug_params(len(idx), tiles.shape[1], tiles.shape[2], self.aug_cfg, rng, train)
                    images = ops.augment_tiles(tiles, params.to(dev), self.image_size, OPENAI_MEAN, OPENAI_STD)
                    ids = torch.from_numpy(index.tile_ids[np.asarray(idx)])
                    batch = {"images": images, "texts": self._texts_of(st, rows, dev), "image_tile_ids": ids,
                             "text_tile_ids": ids.clone(), "neighbor_tile_ids": st["nbr"][np.asarray(idx)],
                             "neighbor_alphas": st["alpha"][np.asarray(idx)], "raw_text": [index.text(i) for i in rows]}
                    ev = None
                    if stream is not None:
                        ev = torch.cuda.Event()
                        ev.record(cur)
                    yield batch, ev

    def _batches(self, name: str, train: bool) -> Iterator[Dict[str, Any]]:
        """Batches of split ``name``.  A producer thread runs ``_produce`` one group ahead on its own HIP stream, so file
        reads, the H2D copy, PNG inflation and augmentation of the next steps overlap the training step that is running
        (the reference's counterpart: DataLoader worker processes + pinned-memory thread, spatial_datamodule.py:91-101);
        ``SC_DATA_THREAD=0`` produces inline on the caller's stream."""
        dev = torch.device("cuda", torch.cuda.current_device())
        if os.environ.get("SC_DATA_THREAD", "1") == "0":
            for batch, _ in self._produce(name, train, dev, None):
                yield batch
            return
        # one producer stream PER SPLIT: an in-epoch validation must not queue behind the decode / augmentation work the
        # training producer has already enqueued (up to decode_ahead batches) on a shared stream (advisor, round 3)
        sides = self.__dict__.setdefault("_side_streams", {})
        side = sides.get(name)
        if side is None:
            side = sides[name] = torch.cuda.Stream(device=dev)
        q: "queue.Queue[Any]" = queue.Queue(maxsize=2 * self.decode_ahead)
        stop = threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                torch.cuda.set_device(dev)
                for item in self._produce(name, train, dev, side):
                    if not put(item):
                        return
                put(_END)
            except BaseException as e:          # noqa: BLE001 -- handed to the consumer, which re-raises it
                put(e)

        th = threading.Thread(target=work, name=f"sc-data-{name}", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                users = streams.consumer_streams()
                users[0].wait_event(ev)
                for v in batch.values():        # allocated on the producer's stream, read on the consumer's
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        for u in users:
                            v.record_stream(u)
                yield batch
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)

    def _loader(self, name: str, train: bool):
        if name not in self._sets:
            raise ValueError(f"split {name!r} was not set up (splits = {list(self.splits)})")
        rank, W = comm.world()
        n_items = len(self._sets[name]["index"])
        if train:
            n = max(1, n_items // max(self.batch_size * W, 1))
        else:
            n = len(self._eval_index_batches(n_items, self.batch_size, rank, W))
        return _Loader(lambda: self._batches(name, train), n)

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("val" if "val" in self._sets else "train", False)

    def test_dataloader(self):
        return self._loader("test" if "test" in self._sets else ("val" if "val" in self._sets else "train"), False)


class _Loader:
    def __init__(self, factory, n):
        self.factory, self.n = factory, n

    def __iter__(self):
        return self.factory()

    def __len__(self):
        return self.n
