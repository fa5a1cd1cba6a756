"""Device side of the input pipeline: sc_knn_alpha against a numpy brute force, sc_augment_tiles
against a plain-torch restatement of RandomResizedCrop / ColorJitter / Normalize, and the shards_v1 datamodule feeding
the trainer end to end."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import ops
    return ops


@pytest.mark.parametrize("N,K", [(1, 3), (5, 8), (200, 6), (3000, 8)])
def test_knn_alpha_vs_bruteforce(N, K):
    ops = _ops()
    rng = np.random.default_rng(N)
    xy = (rng.integers(0, 60, size=(N, 2)) * 2.5).astype(np.float32)          # a grid with ties and duplicates
    nbr, alpha = ops.knn_alpha(torch.from_numpy(xy).cuda(), K)
    nbr, alpha = nbr.cpu().numpy(), alpha.cpu().numpy()
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    for i in range(N):
        order = sorted((j for j in range(N) if j != i), key=lambda j: (d2[i, j], j))[:K]
        want = order + [-1] * (K - len(order))
        assert nbr[i].tolist() == want, (i, nbr[i], want)
        w = np.array([1.0 / (math.sqrt(d2[i, j]) + 1e-6) for j in order], dtype=np.float64)
        if len(order):
            np.testing.assert_allclose(alpha[i, :len(order)], w / w.sum(), rtol=2e-5)
            assert abs(alpha[i].sum() - 1.0) < 1e-5                              # notebook check 1: rows sum to 1
        assert (alpha[i, len(order):] == 0).all()
    nb2, al2 = ops.knn_alpha(torch.from_numpy(xy).cuda(), K, "gaussian", sigma=5.0)
    assert torch.equal(nb2.cpu(), torch.from_numpy(nbr))
    i = N // 2
    order = [j for j in nbr[i] if j >= 0]
    if order:
        w = np.exp(-np.array([d2[i, j] for j in order]) / 50.0)
        np.testing.assert_allclose(al2.cpu().numpy()[i, :len(order)], w / w.sum(), rtol=1e-4, atol=1e-7)


def test_augment_tiles_equals_pil_fixture_and_oracle():
    """sc_augment_tiles against (1) tests/golden/augment_pil.npz -- the reference's train transform applied by PIL itself
    (crop -> antialiased BICUBIC resize -> flip -> ImageEnhance colour jitter -> ToTensor -> Normalize; fixture script
    tests/golden/make_golden_augment.py): upsampling, 3x downsampling (antialias active) and same-size cases, every jitter
    order, flips; BIT-EXACT; (2) the integer CPU restatement oracle/augment_oracle.py (itself pinned to the same fixture in
    tests/test_oracle_golden.py) on 224-pixel tiles with the parameter rows the data module draws."""
    ops = _ops()
    from oracle import augment_oracle as A
    from spatial_clip_amd import shards
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment_pil.npz"))
    for name in ("up", "down", "same"):
        src, P, want, S = z[name + "_src"], z[name + "_params"], z[name + "_out"], int(z[name + "_S"])
        out = ops.augment_tiles(torch.from_numpy(src).cuda(), torch.from_numpy(P).cuda(), S, shards.OPENAI_MEAN,
                                shards.OPENAI_STD).cpu().numpy()
        assert np.array_equal(out, want), (name, float(np.abs(out - want).max()))
    g = torch.Generator().manual_seed(0)
    B, H, W, S = 5, 224, 224, 224
    src = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    P = shards.draw_aug_params(B, H, W, {"scale": [0.9, 1.0], "ratio": [0.75, 1.333], "color_jitter": 0.2, "use_timm": True},
                               np.random.default_rng(3))
    assert set(P[:, 8].tolist()) <= {0.0, 1.0}
    P[1, 8], P[2, 8] = 1.0, 0.0
    out = ops.augment_tiles(src.cuda(), P.cuda(), S, shards.OPENAI_MEAN, shards.OPENAI_STD).cpu().numpy()
    for b in range(B):
        want = A.to_tensor_normalize(A.augment_u8(src[b].numpy(), P[b].numpy(), S), shards.OPENAI_MEAN, shards.OPENAI_STD)
        assert np.array_equal(out[b], want), (b, float(np.abs(out[b] - want).max()))
    # identity parameters at the source size reproduce Normalize(ToTensor(tile)) exactly
    sq = torch.randint(0, 256, (2, 24, 24, 3), generator=g, dtype=torch.uint8)
    Pid = shards.draw_aug_params(2, 24, 24, None, np.random.default_rng(0), train=False)
    o = ops.augment_tiles(sq.cuda(), Pid.cuda(), 24, shards.OPENAI_MEAN, shards.OPENAI_STD).cpu()
    want = (sq.permute(0, 3, 1, 2).float() / 255.0 - torch.tensor(shards.OPENAI_MEAN).view(1, 3, 1, 1)) / \
        torch.tensor(shards.OPENAI_STD).view(1, 3, 1, 1)
    assert torch.equal(o, want)


def test_shards_datamodule_feeds_the_trainer(tmp_path, monkeypatch):
    """shards_v1 tar fixtures (the reference test's layout) -> index -> device KNN / alpha -> neighbour-aware batches ->
    device augmentation -> SpatialLoss training steps on a tiny model.  Checks the batch contract and that the
    neighbour columns really resolve inside the batches."""
    from tests.test_cpu_pipeline import _make_shards
    import functools
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import data, losses, model_configs as mc, module, net, optim, trainer
    root = _make_shards(str(tmp_path / "processed"), slides=2, tiles=36, px=16)
    genes = [f"GENE{i}" for i in range(36)] + ["ACTB", "B2M", "FTL", "MALAT1"]     # the gene-MLP GEMMs want n_genes % 4 == 0
    dm = data.SpatialClipDataModule(data_dir=root, k_neighbors=4, batch_size=12, dataset_format="shards_v1",
                                    splits={"train": ["SAMPLE_A", "SAMPLE_B"], "val": ["SAMPLE_B"]}, image_size=32,
                                    gene_vocab=genes, aug_cfg={"scale": [0.9, 1.0], "color_jitter": 0.2},
                                    centers_per_batch=3, max_neighbors_per_center=3)
    cfg = mc.ModelCfg(embed_dim=32, vision=mc.VisionCfg(32, 8, 64, 2, 32), text=None, gene=mc.GeneCfg(len(genes), 64))
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=2)
    loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                                 neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))
    dm.preprocess_fn, dm.tokenizer = n.preprocess_train, n.tokenizer
    dm.setup("fit")
    batches = list(dm.train_dataloader())
    assert len(batches) == 72 // 12
    hits = 0
    for b in batches:
        assert tuple(b["images"].shape) == (12, 3, 32, 32) and b["images"].is_cuda and b["images"].dtype == torch.float32
        assert tuple(b["texts"].shape) == (12, len(genes)) and b["neighbor_tile_ids"].shape == (12, 4)
        assert torch.equal(b["image_tile_ids"], b["text_tile_ids"]) and len(b["raw_text"]) == 12
        assert torch.allclose(b["neighbor_alphas"].sum(1), torch.ones(12), atol=1e-5)
        ids = set(b["image_tile_ids"].tolist())
        hits += sum(1 for v in b["neighbor_tile_ids"].flatten().tolist() if v in ids)
    assert hits > 0.3 * 12 * 4 * len(batches), hits             # the sampler keeps neighbours in the batch
    # the batches do not depend on how many of them one sc_png_decode launch inflates (decode_ahead: 8 above, 1 and 4 here)
    for ahead in (1, 4):
        dm.decode_ahead = ahead
        again = list(dm.train_dataloader())
        assert len(again) == len(batches)
        for a, b in zip(again, batches):
            assert torch.equal(a["images"], b["images"]) and torch.equal(a["image_tile_ids"], b["image_tile_ids"])
            assert a["raw_text"] == b["raw_text"]
    dm.decode_ahead = 8
    monkeypatch.setenv("SC_DATA_THREAD", "0")                  # ... nor on whether a producer thread / side stream prepares them
    inline = list(dm.train_dataloader())
    monkeypatch.delenv("SC_DATA_THREAD")
    assert all(torch.equal(a["images"], b["images"]) and torch.equal(a["texts"], b["texts"]) for a, b in zip(inline, batches))
    half = []                                                   # a consumer that stops early leaves no thread behind
    for b in dm.train_dataloader():
        half.append(b)
        if len(half) == 2:
            break
    import threading
    import time as _time
    _time.sleep(0.5)
    assert not [t for t in threading.enumerate() if t.name.startswith("sc-data-")]
    # two loaders alive at once (a validation loop inside an epoch): each producer has its own staging buffers
    it_train = iter(dm.train_dataloader())
    first = next(it_train)
    val = list(dm.val_dataloader())
    rest = [first] + list(it_train)
    assert len(val) == 3 and len(rest) == len(batches)
    assert all(torch.equal(a["images"], b["images"]) for a, b in zip(rest, batches))
    from spatial_clip_amd import shards as _sh
    dense = np.stack([_sh.rank_weighted_vector(t, {g: i for i, g in enumerate(genes)}, len(genes)) for t in batches[0]["raw_text"]])
    assert np.array_equal(batches[0]["texts"].cpu().numpy(), dense)
    t = trainer.Trainer(max_epochs=2, gradient_clip_val=1.0, log_every_n_steps=1)
    t.fit(m, dm)
    assert t.global_step == 12 and all(math.isfinite(h["train/loss"]) for h in t.history if "train/loss" in h)
    assert "val/loss" in t.history[-1] and math.isfinite(t.history[-1]["val/loss"])


def test_png_tiles_decoded_on_the_device_equal_pil():
    """sc_png_decode (one wave per tile: inflate + scanline filters on the device) against PIL on 224-pixel tiles of every
    kind the core test covers -- noise (stored-like, two IDAT chunks), smooth / tissue-like (dynamic Huffman, long matches,
    all filters), flat (long overlapping matches), RGBA, compress levels 0 / 1 / 9 -- BIT-EXACT; tiles it must decline
    (wrong size, gray, truncated) come back with a non-zero status and are patched in by the host fallback."""
    import io as _io
    from PIL import Image
    ops = _ops()
    from spatial_clip_amd import shards
    rng = np.random.default_rng(0)
    H = W = 224
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    smooth = np.asarray(Image.fromarray(rng.integers(0, 256, (30, 30, 3), dtype=np.uint8)).resize((W, H), Image.BICUBIC))
    tissue = np.clip(smooth.astype(int) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
    flat = np.full((H, W, 3), 200, np.uint8)
    flat[50:100, 30:180] = (120, 40, 160)
    rgba = np.dstack([tissue, rng.integers(0, 256, (H, W), dtype=np.uint8)])

    def png(arr, **kw):
        bio = _io.BytesIO()
        Image.fromarray(np.ascontiguousarray(arr)).save(bio, format="PNG", **kw)
        return bio.getvalue()
    files = [png(noise), png(smooth), png(tissue), png(flat), png(rgba), png(tissue, compress_level=0),
             png(tissue, compress_level=1), png(tissue, compress_level=9), png(smooth, optimize=True)]
    files += [png(np.clip(tissue.astype(int) + k, 0, 255).astype(np.uint8)) for k in range(23)]          # a fuller batch
    n_good = len(files)
    files += [png(tissue[:100, :100]), png(tissue[:, :, 0]), png(tissue)[:40000]]                        # declined: size, gray, cut
    want = [np.asarray(Image.open(_io.BytesIO(f)).convert("RGB")) for f in files[:n_good]]
    lens = np.array([len(f) for f in files], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    blob = torch.frombuffer(bytearray(b"".join(files)), dtype=torch.uint8).cuda()
    out, status = ops.png_decode(blob, torch.from_numpy(offs).cuda(), H, W)
    st = status.cpu().tolist()
    assert st[:n_good] == [0] * n_good, st
    assert st[n_good] == 10 and st[n_good + 1] == 9 and st[n_good + 2] != 0
    o = out.cpu().numpy()
    for b in range(n_good):
        assert np.array_equal(o[b], want[b]), b
    out2, status2 = ops.png_decode(blob, torch.from_numpy(offs).cuda(), H, W)            # no state carried between launches
    assert torch.equal(out[:n_good], out2[:n_good])
    # the data module's entry point: device decode + host fallback for what the kernel declined
    good_small = png(np.asarray(Image.fromarray(tissue).resize((H, W))))
    batch = shards.decode_png_batch([files[2], files[n_good + 1], good_small], H)
    assert np.array_equal(batch[0].cpu().numpy(), want[2])
    assert np.array_equal(batch[1].cpu().numpy(), np.asarray(Image.open(_io.BytesIO(files[n_good + 1])).convert("RGB")))
    assert batch.shape == (3, H, W, 3)


def test_png_decode_odd_sizes_many_chunks_and_mixed_streams():
    """The wave-parallel literal path of sc_png_decode (64 candidate code positions per round, window slides, re-seated bit
    reader) against zlib / PIL where its bookkeeping is stressed: IDAT payloads cut into many small chunks at arbitrary byte
    positions, literal runs interleaved with matches (a tissue patch tiled over the image), rows that are no multiple of
    anything, every zlib strategy (Huffman-only = literals with long codes, RLE, fixed codes, filtered)."""
    import io as _io
    import struct
    import zlib
    from PIL import Image
    ops = _ops()
    rng = np.random.default_rng(5)

    def hand_png(arr, chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, filt=None):
        h, w, c = arr.shape
        raw = bytearray()
        prev = np.zeros((w, c), np.int32)
        for y in range(h):
            row = arr[y].astype(np.int32)
            ft = (y % 5) if filt is None else filt
            left = np.vstack([np.zeros((1, c), np.int32), row[:-1]])
            ul = np.vstack([np.zeros((1, c), np.int32), prev[:-1]])
            if ft == 0:
                d = row
            elif ft == 1:
                d = row - left
            elif ft == 2:
                d = row - prev
            elif ft == 3:
                d = row - ((left + prev) >> 1)
            else:
                pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
                d = row - pred
            raw += bytes([ft]) + (d & 255).astype(np.uint8).tobytes()
            prev = row
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
        z = co.compress(bytes(raw)) + co.flush()

        def ch(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
        out = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0))
        for i in range(0, len(z), chunk):
            out += ch(b"IDAT", z[i:i + chunk])
        return out + ch(b"IEND", b"")

    for (H, W) in [(96, 96), (53, 37), (16, 16), (7, 300)]:
        small = rng.integers(0, 256, (max(H // 8, 2), max(W // 8, 2), 3), dtype=np.uint8)
        smooth = np.asarray(Image.fromarray(small).resize((W, H), Image.BICUBIC))
        tissue = np.clip(smooth.astype(int) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
        patch = tissue[:max(H // 3, 2), :max(W // 3, 2)]
        tiled = np.tile(patch, (4, 4, 1))[:H, :W]                                    # literal runs between matches
        rgba = np.dstack([tissue, rng.integers(0, 256, (H, W), dtype=np.uint8)])
        files = []
        for arr in (tissue, tiled, rgba):
            n_bytes = len(zlib.compress(arr.tobytes(), 6))
            for chunk in (max(n_bytes // 30 + 1, 7), 997, 1 << 20):                  # <= 32 chunks (the kernel's limit)
                files.append(hand_png(arr, chunk))
            files.append(hand_png(arr, 1 << 20, 6, zlib.Z_HUFFMAN_ONLY))
            files.append(hand_png(arr, 1 << 20, 6, zlib.Z_RLE))
            files.append(hand_png(arr, 1 << 20, 6, zlib.Z_FIXED))
            files.append(hand_png(arr, 1 << 20, 9, zlib.Z_FILTERED, filt=4))
            files.append(hand_png(arr, 1 << 20, 1, filt=0))
        files = [f for f in files if f.count(b"IDAT") <= 32]
        want = [np.asarray(Image.open(_io.BytesIO(f)).convert("RGB")) for f in files]
        lens = np.array([len(f) for f in files], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        blob = torch.frombuffer(bytearray(b"".join(files)), dtype=torch.uint8).cuda()
        out, status = ops.png_decode(blob, torch.from_numpy(offs).cuda(), H, W)
        assert status.cpu().tolist() == [0] * len(files), (H, W, status.cpu().tolist())
        o = out.cpu().numpy()
        for b in range(len(files)):
            assert np.array_equal(o[b], want[b]), (H, W, b)


def test_png_decode_of_damaged_files_agrees_with_the_cpu_build_of_its_core():
    """240 files, 237 of them damaged (tests/test_cpu_png.py::damaged_files; the CPU build of the shared core survives them
    under AddressSanitizer): the kernel reports an error exactly where the CPU build does, returns the same pixels where
    the damage still decodes, and the intact files around them are not disturbed."""
    import ctypes
    import __graft_entry__ as ge
    from tests.test_cpu_png import damaged_files
    ops = _ops()
    H = W = 64
    files = damaged_files(237, H, W)
    host = ctypes.CDLL(ge.build_png_core_host())
    host.sc_png_host_decode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    want_rc, want_px = [], []
    for f in files:
        out = np.zeros((H, W, 3), np.uint8)
        buf = (ctypes.c_ubyte * max(len(f), 1)).from_buffer_copy(f if f else b"\0")
        want_rc.append(host.sc_png_host_decode(buf, len(f), out.ctypes.data, H, W))
        want_px.append(out)
    keep = [k for k, f in enumerate(files) if len(f) > 0]             # an empty file has no bytes to point at
    lens = np.array([len(files[k]) for k in keep], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    blob = torch.frombuffer(bytearray(b"".join(files[k] for k in keep)), dtype=torch.uint8).cuda()
    out, status = ops.png_decode(blob, torch.from_numpy(offs).cuda(), H, W)
    st = status.cpu().tolist()
    o = out.cpu().numpy()
    for j, k in enumerate(keep):
        assert (st[j] == 0) == (want_rc[k] == 0), (k, st[j], want_rc[k])
        if st[j] == 0:
            assert np.array_equal(o[j], want_px[k]), k
    assert st[:3] == [0, 0, 0] and sum(s != 0 for s in st) > 100
