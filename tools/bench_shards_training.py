#!/usr/bin/env python3
"""Training steps fed from shards_v1 tars against the same steps on resident synthetic batches: does the
input pipeline -- file reads, H2D of the compressed PNGs, device inflate + augmentation, gene vectors -- keep up with the step?

    python tools/bench_shards_training.py [--tiles 4096] [--batch 256] [--steps 24] [--root /tmp/sc_shards_bench]

Writes a shards_v1 tree of tissue-like 224 x 224 PNG tiles (2 slides) under --root if it is not there yet (a few seconds per
thousand tiles on 8 worker processes), then times ViT-B/16 + gene-MLP steps (ClipLoss, AdamW) three ways: resident synthetic
batches (bench.py's timed region), shards with the producer thread, shards produced inline (SC_DATA_THREAD=0)."""
import argparse
import functools
import io
import json
import multiprocessing as mp
import os
import sys
import tarfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N_GENES, TOPN, PX = 20000, 50, 224


def _slide(args):
    root, sid, tiles, seed = args
    from PIL import Image
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(tiles)))
    os.makedirs(os.path.join(root, sid), exist_ok=True)
    tmp = os.path.join(root, sid, f"{sid}_000000.tar.tmp")
    with tarfile.open(tmp, "w") as tar:
        for i in range(tiles):
            base = np.asarray(Image.fromarray(rng.integers(0, 256, (28, 28, 3), dtype=np.uint8)).resize((PX, PX), Image.BICUBIC))
            tile = np.clip(base.astype(int) + rng.integers(-10, 11, (PX, PX, 3)), 0, 255).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(tile).save(buf, format="PNG")
            sent = " ".join(f"G{j}" for j in rng.choice(N_GENES, TOPN, replace=False))
            meta = json.dumps({"sample_id": sid, "x": (i % side) * 100.0, "y": (i // side) * 100.0})
            for ext, data in (("png", buf.getvalue()), ("txt", sent.encode()), ("json", meta.encode())):
                info = tarfile.TarInfo(name=f"{sid}_{i:05d}.{ext}")
                info.size = len(data)
                tar.addfile(info, io.BytesIO(data))
    os.replace(tmp, tmp[:-4])


def make_shards(root, tiles):
    slides = [f"SLIDE_{s:02d}" for s in range(8)]
    if all(os.path.exists(os.path.join(root, s, f"{s}_000000.tar")) for s in slides):
        return slides
    t0 = time.time()
    with mp.get_context("spawn").Pool(8) as pool:
        pool.map(_slide, [(root, s, tiles // 8, k) for k, s in enumerate(slides)])
    print(f"wrote {tiles} tiles under {root} in {time.time() - t0:.0f} s", flush=True)
    return slides


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--root", default="/tmp/sc_shards_bench")
    args = ap.parse_args()
    slides = make_shards(args.root, args.tiles)

    import torch
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, module, net, optim, streams
    n = net.SpatialClipNet("ViT-B-16-gene", None, n_genes=N_GENES, seed=0)
    m = module.SpatialClipLitModule(
        n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True),
        functools.partial(optim.FusedAdamW, lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=2000))

    class _T:
        max_steps, max_epochs, estimated_stepping_batches = 1_000_000, None, 1_000_000
    m.trainer = _T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.GradBucketReducer(n.store.grad)

    def step(batch, i):
        with streams.chain_stream():
            loss = m.training_step(batch, i)
            loss.backward()
            reducer.finish()
            opt.step(grad_scale=1.0, max_norm=1.0)
            sched.step()
        return loss

    def timed(batches, label, warm=4):
        it = iter(batches)
        for i in range(warm):
            step(next(it), i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for b in it:
            step(b, warm + k)
            k += 1
            if k == args.steps:
                break
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        print(f"{label:58s} {dt * 1e3:7.2f} ms/step  {args.batch / dt:8.0f} pairs/s  ({k} steps)", flush=True)
        del it
        return dt

    rates = data.make_gene_rates(N_GENES)
    syn = [{k: v.cuda() for k, v in data.synthetic_batch(args.batch, PX, N_GENES, 8, s, 0, 1, rates).items()} for s in range(2)]
    t_syn = timed((syn[i % 2] for i in range(10 ** 6)), "resident synthetic batches (bench.py's timed region)")

    genes = [f"G{j}" for j in range(N_GENES)]
    dm = data.SpatialClipDataModule(data_dir=args.root, k_neighbors=8, batch_size=args.batch, dataset_format="shards_v1",
                                    splits={"train": slides}, image_size=PX, gene_vocab=genes,
                                    aug_cfg={"scale": [0.9, 1.0], "ratio": [0.75, 1.333], "color_jitter": 0.2, "use_timm": True},
                                    centers_per_batch=32, max_neighbors_per_center=7)
    dm.preprocess_fn, dm.tokenizer = n.preprocess_train, n.tokenizer
    t0 = time.time()
    dm.setup("fit")
    print(f"index + device KNN of {args.tiles} tiles: {time.time() - t0:.1f} s", flush=True)

    def epochs():
        e = 0
        while True:
            dm.set_epoch(e)
            yield from dm.train_dataloader()
            e += 1
    t_thr = timed(epochs(), f"shards_v1, producer thread, decode_ahead = {dm.decode_ahead}")
    t_thr2 = timed(epochs(), "  again (sentences cached, page cache warm)")
    os.environ["SC_DATA_THREAD"] = "0"
    t_inl = timed(epochs(), "shards_v1, produced inline on the training stream")
    os.environ.pop("SC_DATA_THREAD")
    # the producer alone: how fast can batches be made when nothing consumes them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for b in epochs():
        k += 1
        if k == args.steps:
            break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{'producer alone (no training)':58s} {dt * 1e3:7.2f} ms/batch {args.batch / dt:8.0f} pairs/s", flush=True)
    print(json.dumps({"synthetic_ms": round(t_syn * 1e3, 2), "shards_thread_ms": round(t_thr2 * 1e3, 2),
                      "shards_inline_ms": round(t_inl * 1e3, 2), "producer_alone_ms": round(dt * 1e3, 2), "batch": args.batch,
                      "tiles": args.tiles}))


if __name__ == "__main__":
    main()
