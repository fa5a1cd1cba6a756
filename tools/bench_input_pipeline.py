#!/usr/bin/env python3
"""Input pipeline of one batch of 256 tiles (224 x 224 RGB PNG files, tissue-like content): device decode (sc_png_decode) +
device augmentation (sc_augment_tiles) against PIL decode + the PIL transform on ONE host core."""
import io, os, sys, time
import numpy as np
import torch
from PIL import Image, ImageEnhance
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops, shards

B, S = 256, 224
rng = np.random.default_rng(0)
files = []
for b in range(B):
    base = np.asarray(Image.fromarray(rng.integers(0, 256, (28, 28, 3), dtype=np.uint8)).resize((S, S), Image.BICUBIC))
    tile = np.clip(base.astype(int) + rng.integers(-10, 11, (S, S, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    Image.fromarray(tile).save(bio, format="PNG")
    files.append(bio.getvalue())
print(f"{B} tiles, {sum(map(len, files)) / B / 1024:.0f} KiB of PNG per tile")
lens = np.array([len(f) for f in files], dtype=np.int64)
offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
blob = torch.frombuffer(bytearray(b"".join(files)), dtype=torch.uint8)
P = shards.draw_aug_params(B, S, S, {"scale": [0.9, 1.0], "ratio": [0.75, 1.333], "color_jitter": 0.2, "use_timm": True},
                           np.random.default_rng(1)).cuda()


def gpu_time(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dblob = blob.cuda()
t_dec = gpu_time(lambda: ops.png_decode(dblob, offs, S, S))
tiles, status = ops.png_decode(dblob, offs, S, S)
assert int(status.abs().sum()) == 0
t_aug = gpu_time(lambda: ops.augment_tiles(tiles, P, S, shards.OPENAI_MEAN, shards.OPENAI_STD))
t0 = time.time(); blob.cuda(); torch.cuda.synchronize(); t_h2d = (time.time() - t0) * 1e3
print(f"device: H2D of the compressed batch {t_h2d:.2f} ms, sc_png_decode {t_dec:.2f} ms, sc_augment_tiles {t_aug:.2f} ms per batch of {B}")
t0 = time.time()
n_host = 32
for f in files[:n_host]:
    im = Image.open(io.BytesIO(f)).convert("RGB")
    im = im.crop((3, 3, 215, 215)).resize((S, S), Image.BICUBIC)
    im = ImageEnhance.Brightness(im).enhance(1.1); im = ImageEnhance.Contrast(im).enhance(0.9); im = ImageEnhance.Color(im).enhance(1.05)
    x = (np.asarray(im, dtype=np.float32) / 255.0 - np.array(shards.OPENAI_MEAN, dtype=np.float32)) / np.array(shards.OPENAI_STD, dtype=np.float32)
t_host = (time.time() - t0) / n_host * 1e3
print(f"host (PIL, one core): {t_host:.2f} ms per tile = {t_host * B:.0f} ms per batch on one core, {t_host * B / 16:.1f} ms on 16")
