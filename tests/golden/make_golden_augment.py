#!/usr/bin/env python3
"""Fixture for the device augmentation (sc_augment_tiles): the train transform of the reference applied by PIL itself.

The reference builds its train transform with timm's ``create_transform`` (src/open_clip/transform.py:186-204,
``use_timm: true`` in configs/model/spatial_clip.yaml:12-17).  timm / torchvision are not installed in the build
container, but on PIL tiles they do nothing except call PIL -- the calls are spelled out here:
  RandomResizedCropAndInterpolation -> torchvision.transforms.functional.resized_crop = img.crop(box).resize(size, BICUBIC)
  RandomHorizontalFlip              -> img.transpose(FLIP_LEFT_RIGHT)
  ColorJitter(b, c, s) in a random order -> ImageEnhance.Brightness / Contrast / Color (img).enhance(factor)
  ToTensor -> float32 / 255;  Normalize -> (x - mean) / std
The random draws (crop box, factors, order, flip) are inputs of the kernel, so the fixture fixes them.
Run in the build container:  python tests/golden/make_golden_augment.py   (writes tests/golden/augment_pil.npz)"""
import os

import numpy as np
from PIL import Image, ImageEnhance

MEAN = np.array((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)     # src/open_clip/constants.py:1-2
STD = np.array((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)
PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def pil_pipeline(tile: np.ndarray, p: np.ndarray, S: int) -> np.ndarray:
    x0, y0, cw, ch = (int(v) for v in p[:4])
    im = Image.fromarray(tile).crop((x0, y0, x0 + cw, y0 + ch)).resize((S, S), Image.BICUBIC)
    if p[8] > 0.5:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    for op in PERMS[int(p[7])]:
        if op == 0:
            im = ImageEnhance.Brightness(im).enhance(float(p[4]))
        elif op == 1:
            im = ImageEnhance.Contrast(im).enhance(float(p[5]))
        else:
            im = ImageEnhance.Color(im).enhance(float(p[6]))
    x = np.asarray(im, dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def case(rng, B, H, W, S, smooth):
    if smooth:          # low-frequency content (like stained tissue) next to white-noise tiles
        base = rng.uniform(0, 255, size=(B, H // 4 + 2, W // 4 + 2, 3)).astype(np.float32)
        src = np.stack([np.asarray(Image.fromarray(b.astype(np.uint8)).resize((W, H), Image.BILINEAR)) for b in base])
    else:
        src = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    P = np.zeros((B, 12), dtype=np.float32)
    for b in range(B):
        cw, ch = int(rng.integers(max(2, W // 3), W + 1)), int(rng.integers(max(2, H // 3), H + 1))
        P[b, 0:4] = (rng.integers(0, W - cw + 1), rng.integers(0, H - ch + 1), cw, ch)
        P[b, 4:7] = rng.uniform(0.6, 1.4, size=3).astype(np.float32)
        P[b, 7] = b % 6
        P[b, 8] = float(b % 2)
    P[0, 0:4] = (0, 0, W, H)          # full tile, identity factors: Normalize(ToTensor(resize(tile)))
    P[0, 4:7] = 1.0
    out = np.stack([pil_pipeline(src[b], P[b], S) for b in range(B)])
    return src.astype(np.uint8), P, out


def main():
    rng = np.random.default_rng(20251004)
    z = {}
    for name, (B, H, W, S, smooth) in {"up": (6, 40, 52, 64, True), "down": (6, 96, 80, 32, False),
                                        "same": (6, 48, 48, 48, True)}.items():
        src, P, out = case(rng, B, H, W, S, smooth)
        z[f"{name}_src"], z[f"{name}_params"], z[f"{name}_out"], z[f"{name}_S"] = src, P, out, np.int64(S)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "augment_pil.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, {k: v.shape for k, v in z.items()})


if __name__ == "__main__":
    main()
