"""TEST INFRASTRUCTURE -- CPU restatement (numpy, integer arithmetic) of what the reference's train transform does to a
tile, i.e. of PIL: only tests/ may import this.  Pinned by tests/golden/augment_pil.npz, which PIL itself produced
(tests/golden/make_golden_augment.py); tests/test_oracle_golden.py checks this file against it byte for byte.

Follows, operation for operation:
  * Pillow libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc,
    ImagingResampleVertical_8bpc): what ``img.crop(box).resize(size, BICUBIC)`` computes -- torchvision's
    ``resized_crop`` on a PIL image, reached from src/open_clip/transform.py:186-204 (timm create_transform, bicubic);
  * libImaging/Blend.c + PIL/ImageEnhance.py (Brightness / Contrast / Color) + libImaging/Convert.c (rgb2l): what
    torchvision's ColorJitter does to a PIL image;
  * torchvision ToTensor / Normalize in float32."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size: int, out_size: int):
    """-> list of (xmin, int coefficients) per output position (Resample.c:precompute_coeffs + normalize_coeffs_8bpc)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        k = [(v / ww if ww != 0.0 else v) for v in w]
        kk = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k]
        out.append((xmin, np.asarray(kk, dtype=np.int64)))
    return out


def _clip8(ss):
    return np.clip(ss >> PRECISION_BITS, 0, 255)


def resize_bicubic_u8(img: np.ndarray, S: int) -> np.ndarray:
    """uint8 [h, w, 3] -> uint8 [S, S, 3]: horizontal pass, 8-bit intermediate, vertical pass."""
    h, w, _ = img.shape
    src = img.astype(np.int64)
    tmp = np.empty((h, S, 3), dtype=np.int64)
    for ox, (xmin, kk) in enumerate(coeffs(w, S)):
        tmp[:, ox] = _clip8((1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, xmin:xmin + len(kk)], kk, axes=([1], [0])))
    out = np.empty((S, S, 3), dtype=np.int64)
    for oy, (ymin, kk) in enumerate(coeffs(h, S)):
        out[oy] = _clip8((1 << (PRECISION_BITS - 1)) + np.tensordot(tmp[ymin:ymin + len(kk)], kk, axes=([0], [0])))
    return out.astype(np.uint8)


def luma(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.int64)
    return (19595 * v[..., 0] + 38470 * v[..., 1] + 7471 * v[..., 2] + 0x8000) >> 16


def blend(d: np.ndarray, v: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend(degenerate, image, alpha) on 8-bit values: float32 arithmetic, truncation (Blend.c)."""
    a = np.float32(alpha)
    t = d.astype(np.float32) + a * (v.astype(np.int64) - d.astype(np.int64)).astype(np.float32)
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int64).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int64))).astype(np.uint8)


def augment_u8(tile: np.ndarray, p: np.ndarray, S: int) -> np.ndarray:
    """One sample: uint8 [H, W, 3] + its 12 parameters -> the 8-bit image in front of ToTensor, uint8 [S, S, 3]."""
    x0, y0, cw, ch = (int(v) for v in p[:4])
    v = resize_bicubic_u8(tile[y0:y0 + ch, x0:x0 + cw], S)
    if p[8] > 0.5:
        v = v[:, ::-1]
    for op in PERMS[int(p[7])]:
        if op == 0:
            v = blend(np.zeros_like(v), v, p[4])
        elif op == 1:
            m = int(float(luma(v).sum()) / (S * S) + 0.5)
            v = blend(np.full_like(v, m), v, p[5])
        else:
            v = blend(np.repeat(luma(v)[..., None], 3, axis=-1).astype(np.uint8), v, p[6])
    return np.ascontiguousarray(v)


def to_tensor_normalize(u8: np.ndarray, mean, std) -> np.ndarray:
    x = u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(np.moveaxis(x, -1, -3))
