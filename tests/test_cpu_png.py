"""The PNG / DEFLATE core of the device decoder (spatial-clip_amd/csrc/sc_png_core.h) on the CPU: the same header the HIP
kernel instantiates, compiled by g++ with plain-array IO (oracle/png_core_host.cpp), against PIL -- every block type
(stored, fixed, dynamic Huffman), long codes, multi-chunk IDAT streams, all five scanline filters, RGBA, and damaged /
unsupported files (error codes, never a crash).  The wave-cooperative parts are checked on the GPU (tests/test_gpu_pipeline.py)."""
import ctypes
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def core():
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_png_core_host())
    lib.sc_png_host_decode.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return lib


def png_bytes(arr, **kw):
    bio = io.BytesIO()
    PIL.fromarray(arr).save(bio, format="PNG", **kw)
    return bio.getvalue()


def decode(lib, data, H, W):
    out = np.full((H, W, 3), 7, np.uint8)
    buf = (ctypes.c_ubyte * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    rc = lib.sc_png_host_decode(buf, len(data), out.ctypes.data, H, W)
    return rc, out


def tiles(seed=0, H=224, W=224):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    smooth = np.asarray(PIL.fromarray(rng.integers(0, 256, (30, 30, 3), dtype=np.uint8)).resize((W, H), PIL.BICUBIC))
    tissue = np.clip(smooth.astype(int) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
    flat = np.full((H, W, 3), 200, np.uint8)
    flat[50:100, 30:180] = (120, 40, 160)
    rgba = np.dstack([tissue, rng.integers(0, 256, (H, W), dtype=np.uint8)])
    return {"noise": noise, "smooth": smooth, "tissue": tissue, "flat": flat, "rgba": rgba}


@pytest.mark.parametrize("kw", [{}, {"compress_level": 0}, {"compress_level": 1}, {"compress_level": 9}, {"optimize": True}])
def test_core_matches_pil(core, kw):
    for name, arr in tiles().items():
        data = png_bytes(np.ascontiguousarray(arr), **kw)
        rc, out = decode(core, data, 224, 224)
        want = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
        assert rc == 0 and np.array_equal(out, want), (name, kw, rc)


def test_core_small_and_odd_sizes(core):
    rng = np.random.default_rng(3)
    for H, W in ((1, 1), (3, 5), (17, 64), (65, 33), (130, 7)):
        arr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        arr[: H // 2] = arr[:1]                                   # repeated rows: long matches, the Up filter
        data = png_bytes(arr)
        rc, out = decode(core, data, H, W)
        assert rc == 0 and np.array_equal(out, arr), (H, W, rc)


def test_core_rejects_what_it_cannot_decode(core):
    t = tiles(1)["tissue"]
    good = png_bytes(t)
    assert decode(core, good, 224, 224)[0] == 0
    assert decode(core, good, 128, 128)[0] == 10                  # ERR_SIZE: not the requested tile size
    assert decode(core, good[: len(good) // 2], 224, 224)[0] != 0  # truncated file
    assert decode(core, b"", 224, 224)[0] != 0
    assert decode(core, b"not a png at all, but long enough to carry a header" * 2, 224, 224)[0] == 8
    bad = bytearray(good)
    for i in range(2000, 2400):                                    # garbage inside the compressed stream: an error or wrong
        bad[i] ^= 0x5A                                             # pixels, but never a crash / out-of-bounds write
    rc, out = decode(core, bytes(bad), 224, 224)
    assert rc != 0 or out.shape == (224, 224, 3)
    gray = png_bytes(t[:, :, 0].copy())
    assert decode(core, gray, 224, 224)[0] == 9                    # ERR_UNSUPPORTED: gray / palette / 16-bit / interlaced
    pal = io.BytesIO()
    PIL.fromarray(t).convert("P").save(pal, format="PNG")
    assert decode(core, pal.getvalue(), 224, 224)[0] == 9
