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


def damaged_files(count, H, W, seed=11):
    """Three intact PNG files followed by ``count`` damaged ones: flipped bits, overwritten runs, cuts, hostile chunk
    lengths, valid zlib streams of the wrong length / content, damaged code-length tables."""
    import struct
    import zlib
    rng = np.random.default_rng(seed)
    base = {k: v[:H, :W].copy() for k, v in tiles(2).items()}
    good = [png_bytes(np.ascontiguousarray(a), **kw) for a in base.values() for kw in ({}, {"compress_level": 0}, {"compress_level": 9})]
    files = list(good[:3])
    for k in range(count):
        src = bytearray(good[k % len(good)])
        kind = k % 6
        if kind == 0:                                              # single bit flips inside the compressed stream
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(41, len(src) - 12))
                src[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                            # a run of garbage
            i = int(rng.integers(33, len(src) - 40))
            n = int(rng.integers(1, 64))
            src[i:i + n] = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 2:                                            # cut anywhere
            src = src[:int(rng.integers(0, len(src)))]
        elif kind == 3:                                            # hostile chunk length
            struct.pack_into(">I", src, 33, int(rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, len(src), len(src) - 45])))
        elif kind == 4:                                            # a valid zlib stream of the wrong length / content
            raw = bytes(rng.integers(0, 256, int(rng.integers(0, 3 * H * (3 * W + 1))), dtype=np.uint8))
            z = zlib.compress(raw, int(rng.integers(0, 10)))
            src = bytearray(src[:33] + struct.pack(">I", len(z)) + b"IDAT" + z + struct.pack(">I", zlib.crc32(b"IDAT" + z)) +
                            struct.pack(">I", 0) + b"IEND" + struct.pack(">I", zlib.crc32(b"IEND")))
        else:                                                      # damaged dynamic-block header (code-length tables)
            i = 41 + 2 + int(rng.integers(0, 40))
            src[i] = int(rng.integers(0, 256))
        files.append(bytes(src))
    return files


def test_core_on_damaged_files_under_address_sanitizer(tmp_path):
    """The bit-stream logic the device kernel shares, compiled with -fsanitize=address,undefined, over 240 damaged files
    (flipped bits, overwritten runs, cuts, spliced streams, hostile chunk lengths and code tables): every file ends in a
    status code -- no out-of-bounds read of the file, table or output buffer, no undefined shift, no endless loop."""
    import subprocess
    exe = str(tmp_path / "png_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "oracle", "png_fuzz_main.cpp")], check=True)
    H = W = 64
    files = damaged_files(237, H, W)
    names = []
    for k, f in enumerate(files):
        p = tmp_path / f"f{k:03d}.png"
        p.write_bytes(f)
        names.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(H), str(W)] + names, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    codes = [int(x) for x in r.stdout.split()]
    assert len(codes) == len(files) and codes[:3] == [0, 0, 0]
    assert sum(c != 0 for c in codes[3:]) > 100                   # most damaged files are recognised as such
