#!/usr/bin/env python3
"""sc_png_decode throughput against the number of tiles per launch.  One wave decodes one tile, and a DEFLATE stream is a chain
of dependent scalar steps: a single wave per SIMD (256 tiles = 1 wave on a quarter of the SIMDs) leaves the machine almost
idle, so the cost per tile falls until every SIMD holds its 6 waves (84 VGPRs each; 5.9 KiB of LDS per wave).  The data module
therefore decodes several batches ahead in one launch (shards.decode_png_batch takes any number of files)."""
import io, os, sys
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

B, S = 256, 224
rng = np.random.default_rng(0)
files = []
for b in range(B):
    base = np.asarray(Image.fromarray(rng.integers(0, 256, (28, 28, 3), dtype=np.uint8)).resize((S, S), Image.BICUBIC))
    tile = np.clip(base.astype(int) + rng.integers(-10, 11, (S, S, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    Image.fromarray(tile).save(bio, format="PNG")
    files.append(bio.getvalue())
print(f"tissue-like {S}x{S} tiles, {sum(map(len, files)) / B / 1024:.0f} KiB of PNG each")
for mult in (1, 2, 4, 8, 16, 24, 32):
    fs = files * mult
    lens = np.array([len(f) for f in fs], dtype=np.int64)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
    blob = torch.frombuffer(bytearray(b"".join(fs)), dtype=torch.uint8).cuda()
    out, st = ops.png_decode(blob, offs, S, S)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.png_decode(blob, offs, S, S)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = B * mult
    print(f"{n:5d} tiles per launch: {ms:7.2f} ms = {ms / mult:5.2f} ms per 256 tiles, {n / ms * 1e3 / 1e3:7.1f} k tiles/s, "
          f"{sum(lens) / ms / 1e6:6.2f} GB/s of PNG in, {n * S * S * 3 / ms / 1e6:6.2f} GB/s of pixels out")
    del out, st, blob
