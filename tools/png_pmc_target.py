#!/usr/bin/env python3
"""Target of the PMC passes on sc_png_decode: two launches over N tissue-like tiles (N = argv[1], default 8192)."""
import io, os, sys
import numpy as np
import torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 224
rng = np.random.default_rng(0)
files = []
for b in range(64):
    base = np.asarray(Image.fromarray(rng.integers(0, 256, (28, 28, 3), dtype=np.uint8)).resize((S, S), Image.BICUBIC))
    tile = np.clip(base.astype(int) + rng.integers(-10, 11, (S, S, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    Image.fromarray(tile).save(bio, format="PNG")
    files.append(bio.getvalue())
fs = (files * (N // 64 + 1))[:N]
lens = np.array([len(f) for f in fs], dtype=np.int64)
offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
blob = torch.frombuffer(bytearray(b"".join(fs)), dtype=torch.uint8).cuda()
for _ in range(2):
    out, st = ops.png_decode(blob, offs, S, S)
torch.cuda.synchronize()
assert int(st.abs().sum()) == 0
