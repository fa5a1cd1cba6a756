#!/usr/bin/env python3
"""Golden vectors for checkpoint interop: the REFERENCE's ``resize_pos_embed``
(src/open_clip/model.py:792-823) run in the build container on a random positional embedding:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pos_embed.py

Writes tests/golden/pos_embed_resize.npz (input grid 7x7 + class token, outputs for 4x4, 14x14 and the unchanged 7x7)."""
import sys

sys.dont_write_bytecode = True
import os
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    model, _, _ = import_reference()
    g = torch.Generator().manual_seed(11)
    old = torch.randn(1 + 49, 24, generator=g)
    out = {"old": old.numpy()}
    for gs in (4, 7, 14):
        sd = {"visual.positional_embedding": old.clone()}
        fake = types.SimpleNamespace(visual=types.SimpleNamespace(grid_size=(gs, gs)))
        model.resize_pos_embed(sd, fake)
        out[f"grid{gs}"] = sd["visual.positional_embedding"].numpy()
    np.savez(os.path.join(OUT, "pos_embed_resize.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
