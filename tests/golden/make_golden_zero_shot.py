#!/usr/bin/env python3
"""Golden vectors for the validation-only zero-shot gene-expression metric, produced by running
the REFERENCE's own ``ZeroShotGeneExpressionMetric`` (src/metrics/zero_shot.py) in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_zero_shot.py

torchmetrics is not installed here; the class only uses ``Metric.__init__`` / ``add_state``, so a 10-line base class
with those two methods stands in for the package at import time (nothing of the metric's arithmetic is replaced).
Writes tests/golden/zero_shot_metric.npz + zero_shot_metric.json (inputs and expected outputs only)."""
import sys

sys.dont_write_bytecode = True
import importlib.util
import json
import os
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_metric():
    tm = types.ModuleType("torchmetrics")

    class Metric(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default.clone())

    tm.Metric = Metric
    sys.modules["torchmetrics"] = tm
    spec = importlib.util.spec_from_file_location("ref_zero_shot", f"{REF}/src/metrics/zero_shot.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ZeroShotGeneExpressionMetric


def main():
    Metric = import_metric()
    g = torch.Generator().manual_seed(7)
    genes = [f"GENE{i}" for i in range(37)]
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join(genes) + "\n\n")
        path = f.name
    m = Metric(global_hvg_path=path)
    batches = []
    caps_all = []
    for b in range(3):
        B = 5 + b
        preds = torch.randn(B, len(genes), generator=g)
        caps = []
        for i in range(B):
            n = int(torch.randint(0, 12, (1,), generator=g))
            idx = torch.randperm(len(genes) + 6, generator=g)[:n].tolist()       # indices >= 37 are unknown genes
            caps.append(" ".join(f"GENE{j}" for j in idx))
        if b == 1:
            caps[0] = ""                      # empty caption -> zero target -> denominator 0 -> pcc 0
            preds[1] = 0.25                   # constant prediction row -> denominator 0 -> pcc 0
        targets = m._compute_rank_weighted_vector(caps, torch.device("cpu"))
        before = float(m.sum_pcc)
        m.update(preds, caps)
        batches.append({"preds": preds.numpy(), "targets": targets.numpy(), "sum_after": float(m.sum_pcc),
                        "batch_sum": float(m.sum_pcc) - before})
        caps_all.append(caps)
    os.unlink(path)
    arrs = {}
    for i, b in enumerate(batches):
        arrs[f"preds{i}"] = b["preds"]
        arrs[f"targets{i}"] = b["targets"]
    np.savez(os.path.join(OUT, "zero_shot_metric.npz"), **arrs)
    json.dump({"genes": genes, "captions": caps_all, "sum_after": [b["sum_after"] for b in batches],
               "total_count": int(m.total_count), "compute": float(m.compute())},
              open(os.path.join(OUT, "zero_shot_metric.json"), "w"), indent=1)
    print("compute() =", float(m.compute()), "count", int(m.total_count))


if __name__ == "__main__":
    main()
