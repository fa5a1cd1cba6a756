"""GPU parity of the validation-only zero-shot gene-expression path: sc_pcc_rows and the metric class
against the oracle and the reference-generated golden vectors; gene-bank construction + validation_step on a tiny CLIP
with the text tower."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import spatial_clip_oracle as O

pytestmark = pytest.mark.gpu


def _pkg():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import data, losses, metrics, model_configs, module, net, ops
    return data, losses, metrics, model_configs, module, net, ops


def test_pcc_rows_vs_oracle():
    *_, ops = _pkg()
    g = torch.Generator().manual_seed(1)
    for rows, cols in ((7, 37), (64, 2000), (3, 5001)):
        p = torch.randn(rows, cols + 3, generator=g)[:, :cols]             # strided rows
        t = (torch.rand(rows, cols, generator=g) > 0.9).float() * torch.rand(rows, cols, generator=g)
        p[0] = 0.5                                                           # constant prediction -> 0
        t[1] = 0.0                                                           # empty target -> 0
        ref = O.zero_shot_pcc_rows(p, t)
        out = torch.full((rows,), 9.0, device="cuda")
        st = torch.zeros(2, device="cuda")
        ops.pcc_rows(p.cuda(), t.cuda(), out, st)
        assert float(out[0]) == 0.0 and float(out[1]) == 0.0
        torch.testing.assert_close(out.cpu(), ref, atol=2e-6, rtol=1e-5)
        assert abs(float(st[0]) - float(ref.sum())) < 1e-4 and float(st[1]) == rows


def test_metric_matches_reference_golden(golden_dir):
    _, _, metrics, *_ = _pkg()
    j = json.load(open(os.path.join(golden_dir, "zero_shot_metric.json")))
    z = np.load(os.path.join(golden_dir, "zero_shot_metric.npz"))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join(j["genes"]) + "\n")
    m = metrics.ZeroShotGeneExpressionMetric(global_hvg_path=f.name)
    os.unlink(f.name)
    for i, caps in enumerate(j["captions"]):
        m.update(torch.from_numpy(z[f"preds{i}"]).cuda(), caps)
        assert abs(float(m.state[0]) - j["sum_after"][i]) < 2e-6
    assert float(m.state[1]) == j["total_count"]
    assert abs(m.compute() - j["compute"]) < 1e-6
    m.reset()
    assert m.compute() == 0.0


def test_validation_step_builds_gene_bank_and_scores(tmp_path):
    data, losses, metrics, mc, module, net, ops = _pkg()
    genes = [f"G{i}" for i in range(20)]
    hvg = tmp_path / "global_hvgs.txt"
    hvg.write_text("\n".join(genes) + "\n")
    ctx, vocab = 12, 64
    cfg = mc.ModelCfg(embed_dim=32, vision=mc.VisionCfg(32, 8, 64, 2, 32),
                      text=mc.TextCfg(context_length=ctx, vocab_size=vocab, width=64, heads=2, layers=2), gene=None)
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=5)

    def toy_tokenizer(names):                       # [SOT, id(name), EOT = vocab - 1 (arg-max pooled), 0 ...]
        out = torch.zeros(len(names), ctx, dtype=torch.int64)
        for i, s in enumerate(names):
            ids = [1] + [2 + (int(tok[1:]) % (vocab - 4)) for tok in s.split()][:ctx - 2] + [vocab - 1]
            out[i, :len(ids)] = torch.tensor(ids)
        return out

    n.tokenizer = toy_tokenizer
    m = module.SpatialClipLitModule(n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True), None, None,
                                    global_hvg_path=str(hvg))
    assert isinstance(m.zero_shot_metric, metrics.ZeroShotGeneExpressionMetric)
    m.on_validation_start()
    bank = m.gene_bank_embeddings
    assert bank.shape == (20, 32)
    torch.testing.assert_close(bank.norm(dim=1).cpu(), torch.ones(20), atol=1e-4, rtol=0)
    B = 6
    g = torch.Generator().manual_seed(0)
    caps = [" ".join(genes[int(k)] for k in torch.randperm(20, generator=g)[:5]) for _ in range(B)]
    batch = {"images": torch.randn(B, 3, 32, 32, generator=g).cuda(), "texts": toy_tokenizer(caps).cuda(), "raw_text": caps}
    m.validation_step(batch, 0)
    with torch.no_grad():
        f_i = n(batch["images"], batch["texts"])["image_features"].float().cpu()
    ref_rows = O.zero_shot_pcc_rows(f_i @ bank.cpu().t(), O.zero_shot_targets(caps, genes))
    assert abs(m.zero_shot_metric.compute() - float(ref_rows.mean())) < 1e-5
    assert "val/zero_shot_pcc" in m.logged
