"""GPU parity of the fused contrastive head against the golden vectors (reference outputs) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import spatial_clip_oracle as O

pytestmark = pytest.mark.gpu


def _head():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import contrastive
    return contrastive


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", ["w1_default", "w1_edges", "w1_capped", "w1_noreg_nocap", "w1_bias"])
def test_losses_w1_vs_reference_golden(golden_dir, tag):
    C = _head()
    z = load(golden_dir, f"loss_{tag}.npz")
    cap = float(z["cap"]); cap = None if cap < 0 else cap
    bias = torch.tensor(float(z["bias"])).cuda() if int(z["has_bias"]) else None
    img, txt = z["img"].cuda(), z["txt"].cuda()
    s = z["scale"].float().cuda()
    for which in ("spatial", "clip"):
        if which == "spatial":
            res = C.contrastive_forward_backward(
                img, txt, s, mode="spatial", image_tile_ids=z["ids"].cuda(), text_tile_ids=z["ids"].cuda(),
                neighbor_tile_ids=z["nb"].cuda(), neighbor_alphas=z["alpha"].cuda(), cap_logit_scale=cap,
                temp_reg_weight=float(z["w"]), neighbor_alpha_scale=0.5, logit_bias=bias)
            pre = "sp"
        else:
            res = C.contrastive_forward_backward(img, txt, s, mode="clip", logit_bias=bias)
            pre = "cl"
        assert abs(float(res["loss"]) - float(z[f"{which}_loss"])) < 5e-6
        # world_size 1: the gathered features ARE the local ones, so both terms land on the same rows
        torch.testing.assert_close((res["d_image"] + res["d_all_image"]).cpu(), z[f"{pre}_gimg"], atol=2e-6, rtol=1e-4)
        torch.testing.assert_close((res["d_text"] + res["d_all_text"]).cpu(), z[f"{pre}_gtxt"], atol=2e-6, rtol=1e-4)
        assert abs(float(res["d_scale"]) - float(z[f"{pre}_gscale"])) < 2e-6


def test_single_process_head_and_loss_node_vs_reference_golden(golden_dir):
    """Round 4: in a single process the head's GEMMs accumulate the gathered-feature terms onto the direct ones
    (``join_local``: no ATen adds), and the loss node applies the upstream gradient with one ``sc_scale_by_scalar`` launch
    over d_image | d_text | d_scale.  Both checked against the reference's own gradients (golden), the second through the
    product's loss classes with an upstream gradient of 1 and of -2.5."""
    C = _head()
    from spatial_clip_amd import losses
    z = load(golden_dir, "loss_w1_default.npz")
    img, txt, s = z["img"].cuda(), z["txt"].cuda(), z["scale"].float().cuda()
    res = C.contrastive_forward_backward(img, txt, s, mode="clip", join_local=True, want_recall=False)
    assert "d_all" not in res and res["recall_hits"] is None
    torch.testing.assert_close(res["d_image"].cpu(), z["cl_gimg"], atol=2e-6, rtol=1e-4)
    torch.testing.assert_close(res["d_text"].cpu(), z["cl_gtxt"], atol=2e-6, rtol=1e-4)
    for up in (1.0, -2.5):
        a, b, sc = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), s.clone().requires_grad_(True)
        fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=float(z["cap"]) if float(z["cap"]) >= 0 else None,
                                temp_reg_weight=float(z["w"]), neighbor_alpha_scale=0.5, float32_logits=True)
        out = fn(a, b, sc, z["ids"].cuda(), z["ids"].cuda(), z["nb"].cuda(), z["alpha"].cuda())["contrastive_loss"]
        assert abs(float(out) - float(z["spatial_loss"])) < 5e-6
        (out * up).backward()
        torch.testing.assert_close(a.grad.cpu(), up * z["sp_gimg"], atol=5e-6, rtol=1e-4)
        torch.testing.assert_close(b.grad.cpu(), up * z["sp_gtxt"], atol=5e-6, rtol=1e-4)
        assert abs(float(sc.grad) - up * float(z["sp_gscale"])) < 5e-6


def test_losses_w2_emulated_ranks(golden_dir):
    """Each rank's loss / local grads for a 2-rank global batch, emulated on one GPU: the cross-rank
    d(all_features) terms are summed by hand exactly as the reduce-scatter would."""
    C = _head()
    z = load(golden_dir, "loss_w2.npz")
    W, G = 2, z["img"].shape[0]
    B = G // W
    img, txt, ids = z["img"].cuda(), z["txt"].cuda(), z["ids"].cuda()
    s = torch.tensor(float(z["scale"])).cuda()
    for which in ("spatial", "clip"):
        outs = []
        for r in range(W):
            sl = slice(r * B, (r + 1) * B)
            kw = dict(all_image=img, all_text=txt, rank=r)
            if which == "spatial":
                kw.update(mode="spatial", image_tile_ids=ids[sl], text_tile_ids=ids[sl], all_image_tile_ids=ids,
                          all_text_tile_ids=ids, neighbor_tile_ids=z["nb"][sl].cuda(),
                          neighbor_alphas=z["alpha"][sl].cuda(), cap_logit_scale=40.0, temp_reg_weight=0.05,
                          neighbor_alpha_scale=0.5)
            else:
                kw.update(mode="clip")
            res = C.contrastive_forward_backward(img[sl].contiguous(), txt[sl].contiguous(), s, **kw)
            assert abs(float(res["loss"]) - float(z[f"r{r}_{which}_loss"])) < 5e-6
            outs.append(res)
        for r in range(W):
            sl = slice(r * B, (r + 1) * B)
            gi = outs[r]["d_image"] + sum(o["d_all_image"][sl] for o in outs)
            gt = outs[r]["d_text"] + sum(o["d_all_text"][sl] for o in outs)
            torch.testing.assert_close(gi.cpu(), z[f"r{r}_{which}_gimg"], atol=2e-6, rtol=1e-4)
            torch.testing.assert_close(gt.cpu(), z[f"r{r}_{which}_gtxt"], atol=2e-6, rtol=1e-4)
            assert abs(float(outs[r]["d_scale"]) - float(z[f"r{r}_{which}_gscale"])) < 2e-6


def test_spatial_loss_large_vs_oracle():
    C = _head()
    g = torch.Generator().manual_seed(0)
    B, G, D, K = 64, 256, 128, 8
    img = torch.nn.functional.normalize(torch.randn(G, D, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img + 0.5 * torch.randn(G, D, generator=g), dim=-1)
    ids = 1000 + torch.randperm(G, generator=g)
    nb = ids[torch.randint(0, G, (B, K), generator=g)]
    nb[:, -1] = -1
    al = torch.rand(B, K, generator=g); al[:, -1] = 0
    r = 2
    sl = slice(r * B, (r + 1) * B)
    il = img[sl].clone().requires_grad_(True); tl = txt[sl].clone().requires_grad_(True)
    s = torch.tensor(30.0, requires_grad=True)
    ai = img.clone().requires_grad_(True); at = txt.clone().requires_grad_(True)
    loss = O.spatial_loss(il, tl, s, ids[sl], ids[sl], nb, al, ai, at, ids, ids, rank=r)
    loss.backward()
    res = C.contrastive_forward_backward(img[sl].contiguous().cuda(), txt[sl].contiguous().cuda(), s.detach().cuda(),
                                         mode="spatial", all_image=img.cuda(), all_text=txt.cuda(), rank=r,
                                         image_tile_ids=ids[sl].cuda(), text_tile_ids=ids[sl].cuda(),
                                         all_image_tile_ids=ids.cuda(), all_text_tile_ids=ids.cuda(),
                                         neighbor_tile_ids=nb.cuda(), neighbor_alphas=al.cuda(), cap_logit_scale=40.0,
                                         temp_reg_weight=0.05, neighbor_alpha_scale=0.5)
    assert abs(float(res["loss"]) - float(loss)) < 1e-5
    torch.testing.assert_close(res["d_image"].cpu(), il.grad, atol=1e-6, rtol=1e-3)
    torch.testing.assert_close(res["d_all_text"].cpu(), at.grad, atol=1e-6, rtol=1e-3)
    torch.testing.assert_close(res["d_all_image"].cpu(), ai.grad, atol=1e-6, rtol=1e-3)
    # recall hits vs metrics.py semantics
    logits = (il.detach() @ tl.detach().t()) * 30.0
    for k, h in zip((1, 5, 10), res["recall_hits"].cpu().tolist()):
        assert h == O.recall_at_k(logits, torch.arange(B), k)[0]


@pytest.mark.parametrize("which", ["clip", "spatial"])
@pytest.mark.parametrize("B,D,ms_bound", [(256, 512, 0.6), (1024, 768, 2.5)])
def test_head_at_8gpu_per_rank_size_vs_oracle(which, B, D, ms_bound):
    """Per-rank head workload of an 8-GPU run emulated on one GPU, rank 3 of 8, features read in place from a padded gather
    buffer -- loss, every gradient and the time.  (256, 512): BASELINE configs[2]/[3], local batch 256 of a global batch
    2048.  (1024, 768): configs[4], local batch 1024 of a global batch 8192, ViT-L/14's embed_dim (round 4)."""
    C = _head()
    g = torch.Generator().manual_seed(3)
    W, K, r = 8, 8, 3
    G = B * W
    img = torch.nn.functional.normalize(torch.randn(G, D, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(img + 0.7 * torch.randn(G, D, generator=g), dim=-1)
    ids = 10_000 + torch.randperm(G, generator=g)
    sl = slice(r * B, (r + 1) * B)
    nb = ids[torch.randint(0, G, (B, K), generator=g)]
    nb[:, -1] = -1
    al = torch.rand(B, K, generator=g); al[:, -1] = 0
    al = al / al.sum(1, keepdim=True)
    il = img[sl].clone().requires_grad_(True); tl = txt[sl].clone().requires_grad_(True)
    s = torch.tensor(14.2857, requires_grad=True)
    ai = img.clone().requires_grad_(True); at = txt.clone().requires_grad_(True)
    if which == "clip":
        loss = O.clip_loss(il, tl, s, ai, at, rank=r)
        kw = dict(mode="clip")
    else:
        loss = O.spatial_loss(il, tl, s, ids[sl], ids[sl], nb, al, ai, at, ids, ids, rank=r)
        kw = dict(mode="spatial", image_tile_ids=ids[sl].cuda(), text_tile_ids=ids[sl].cuda(),
                  all_image_tile_ids=ids.cuda(), all_text_tile_ids=ids.cuda(), neighbor_tile_ids=nb.cuda(),
                  neighbor_alphas=al.cuda(), cap_logit_scale=40.0, temp_reg_weight=0.05, neighbor_alpha_scale=0.5)
    loss.backward()
    buf_i = torch.zeros(G, D + 4, device="cuda"); buf_i[:, :D] = img.cuda()       # as comm.FeatureGather hands them over
    buf_t = torch.zeros(G, D + 4, device="cuda"); buf_t[:, :D] = txt.cuda()
    args = (img[sl].contiguous().cuda(), txt[sl].contiguous().cuda(), s.detach().cuda())
    res = C.contrastive_forward_backward(*args, all_text=buf_t[:, :D], rank=r, late_all_image=lambda: buf_i[:, :D], **kw)
    assert abs(float(res["loss"]) - float(loss)) < 1e-5, (float(res["loss"]), float(loss))
    torch.testing.assert_close(res["d_image"].cpu(), il.grad, atol=1e-6, rtol=1e-3)
    torch.testing.assert_close(res["d_text"].cpu(), tl.grad, atol=1e-6, rtol=1e-3)
    torch.testing.assert_close(res["d_all_text"].cpu(), at.grad, atol=1e-6, rtol=1e-3)
    torch.testing.assert_close(res["d_all_image"].cpu(), ai.grad, atol=1e-6, rtol=1e-3)
    assert abs(float(res["d_scale"]) - float(s.grad)) < 1e-5
    # the whole head (2 + 4 GEMMs, label join, row passes) < 0.3 ms at B=256, G=2048
    for _ in range(3):
        C.contrastive_forward_backward(*args, all_text=buf_t[:, :D], rank=r, all_image=buf_i[:, :D], **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        C.contrastive_forward_backward(*args, all_text=buf_t[:, :D], rank=r, all_image=buf_i[:, :D], **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"[head {which}] B={B} G={G} D={D}: {ms:.3f} ms per forward+backward")
    assert ms < ms_bound, ms     # wall incl. Python launch overhead of ~14 launches; kernel time is reported by tools/bench_head.py
