#!/usr/bin/env python3
"""Generate golden vectors for the hot path by importing the REFERENCE's own modules.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import recipe: package shell for ``open_clip`` that skips its ``__init__`` (which
needs torchvision/ftfy), a one-symbol torchvision stub, ``transformers`` masked so hf_model takes its
ImportError branch; then ``open_clip.model`` / ``.loss`` / ``.transformer`` import cleanly and
``src/models/components/losses.py`` is loaded by file path.  Nothing is written into /root/reference
and no reference source is copied: the outputs are inputs + expected outputs (.npz) only.
"""
import sys

sys.dont_write_bytecode = True
import importlib
import importlib.util
import json
import os
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    shell = types.ModuleType("open_clip")
    shell.__path__ = [f"{REF}/src/open_clip"]
    sys.modules["open_clip"] = shell
    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")
    tvm = types.ModuleType("torchvision.ops.misc")

    class FrozenBatchNorm2d(nn.Module):
        pass

    tvm.FrozenBatchNorm2d = FrozenBatchNorm2d
    tv.ops = tvo
    tvo.misc = tvm
    sys.modules.update({"torchvision": tv, "torchvision.ops": tvo, "torchvision.ops.misc": tvm})
    sys.modules["transformers"] = None
    model = importlib.import_module("open_clip.model")
    loss = importlib.import_module("open_clip.loss")
    importlib.import_module("open_clip.transformer")
    shell.ClipLoss = loss.ClipLoss
    spec = importlib.util.spec_from_file_location("ref_losses", f"{REF}/src/models/components/losses.py")
    ref_losses = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_losses)
    return model, loss, ref_losses


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if v.ndim} if len(out) < 12 else len(out))


def sd_np(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def unit(x):
    return torch.nn.functional.normalize(x, dim=-1)


def make_batch_ids(G, K, g, dup=False, miss=False):
    """tile ids on a grid + Moore neighbours; optional duplicates / absent neighbours / pads."""
    ids = 10_000 + torch.arange(G)
    cols = 8
    nb = torch.full((G, K), -1, dtype=torch.long)
    al = torch.zeros(G, K)
    for i in range(G):
        r, c = divmod(i, cols)
        cand = []
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                if dr == 0 and dc == 0:
                    continue
                rr, cc = r + dr, c + dc
                if 0 <= cc < cols and rr >= 0:
                    cand.append(10_000 + rr * cols + cc)   # may exceed G -> absent from batch
        cand = cand[:K]
        w = torch.rand(len(cand), generator=g) + 0.1
        w = w / w.sum()
        nb[i, :len(cand)] = torch.tensor(cand)
        al[i, :len(cand)] = w
    if dup:
        ids[3] = ids[1]            # duplicate id: LAST index must win in the lookup
        nb[0, 0] = ids[0]          # self-neighbour adds onto the diagonal
        nb[2, 1] = nb[2, 0]        # repeated neighbour accumulates
        al[4, 0] = -0.3            # alpha <= 0 is skipped
    if miss:
        nb[5, :2] = 99_999_999     # neighbour absent from batch
    return ids, nb, al


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, loss_mod, ref_losses = import_reference()
    tf = sys.modules["open_clip.transformer"]

    # ---------------- per-op: residual attention block (non-causal and causal) ----------------
    for tag, d, h, L, causal in (("blk_d64", 64, 2, 17, False), ("blk_d64_causal", 64, 2, 13, True),
                                 ("blk_d128", 128, 2, 197, False)):
        torch.manual_seed(1)
        blk = tf.ResidualAttentionBlock(d, h)
        for n, p_ in blk.named_parameters():   # make biases / LN affine non-trivial
            if p_.ndim == 1:
                p_.data.add_(0.1 * torch.randn_like(p_))
        x = torch.randn(3, L, d, requires_grad=True)
        mask = torch.full((L, L), float("-inf")).triu_(1) if causal else None
        y = blk(x, attn_mask=mask)
        gy = torch.randn_like(y)
        y.backward(gy)
        arrs = {"x": x, "y": y, "gy": gy, "gx": x.grad, "heads": h, "causal": int(causal)}
        for n, p_ in blk.named_parameters():
            arrs["p." + n] = p_
            arrs["g." + n] = p_.grad
        npz(f"{tag}.npz", **arrs)

    # ---------------- tiny CLIP (vision + reference text tower): fwd + grads ----------------
    tiny = {"embed_dim": 32,
            "vision_cfg": {"image_size": 32, "layers": 2, "width": 64, "patch_size": 8, "head_width": 32},
            "text_cfg": {"context_length": 16, "vocab_size": 97, "width": 64, "heads": 2, "layers": 2}}
    torch.manual_seed(2)
    clip = model.CLIP(**tiny)
    for n, p_ in clip.named_parameters():
        if p_.ndim == 1 and "logit" not in n:
            p_.data.add_(0.05 * torch.randn_like(p_))
    B = 6
    images = torch.randn(B, 3, 32, 32)
    texts = torch.zeros(B, 16, dtype=torch.long)
    g = torch.Generator().manual_seed(3)
    for b in range(B):
        n = int(torch.randint(3, 14, (1,), generator=g))
        texts[b, 0] = 95
        texts[b, 1:1 + n] = torch.randint(1, 95, (n,), generator=g)
        texts[b, 1 + n] = 96      # EOT = max id
    img_f = clip.encode_image(images, normalize=True)
    txt_f = clip.encode_text(texts, normalize=True)
    crit = ref_losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
    loss = crit(img_f, txt_f, clip.logit_scale.exp())["contrastive_loss"]
    loss.backward()
    arrs = {"images": images, "texts": texts, "image_features": img_f, "text_features": txt_f,
            "loss": loss, "cfg": json.dumps(tiny)}
    for n, p_ in clip.named_parameters():
        arrs["p." + n] = p_
        arrs["g." + n] = p_.grad if p_.grad is not None else torch.zeros_like(p_)
    npz("clip_tiny_fwd_bwd.npz", **arrs)

    # ---------------- losses, W=1 -----------------------------------------------------------
    g = torch.Generator().manual_seed(4)
    for tag, G, D, K, s, dup, miss, cap, w, bias in (
            ("w1_default", 16, 32, 4, 14.2857, False, False, 40.0, 0.05, None),
            ("w1_edges", 16, 32, 4, 14.2857, True, True, 40.0, 0.05, None),
            ("w1_capped", 12, 16, 4, 50.0, True, False, 40.0, 0.05, None),
            ("w1_noreg_nocap", 12, 16, 4, 20.0, False, True, None, 0.0, None),
            ("w1_bias", 12, 16, 4, 10.0, False, False, 40.0, 0.05, -1.5)):
        img = unit(torch.randn(G, D, generator=g)).requires_grad_(True)
        txt = unit(torch.randn(G, D, generator=g)).requires_grad_(True)
        sc = torch.tensor(s, requires_grad=True)
        ids, nb, al = make_batch_ids(G, K, g, dup, miss)
        lb = torch.tensor(bias) if bias is not None else None
        sp = ref_losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=cap,
                                    temp_reg_weight=w, neighbor_alpha_scale=0.5, float32_logits=True)
        l_sp = sp(img, txt, sc, ids, ids.clone(), nb, al, logit_bias=lb)["contrastive_loss"]
        l_sp.backward()
        gs = (img.grad.clone(), txt.grad.clone(), sc.grad.clone())
        img.grad = txt.grad = sc.grad = None
        cl = ref_losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
        l_cl = cl(img, txt, sc, logit_bias=lb)["contrastive_loss"]
        l_cl.backward()
        npz(f"loss_{tag}.npz", img=img, txt=txt, scale=sc, ids=ids, nb=nb, alpha=al,
            cap=(cap if cap is not None else -1.0), w=w, bias=(bias if bias is not None else 0.0),
            has_bias=int(bias is not None),
            spatial_loss=l_sp, sp_gimg=gs[0], sp_gtxt=gs[1], sp_gscale=gs[2],
            clip_loss=l_cl, cl_gimg=img.grad, cl_gtxt=txt.grad, cl_gscale=sc.grad)

    # ---------------- losses, W=2 over gloo (real gather_features) ---------------------------
    import torch.multiprocessing as mp
    G, D, K = 16, 32, 4
    g = torch.Generator().manual_seed(5)
    img = unit(torch.randn(G, D, generator=g))
    txt = unit(torch.randn(G, D, generator=g))
    ids, nb, al = make_batch_ids(G, K, g, True, True)
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_w2_worker, args=(2, img, txt, ids, nb, al, ret), nprocs=2, join=True)
        res = dict(ret)
    arrs = {"img": img, "txt": txt, "ids": ids, "nb": nb, "alpha": al, "scale": 14.2857}
    for r in (0, 1):
        for k, v in res[r].items():
            arrs[f"r{r}_{k}"] = v
    npz("loss_w2.npz", **arrs)

    # ---------------- full tiny training_step x3 (vision tower + reference text tower) -------
    torch.manual_seed(6)
    clip = model.CLIP(**tiny)
    p0 = sd_np(clip)
    opt = torch.optim.AdamW(clip.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    warm, total = 2, 10

    def lam(step):
        import math
        if step < warm:
            return step / max(1, warm)
        pr = (step - warm) / max(1, total - warm)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * pr)))

    sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    crit = ref_losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0,
                                  temp_reg_weight=0.05, neighbor_alpha_scale=0.5, float32_logits=True)
    ids, nb, al = make_batch_ids(B, 4, torch.Generator().manual_seed(7))
    losses, norms = [], []
    for step in range(3):
        opt.zero_grad()
        f_i = clip.encode_image(images, normalize=True)
        f_t = clip.encode_text(texts, normalize=True)
        l = crit(f_i, f_t, clip.logit_scale.exp(), ids, ids.clone(), nb, al)["contrastive_loss"]
        l.backward()
        n = torch.nn.utils.clip_grad_norm_(clip.parameters(), 1.0)
        opt.step()
        sched.step()
        losses.append(float(l))
        norms.append(float(n))
    arrs = {"images": images, "texts": texts, "ids": ids, "nb": nb, "alpha": al,
            "losses": np.array(losses), "grad_norms": np.array(norms), "cfg": json.dumps(tiny),
            "warmup": warm, "total": total}
    for k, v in p0.items():
        arrs["p0." + k] = v
    for k, v in sd_np(clip).items():
        arrs["p3." + k] = v
    npz("train3_tiny_text.npz", **arrs)

    # ---------------- state_dict manifests -------------------------------------------------
    manifest = {}
    for name in ("ViT-B-16", "ViT-L-14", "ViT-B-32"):
        cfg = json.load(open(f"{REF}/src/open_clip/model_configs/{name}.json"))
        with torch.device("meta"):
            m = model.CLIP(**cfg)
        manifest[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(manifest, open(os.path.join(OUT, "state_dict_manifest.json"), "w"), indent=0)
    print("wrote state_dict_manifest.json")


def _w2_worker(rank, world, img, txt, ids, nb, al, ret):
    sys.dont_write_bytecode = True
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29533"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, ref_losses = import_reference()
    B = img.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    out = {}
    for name in ("spatial", "clip"):
        i = img[sl].clone().requires_grad_(True)
        t = txt[sl].clone().requires_grad_(True)
        s = torch.tensor(14.2857, requires_grad=True)
        if name == "spatial":
            crit = ref_losses.SpatialLoss(local_loss=True, gather_with_grad=True, rank=rank, world_size=world,
                                          cap_logit_scale=40.0, temp_reg_weight=0.05,
                                          neighbor_alpha_scale=0.5, float32_logits=True)
            l = crit(i, t, s, ids[sl].clone(), ids[sl].clone(), nb[sl], al[sl])["contrastive_loss"]
        else:
            crit = ref_losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True,
                                       rank=rank, world_size=world)
            l = crit(i, t, s)["contrastive_loss"]
        l.backward()
        out[f"{name}_loss"] = l.detach().numpy()
        out[f"{name}_gimg"] = i.grad.numpy()
        out[f"{name}_gtxt"] = t.grad.numpy()
        out[f"{name}_gscale"] = s.grad.numpy()
    ret[rank] = out
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
