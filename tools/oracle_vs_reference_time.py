#!/usr/bin/env python3
"""SURVEY.md 8(d) cross-check, BUILD CONTAINER ONLY (imports the reference from /root/reference, never copies it):
the CPU step time of the oracle restatement against the imported reference's own modules on the same weights, batch
and optimiser -- the oracle is what bench.py times as `cpu_baseline` on the GPU box (where the reference cannot
travel), so its step time has to be representative of the reference's (target: within +-10 %).

    python tools/oracle_vs_reference_time.py            -> profiles/r02_oracle_vs_reference_cpu.json

Workload: ViT-B/16 + the reference's 12-layer text tower (the configuration the reference can run: it has no gene
tower), fp32, B = 8, ClipLoss, AdamW(lr 1e-3, betas (0.9, 0.98), eps 1e-6, wd 0.1), clip_grad_norm_ 1.0, 1 warm-up + 3
timed steps each, all host cores of this container."""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

from make_golden import import_reference          # the reference-import recipe of SURVEY.md 8(c)
from oracle import spatial_clip_oracle as O


def main():
    torch.manual_seed(0)
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    model, loss, ref_losses = import_reference()
    cfg = json.load(open("/root/reference/src/open_clip/model_configs/ViT-B-16.json"))
    clip = model.CLIP(**cfg)
    B = 8
    g = torch.Generator().manual_seed(1)
    images = torch.randn(B, 3, 224, 224, generator=g)
    texts = torch.zeros(B, 77, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(20, 76, (1,), generator=g))
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49406, (n,), generator=g)
        texts[b, 1 + n] = 49407
    params = {k: v.detach().clone() for k, v in clip.state_dict().items() if k != "attn_mask"}
    # ---- reference
    opt = torch.optim.AdamW(clip.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1)
    crit = ref_losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
    ref_t, ref_l = [], []
    for step in range(4):
        t0 = time.time()
        opt.zero_grad()
        f_i = clip.encode_image(images, normalize=True)
        f_t = clip.encode_text(texts, normalize=True)
        l = crit(f_i, f_t, clip.logit_scale.exp())["contrastive_loss"]
        l.backward()
        torch.nn.utils.clip_grad_norm_(clip.parameters(), 1.0)
        opt.step()
        ref_t.append(time.time() - t0)
        ref_l.append(float(l))
    # ---- oracle restatement on the same initial weights
    v = cfg["vision_cfg"]
    t = cfg["text_cfg"]
    ocfg = O.ModelCfg(cfg["embed_dim"], O.VisionCfg(v["image_size"], v["patch_size"], v["width"], v["layers"], 64),
                      O.TextCfg(t["context_length"], t["vocab_size"], t["width"], t["heads"], t["layers"]), None)
    O.USE_ATEN_KERNELS = True       # the timed form of the oracle: same maths through the stock ATen kernels the reference uses
    tr = O.OracleTrainer(ocfg, params, loss="clip", lr=1e-3, warmup=0, total_steps=10 ** 9)
    batch = {"images": images, "texts": texts}
    or_t, or_l = [], []
    for step in range(4):
        t0 = time.time()
        out = tr.training_step(batch)
        or_t.append(time.time() - t0)
        or_l.append(float(out["loss"]))
    best_ref, best_or = min(ref_t[1:]), min(or_t[1:])
    rec = {"workload": "ViT-B/16 + reference 12-layer text tower, fp32, B=8, ClipLoss, AdamW + clip 1.0",
           "host": {"cores": threads, "torch": torch.__version__},
           "reference": {"step_s": [round(x, 3) for x in ref_t], "best_pairs_per_s": round(B / best_ref, 3), "loss": ref_l},
           "oracle": {"step_s": [round(x, 3) for x in or_t], "best_pairs_per_s": round(B / best_or, 3), "loss": or_l},
           "oracle_over_reference_step_time": round(best_or / best_ref, 3),
           "first_step_loss_abs_diff": abs(ref_l[0] - or_l[0]),
           "oracle_mode": "USE_ATEN_KERNELS=True (F.layer_norm / F.gelu / SDPA / foreach AdamW), as timed by bench.py's cpu_baseline",
           "note": "oracle lr schedule set to constant (warmup 0) to mirror the plain AdamW loop used for the reference here"}
    out_path = os.path.join(ROOT, "profiles", "r02_oracle_vs_reference_cpu.json")
    json.dump(rec, open(out_path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
