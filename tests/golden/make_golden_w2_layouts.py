"""Golden vectors for the remaining (local_loss, gather_with_grad) layouts of the reference's ClipLoss at world_size 2.

Run in the BUILD container only (imports the reference from /root/reference, never copies it):
    python tests/golden/make_golden_w2_layouts.py
Re-uses the inputs of loss_w2.npz (make_golden.py) and records, per rank, loss and the gradients w.r.t. the local
features and logit_scale for
    local_loss=False, gather_with_grad=True    full [G,G] logits, differentiable gather   (loss.py:50-52,119-121)
    local_loss=False, gather_with_grad=False   full [G,G] logits, local shard spliced in   (loss.py:54-63)
    local_loss=True,  gather_with_grad=False   [B,G] logits, remote shards detached
through the reference's real gather_features over gloo.  Output: loss_w2_layouts.npz."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

LAYOUTS = {"gg_grad": (False, True), "gg_nograd": (False, False), "bg_nograd": (True, False)}


def _worker(rank, world, img, txt, scale, ret):
    sys.dont_write_bytecode = True
    import torch.distributed as dist
    from make_golden import import_reference
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29541"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, _, ref_losses = import_reference()
    B = img.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    out = {}
    for name, (local_loss, gwg) in LAYOUTS.items():
        i = img[sl].clone().requires_grad_(True)
        t = txt[sl].clone().requires_grad_(True)
        s = torch.tensor(scale, requires_grad=True)
        crit = ref_losses.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, cache_labels=True, rank=rank,
                                   world_size=world)
        l = crit(i, t, s)["contrastive_loss"]
        l.backward()
        out[f"{name}_loss"] = l.detach().numpy()
        out[f"{name}_gimg"] = i.grad.numpy()
        out[f"{name}_gtxt"] = t.grad.numpy()
        out[f"{name}_gscale"] = s.grad.numpy()
    ret[rank] = out
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    z = np.load(os.path.join(HERE, "loss_w2.npz"))
    img, txt, scale = torch.from_numpy(z["img"]), torch.from_numpy(z["txt"]), float(z["scale"])
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, img, txt, scale, ret), nprocs=2, join=True)
        res = dict(ret)
    arrs = {}
    for r in (0, 1):
        for k, v in res[r].items():
            arrs[f"r{r}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "loss_w2_layouts.npz"), **arrs)
    print("wrote loss_w2_layouts.npz", sorted(arrs)[:6], len(arrs))


if __name__ == "__main__":
    main()
