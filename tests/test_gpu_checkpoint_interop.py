"""SURVEY 8(f) rank 2, finished in round 5: what a user of the reference needs to fine-tune OpenAI / LAION weights --
QuickGELU towers (`quick_gelu: true`: src/open_clip/model.py:142-145,228, transformer.py:32-35,
model_configs/ViT-B-16-quickgelu.json), the checkpoint file formats of open_clip.factory.load_state_dict
(factory.py:153-178: .safetensors, pickles, TorchScript archives) behind ``SpatialClipNet(pretrained=<local file>)``, and
``resize_text_pos_embed`` (model.py:826-860) next to ``resize_pos_embed`` at load time."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pkg():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import losses, model_configs, module, net, ops
    return losses, model_configs, module, net, ops


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


@pytest.mark.parametrize("M,N,K", [(333, 192, 256), (197 * 4, 768, 192), (256, 3072, 768), (256 * 86 + 24, 3072, 192),
                                   (64, 256, 64)])
def test_quickgelu_epilogues(M, N, K):
    """SC_EPI_QGELU_PAIR / SC_EPI_QGELU_GRAD_PAIR / SC_EPI_BF16_DQGELU over the 128-tile kernel, the 256-tile kernel (table
    epilogue) and the persistent walk: h and the stored factor against torch on the kernel's own bf16 u; the stored-factor and
    the recomputation path agree bit for bit; sc_quick_gelu_bf16 rebuilds h bit for bit; none of it equals the erf GELU."""
    losses, mc, module, net, ops = _pkg()
    g = torch.Generator().manual_seed(3 + M)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.15)
    bias = torch.randn(N, generator=g)
    ad, bd, biasd = a.cuda(), b.cuda(), bias.cuda()
    u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_QGELU_PAIR, ad, bd, u, M=M, N=N, K=K, bias=biasd, out2=h)
    gd = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    h2 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_QGELU_GRAD_PAIR, ad, bd, gd, M=M, N=N, K=K, bias=biasd, out2=h2)
    assert torch.equal(h, h2)
    u_erf = torch.empty_like(u)
    h_erf = torch.empty_like(h)
    ops.gemm(ops.NT, ops.EPI_GELU_PAIR, ad, bd, u_erf, M=M, N=N, K=K, bias=biasd, out2=h_erf)
    assert torch.equal(u, u_erf) and not torch.equal(h, h_erf)             # same pre-activation, another activation
    x = u.float().cpu().requires_grad_(True)
    y = quick_gelu(x)
    y.sum().backward()
    torch.testing.assert_close(h.float().cpu(), y.detach().to(torch.bfloat16).float(), atol=8e-3, rtol=8e-3)
    torch.testing.assert_close(gd.float().cpu(), x.grad.to(torch.bfloat16).float(), atol=4e-3, rtol=8e-3)
    h3 = torch.empty_like(h)
    ops.gelu_bf16(u, h3, quick=True)                                          # activation recomputation's rebuild
    assert torch.equal(h3, h)
    K2 = 128
    dy, w = _rand((M, K2), g), _rand((N, K2), g, 0.1)
    d_mul = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    d_rec = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_BF16_MUL_AUX, dy.cuda(), w.cuda(), d_mul, M=M, N=N, K=K2, aux=gd)
    ops.gemm(ops.NT, ops.EPI_BF16_DQGELU, dy.cuda(), w.cuda(), d_rec, M=M, N=N, K=K2, aux=u)
    assert torch.equal(d_mul, d_rec)
    ref = (dy.float() @ w.float().t()) * x.grad
    torch.testing.assert_close(d_mul.float().cpu(), ref.to(torch.bfloat16).float(), atol=3e-2, rtol=3e-2)


def test_quickgelu_table_equals_formula_for_every_bf16_input(monkeypatch):
    """The 256-tile kernel's table epilogue (one table per activation, filled by the QuickGELU formula) against the formula
    path of the same kernel (SC_GELU_LUT=0) for EVERY finite bf16 pre-activation (u = a . 1 exactly), and against torch."""
    losses, mc, module, net, ops = _pkg()
    bits = torch.arange(65536, dtype=torch.int32)
    vals = bits.to(torch.int16).view(torch.bfloat16)
    vals = torch.where(torch.isfinite(vals.float()), vals, torch.zeros_like(vals))
    M, N, K = 65536, 256, 64
    a = torch.zeros((M, K), dtype=torch.bfloat16)
    a[:, 0] = vals
    b = torch.zeros((N, K), dtype=torch.bfloat16)
    b[:, 0] = 1.0
    bias = torch.zeros(N)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SC_GELU_LUT", sw)
        gd = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
        h = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(ops.NT, ops.EPI_QGELU_GRAD_PAIR, a.cuda(), b.cuda(), gd, M=M, N=N, K=K, bias=bias.cuda(), out2=h)
        outs[sw] = (gd.view(torch.int16).cpu(), h.view(torch.int16).cpu())
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    u = vals.float().requires_grad_(True)
    y = quick_gelu(u)
    y.sum().backward()
    h = outs["1"][1].view(torch.bfloat16).float()[:, 7]
    gq = outs["1"][0].view(torch.bfloat16).float()[:, 7]
    assert torch.isfinite(h).all() and torch.isfinite(gq).all()          # no overflow at the ends of the bf16 range
    sane = u.detach().abs() < 1e30
    torch.testing.assert_close(h[sane], y.detach()[sane].to(torch.bfloat16).float(), atol=4e-3, rtol=8e-3)
    torch.testing.assert_close(gq[sane], u.grad[sane].to(torch.bfloat16).float(), atol=4e-3, rtol=8e-3)


def _load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    out = {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}
    c = json.loads(str(out["cfg"]))
    losses, mc, module, net, ops = _pkg()
    v, t = c["vision_cfg"], c["text_cfg"]
    cfg = mc.ModelCfg(embed_dim=c["embed_dim"],
                      vision=mc.VisionCfg(v["image_size"], v["patch_size"], v["width"], v["layers"], v.get("head_width", 64)),
                      text=mc.TextCfg(t["context_length"], t["vocab_size"], t["width"], t["heads"], t["layers"]), gene=None,
                      quick_gelu=bool(c.get("quick_gelu", False)))
    return out, cfg


@pytest.mark.parametrize("recompute", [False, True])
def test_reference_clip_tiny_quickgelu_forward_backward(golden_dir, recompute):
    """The reference's own CLIP(quick_gelu=True) outputs (tests/golden/clip_tiny_quickgelu_fwd_bwd.npz) on the HIP path:
    features, ClipLoss, every parameter gradient -- default mode (stored factor) and activation recomputation."""
    losses, mc, module, net, ops = _pkg()
    z, cfg = _load_golden(golden_dir, "clip_tiny_quickgelu_fwd_bwd.npz")
    assert cfg.quick_gelu
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, grad_checkpointing=recompute)
    assert n.vision.stack.quick_gelu and n.second.stack.quick_gelu
    n.load_state_dict({k[2:]: v for k, v in z.items() if k.startswith("p.")})
    m = module.SpatialClipLitModule(n, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True), None, None)
    out = m.model_step({"images": z["images"].cuda(), "texts": z["texts"].cuda()})
    assert (out["image_features"].cpu() - z["image_features"]).abs().max() < 5e-3
    assert (out["text_features"].cpu() - z["text_features"]).abs().max() < 5e-3
    assert abs(float(out["loss"].detach()) - float(z["loss"])) < 4e-3
    out["loss"].backward()
    torch.cuda.synchronize()

    def grad_errors(netobj):
        e = {}
        for k in netobj.store.by_name:
            g_ref = z["g." + k].double()
            if float(g_ref.norm()) > 1e-5:
                e[k] = float((netobj.store.g(k).cpu().double() - g_ref).norm() / g_ref.norm())
        return e
    e = grad_errors(n)
    worst, med = max(e.values()), float(np.median(list(e.values())))
    print(f"[quick_gelu, recompute={recompute}] relative L2 gradient error vs the reference: median {med:.4f}, worst {worst:.4f}")
    # measured 0.7 % / 1.7 %.  The same weights through an erf-GELU net sit at 2.4 % / 4.3 % from this fixture even in fp32
    # (oracle with quick=False), so these bounds separate the two activations
    assert med <= 0.015 and worst <= 0.03, (med, worst, max(e, key=e.get))
    # ... and the flag matters on the HIP path: an erf-GELU net with these weights misses the fixture's gradients
    cfg2 = mc.ModelCfg(cfg.embed_dim, cfg.vision, cfg.text, None, cfg.init_logit_scale, False)
    n2 = net.SpatialClipNet("custom", None, model_cfg=cfg2, grad_checkpointing=recompute)
    n2.load_state_dict({k[2:]: v for k, v in z.items() if k.startswith("p.")})
    m2 = module.SpatialClipLitModule(n2, losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True), None, None)
    m2.model_step({"images": z["images"].cuda(), "texts": z["texts"].cuda()})["loss"].backward()
    torch.cuda.synchronize()
    e2 = grad_errors(n2)
    assert float(np.median(list(e2.values()))) > 0.018, float(np.median(list(e2.values())))


@pytest.mark.parametrize("fmt", ["safetensors", "pickle", "lightning", "torchscript"])
def test_pretrained_local_file_formats_round_trip(tmp_path, fmt):
    """SpatialClipNet(pretrained=<file>) for every format open_clip.factory.load_state_dict reads (factory.py:153-178): the
    weights of one net, written out, land bit-identically in a second net with another seed -- including a text table saved at
    another context length (resize_text_pos_embed) and an image grid at another resolution (resize_pos_embed)."""
    losses, mc, module, net, ops = _pkg()
    cfg = mc.ModelCfg(embed_dim=32, vision=mc.VisionCfg(32, 8, 64, 2, 32), text=mc.TextCfg(16, 97, 64, 2, 2), gene=None)
    a = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=1)
    sd = {k: v.cpu() for k, v in a.state_dict().items()}
    path = str(tmp_path / {"safetensors": "open_clip_model.safetensors", "pickle": "w.pt", "lightning": "last.ckpt",
                           "torchscript": "ViT-tiny.pt"}[fmt])
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, path)
    elif fmt == "pickle":
        torch.save(sd, path)
    elif fmt == "lightning":
        torch.save({"epoch": 3, "hyper_parameters": {"x": 1}, "state_dict": {"net.model." + k: v for k, v in sd.items()}}, path)
    else:
        from tests.test_cpu_host import _module_tree
        mt = _module_tree({k: (v.half() if v.ndim >= 2 else v) for k, v in sd.items()})          # OpenAI's archives carry fp16 matrices
        for name, val in (("input_resolution", 32), ("context_length", 16), ("vocab_size", 97)):
            mt.register_buffer(name, torch.tensor(val))
        torch.jit.script(mt).save(path)
    b = net.SpatialClipNet("custom", pretrained=path, model_cfg=cfg, seed=2)
    for k, v in b.state_dict().items():
        want = sd[k].half().float() if (fmt == "torchscript" and sd[k].ndim >= 2) else sd[k]
        assert torch.equal(v.cpu(), want), (fmt, k)
    if fmt != "safetensors":
        return
    # another context length + another grid in the file: both tables are resampled at load time
    from spatial_clip_amd.net import resize_pos_embed, resize_text_pos_embed
    big = dict(sd)
    g = torch.Generator().manual_seed(4)
    big["positional_embedding"] = torch.randn(40, 64, generator=g)
    big["visual.positional_embedding"] = torch.randn(1 + 7 * 7, 64, generator=g)
    from safetensors.torch import save_file
    p2 = str(tmp_path / "other_grid.safetensors")
    save_file({k: v.contiguous() for k, v in big.items()}, p2)
    c = net.SpatialClipNet("custom", pretrained=p2, model_cfg=cfg, seed=3)
    want = dict(big)
    resize_pos_embed(want, (4, 4))
    resize_text_pos_embed(want, 16)
    got = c.state_dict()
    assert torch.equal(got["positional_embedding"].cpu(), want["positional_embedding"])
    assert torch.equal(got["visual.positional_embedding"].cpu(), want["visual.positional_embedding"])
    assert got["positional_embedding"].shape == (16, 64) and got["visual.positional_embedding"].shape == (17, 64)
