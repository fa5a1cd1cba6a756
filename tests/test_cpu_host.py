"""CPU-only checks (no GPU in the build container): the C-ABI library loads and exports every symbol the public
header declares; host-side logic (model registry, parameter layout, schedule, synthetic data) behaves."""
import ctypes
import math
import os

import pytest
import torch

import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import _lib, data, hydra_lite as H, model_configs as mc, optim, params
from oracle import spatial_clip_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    decl = _lib.parse_header()
    assert len(decl) >= 29 and "sc_gemm_bf16" in decl and "sc_attn_bwd" in decl
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(l, name), name
    assert _lib.lib().sc_abi_version() == 1


def test_header_signatures_are_plain_c():
    src = open(os.path.join(ROOT, "include", "spatial_clip_hip.h")).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src
    for name, (_, args) in _lib.parse_header().items():
        assert all(a in (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong) for a in args), name


def test_argument_validation_without_gpu():
    """Shape errors are rejected by the library before any launch (return < 0 + message)."""
    l = _lib.lib()
    rc = l.sc_gemm_bf16(0, 0, None, 8, None, 8, 0, 8, 64, None, 8, None, 0, None, None, 0, None, 0, 1, None, None)
    assert rc < 0 and b"empty problem" in l.sc_last_error()
    rc = l.sc_attn_fwd(None, None, None, 1, 1000, 1, 64, 0, 0, None)
    assert rc < 0 and b"L <=" in l.sc_last_error()
    rc = l.sc_layernorm_fwd(None, 6, None, None, None, 6, None, None, 4, 6, 1e-5, None)
    assert rc < 0


def test_ops_refuse_cpu_tensors():
    from spatial_clip_amd import ops
    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.SpatialClipHipError):
        ops.gemm(ops.NT, ops.EPI_BF16, a, a, a, M=64, N=64, K=64)


def test_net_fails_loudly_without_gpu():
    from spatial_clip_amd import net
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.SpatialClipNet("ViT-B-16-gene", None)


def test_model_registry_matches_reference_manifest(golden_dir):
    import json
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest.json")))
    for name in ("ViT-B-16", "ViT-L-14", "ViT-B-32"):
        cfg = mc.get_model_config(name)
        specs = {s.name: list(s.shape) for s in params.build_specs(cfg)}
        ref = man[name]
        assert set(specs) == set(ref), (set(specs) ^ set(ref))
        for k, shp in ref.items():
            assert specs[k] == shp, k
    with pytest.raises(RuntimeError, match="not found"):
        mc.get_model_config("ViT-nope")
    g = mc.get_model_config("ViT-B-16-gene")
    assert g.text is None and g.gene.n_genes == 20000 and g.vision.heads == 12 and g.vision.tokens == 197


def test_param_layout_alignment_and_order():
    cfg = mc.get_model_config("ViT-Ti-16-gene", n_genes=1000)
    specs = params.build_specs(cfg)
    offs = [s.offset for s in specs]
    assert offs == sorted(offs) and all(o % params.ALIGN == 0 for o in offs)
    for a, b in zip(specs, specs[1:]):
        assert a.offset + a.numel <= b.offset
    assert specs[-1].name == "logit_scale"


def test_cosine_schedule_matches_oracle_and_lambdalr():
    class Opt:
        param_groups = [{"lr": 1e-3, "initial_lr": 1e-3}]
    o = Opt()
    s = optim.get_cosine_schedule_with_warmup(o, num_warmup_steps=5, num_training_steps=50)
    lrs = []
    for step in range(60):
        lrs.append(o.param_groups[0]["lr"])
        s.step()
    assert lrs[0] == 0.0                                        # first optimiser step runs with lr = 0
    for step, lr in enumerate(lrs):
        assert abs(lr - 1e-3 * O.cosine_warmup_lambda(step, 5, 50)) < 1e-12
    assert abs(lrs[5] - 1e-3) < 1e-12 and abs(lrs[50]) < 1e-12


def test_synthetic_batch_contract():
    b = data.synthetic_batch(16, 32, 50, K=8, step=3, rank=1, world_size=2)
    assert b["images"].shape == (16, 3, 32, 32) and b["images"].dtype == torch.float32
    assert b["texts"].shape == (16, 50) and b["texts"].dtype == torch.float32
    assert b["image_tile_ids"].dtype == torch.int64 and torch.equal(b["image_tile_ids"], b["text_tile_ids"])
    assert b["neighbor_tile_ids"].shape == (16, 8) and b["neighbor_alphas"].shape == (16, 8)
    assert int(b["image_tile_ids"][0]) == 10_000 + 16              # rank 1 owns the second contiguous block
    pad = b["neighbor_tile_ids"] < 0
    assert float(b["neighbor_alphas"][pad].abs().sum()) == 0.0
    rows = b["neighbor_alphas"].sum(1)
    assert torch.allclose(rows[rows > 0], torch.ones_like(rows[rows > 0]), atol=1e-6)
    dm = data.SyntheticSpatialDataModule(batch_size=4, image_size=32, n_genes=50)
    with pytest.raises(ValueError):
        dm.setup()


def test_zero_shot_metric_host_targets_match_reference(golden_dir):
    """The host half of ZeroShotGeneExpressionMetric (caption -> rank-weighted vector) against the reference's output."""
    import json
    import numpy as np
    from spatial_clip_amd import metrics
    j = json.load(open(os.path.join(golden_dir, "zero_shot_metric.json")))
    z = np.load(os.path.join(golden_dir, "zero_shot_metric.npz"))
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("\n".join(j["genes"]) + "\n")
    m = metrics.ZeroShotGeneExpressionMetric(global_hvg_path=f.name)
    os.unlink(f.name)
    assert m.num_global_genes == len(j["genes"])
    for i, caps in enumerate(j["captions"]):
        t = m._compute_rank_weighted_vector(caps, "cpu")
        assert torch.equal(t, torch.from_numpy(z[f"targets{i}"]))
    assert metrics.ZeroShotGeneExpressionMetric(global_hvg_path="/nonexistent").num_global_genes == 0
    assert m.compute() == 0.0


def test_resize_pos_embed_matches_reference(golden_dir):
    """Checkpoint interop: position-embedding grid resampling at load time against the reference's
    resize_pos_embed (tests/golden/make_golden_pos_embed.py): down- and up-sampling, and the no-op case."""
    import numpy as np
    from spatial_clip_amd import net
    z = np.load(os.path.join(golden_dir, "pos_embed_resize.npz"))
    for gs in (4, 7, 14):
        sd = {"visual.positional_embedding": torch.from_numpy(z["old"]).clone()}
        net.resize_pos_embed(sd, (gs, gs))
        torch.testing.assert_close(sd["visual.positional_embedding"], torch.from_numpy(z[f"grid{gs}"]), atol=1e-6, rtol=1e-6)
    sd = {"other": torch.zeros(1)}
    net.resize_pos_embed(sd, (4, 4))          # no visual embedding: untouched
    assert list(sd) == ["other"]


def test_trainer_refuses_a_multi_gpu_config_without_a_launcher(monkeypatch):
    """configs/trainer/ddp.yaml (devices: 8, strategy: ddp) started as ONE process must fail loudly, not train a
    single replica that looks like DP8 (reference: Lightning launches the ranks itself, configs/trainer/ddp.yaml:4)."""
    from spatial_clip_amd import trainer as T
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert T._requested_world("auto", 1) is None and T._requested_world(8, 1) == 8
    assert T._requested_world([0, 1, 2], 2) == 6 and T._requested_world("4", 1) == 4
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        T.Trainer(devices=8, strategy="ddp", accelerator="gpu")
    with pytest.raises(ValueError, match="strategy"):
        T.Trainer(strategy="fsdp")
    t = T.Trainer(devices=1, default_root_dir="/tmp/x", enable_checkpointing=True)
    assert t.world_size == 1 and t.checkpoint_callback is not None and t.checkpoint_callback.monitor == "val/R@1"
    cfg = H.compose("train.yaml", ["experiment=vitb16_gene_8gpu"])
    assert cfg.trainer.devices == 8 and cfg.trainer.strategy == "ddp"
    with pytest.raises(RuntimeError, match="8 ranks"):
        H.instantiate(cfg.trainer)


def test_reference_ddp_trainer_config_composes_and_is_enforced(monkeypatch):
    """The reference's own configs/trainer/ddp.yaml (devices: 4) through the composer."""
    ref = "/root/reference/configs"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("PROJECT_ROOT", "/tmp")
    cfg = H.compose("train.yaml", ["trainer=ddp", "logger=null", "callbacks=null"], config_dir=ref)
    assert cfg.trainer.strategy == "ddp" and cfg.trainer.devices == 4 and cfg.trainer.sync_batchnorm is True
    with pytest.raises(RuntimeError, match="4 ranks"):
        H.instantiate(cfg.trainer)


def test_checkpoint_prefixes_and_step_output_contract():
    from spatial_clip_amd import module, net
    assert net.strip_checkpoint_prefix("net.model.visual.conv1.weight") == "visual.conv1.weight"
    assert net.strip_checkpoint_prefix("module.visual.proj") == "visual.proj"
    assert net.strip_checkpoint_prefix("module.net.model.logit_scale") == "logit_scale"
    assert net.strip_checkpoint_prefix("transformer.resblocks.0.ln_1.weight") == "transformer.resblocks.0.ln_1.weight"
    out = module.StepOutput({"loss": 1.0, "image_features": None})
    assert "logits" in out and set(out.keys()) >= {"loss", "logits", "image_features"}     # spatial_clip_module.py:70


def test_gene_transformer_registry_and_parameter_layout():
    """BASELINE configs[4]: ``ViT-L-14-genetr`` = ViT-L/14 image tower + 6-layer gene transformer; the product's flat
    parameter layout and the oracle's init agree on every name and shape."""
    cfg = mc.get_model_config("ViT-L-14-genetr", 20000)
    g = cfg.gene
    assert (g.kind, g.layers, g.width, g.patch, g.tokens, g.heads) == ("transformer", 6, 512, 256, 80, 8)
    assert cfg.vision.width == 1024 and cfg.vision.layers == 24 and cfg.embed_dim == 768 and cfg.text is None
    small = mc.ModelCfg(32, mc.VisionCfg(32, 8, 64, 2, 32), None, mc.GeneCfg(300, 0, "transformer", 64, 64, 2, 32))
    specs = {s.name: tuple(s.shape) for s in params.build_specs(small)}
    ocfg = O.ModelCfg(32, O.VisionCfg(32, 8, 64, 2, 32), None, O.GeneCfg(300, 0, "transformer", 64, 64, 2, 32))
    op = O.init_params(ocfg, 0)
    assert specs == {k: tuple(v.shape) for k, v in op.items()}
    names = [s.name for s in params.build_specs(small)]
    assert names.index("gene.conv1.weight") > names.index("visual.proj") and names[-1] == "logit_scale"
    # the oracle tower: unit-norm output, finite gradients for every gene parameter, padding genes are inert
    x = torch.rand(3, 300)
    p = {k: v.clone().requires_grad_(True) for k, v in op.items()}
    f = O.encode_gene_transformer(x, p, ocfg)
    assert torch.allclose(f.norm(dim=1), torch.ones(3), atol=1e-5)
    f.sum().backward()
    assert all(torch.isfinite(p[k].grad).all() and p[k].grad.abs().sum() > 0 for k in p if k.startswith("gene."))
    assert float(p["gene.conv1.weight"].grad.abs().sum()) > 0


def test_trainer_reads_checkpoint_and_early_stopping_config(tmp_path):
    """cfg.callbacks of the reference (configs/callbacks/default.yaml: model_checkpoint {dirpath, monitor, mode},
    early_stopping {monitor, mode, patience}) is read as configuration; Lightning's EarlyStopping rule on epoch records."""
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd.trainer import Trainer
    cbs = {"model_checkpoint": {"_target_": "lightning.pytorch.callbacks.ModelCheckpoint", "dirpath": str(tmp_path / "ck"),
                                "monitor": "val/loss", "mode": "min", "save_last": True},
           "early_stopping": {"_target_": "lightning.pytorch.callbacks.EarlyStopping", "monitor": "val/R@1", "mode": "max",
                              "patience": 2, "min_delta": 0.01},
           "model_summary": {"max_depth": -1}}
    tr = Trainer(max_epochs=5, enable_checkpointing=True, callbacks=cbs)
    cb = tr.checkpoint_callback
    assert cb.dirpath == str(tmp_path / "ck") and cb.monitor == "val/loss" and cb.mode == "min"
    assert cb.is_better(1.0) and not (setattr(cb, "best_model_score", 0.5) or cb.is_better(0.7))
    seq = [0.10, 0.20, 0.205, 0.19, 0.5]                       # +0.005 and -0.015 do not beat min_delta: stop at the 4th
    stops = [tr._early_stop({"val/R@1": v}) for v in seq[:4]]
    assert stops == [False, False, False, True] and tr.should_stop
    assert not Trainer(max_epochs=1)._early_stop({"val/R@1": 0.1})                       # no early_stopping configured
    assert Trainer(max_epochs=1, enable_checkpointing=True).checkpoint_callback is None  # nowhere to write
    d = Trainer(max_epochs=1, enable_checkpointing=True, default_root_dir=str(tmp_path)).checkpoint_callback
    assert d.dirpath == str(tmp_path / "checkpoints") and d.monitor == "val/R@1" and d.mode == "max"


# ---------------------------------------------------------------------------------------------- round 5: checkpoint interop (f2)
def test_resize_text_pos_embed_matches_reference(golden_dir):
    """src/open_clip/model.py:826-860 on the fixture written by the reference's own function (make_golden_r5.py)."""
    import numpy as np
    from spatial_clip_amd import net
    z = np.load(os.path.join(golden_dir, "text_pos_resize.npz"))
    for n in (32, 128, 77):
        sd = {"positional_embedding": torch.from_numpy(z["old"]).clone()}
        net.resize_text_pos_embed(sd, n)
        assert sd["positional_embedding"].shape == (n, 24)
        assert torch.allclose(sd["positional_embedding"], torch.from_numpy(z[f"new_{n}"]), atol=1e-6, rtol=0)
    sd = {"text.positional_embedding": torch.from_numpy(z["old"]).clone()}          # custom-text key layout
    net.resize_text_pos_embed(sd, 32)
    assert torch.allclose(sd["text.positional_embedding"], torch.from_numpy(z["new_32"]), atol=1e-6, rtol=0)


def _module_tree(sd):
    root = torch.nn.Module()
    for k, v in sd.items():
        parts = k.split(".")
        m = root
        for p_ in parts[:-1]:
            if not hasattr(m, p_):
                m.add_module(p_, torch.nn.Module())
            m = getattr(m, p_)
        m.register_parameter(parts[-1], torch.nn.Parameter(v.clone()))
    return root


def test_checkpoint_file_formats_of_the_reference_loader(tmp_path):
    """open_clip.factory.load_state_dict (src/open_clip/factory.py:153-178): .safetensors, pickled state_dict, pickled
    {'state_dict': ...} training checkpoint, and a TorchScript archive (state_dict() minus input_resolution / context_length
    / vocab_size) all come back as the same host tensors."""
    from safetensors.torch import save_file
    from spatial_clip_amd import net
    g = torch.Generator().manual_seed(0)
    sd = {"visual.conv1.weight": torch.randn(8, 3, 4, 4, generator=g), "visual.transformer.resblocks.0.ln_1.weight": torch.randn(8, generator=g),
          "positional_embedding": torch.randn(16, 8, generator=g), "logit_scale": torch.tensor(2.6593)}
    f1 = str(tmp_path / "w.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, f1)
    f2 = str(tmp_path / "w.pt")
    torch.save(sd, f2)
    f3 = str(tmp_path / "epoch_1.pt")
    torch.save({"epoch": 1, "name": "run", "state_dict": {"module." + k: v for k, v in sd.items()}}, f3)
    m = _module_tree(sd)
    for name, val in (("input_resolution", 224), ("context_length", 77), ("vocab_size", 49408)):
        m.register_buffer(name, torch.tensor(val))
    f4 = str(tmp_path / "openai_jit.pt")
    torch.jit.script(m).save(f4)
    for path in (f1, f2, f4):
        got = net.read_checkpoint_file(path)
        assert set(got) == set(sd), (path, sorted(got))
        for k in sd:
            assert torch.equal(got[k].float(), sd[k]), (path, k)
    got = net.read_checkpoint_file(f3)
    inner = {net.strip_checkpoint_prefix(k): v for k, v in got["state_dict"].items()}
    assert set(inner) == set(sd) and all(torch.equal(inner[k], sd[k]) for k in sd)


class _NotATensor:          # module-level so that pickle can name it
    def __init__(self):
        self.x = 1


def test_checkpoint_reader_refuses_arbitrary_pickles_unless_asked(tmp_path, monkeypatch):
    """The reference's loader never unpickles arbitrary objects by default (factory.py:153-178 keeps weights_only=True); a file
    that pickles more than tensors is refused with the original error, and read only on the explicit opt-in."""
    from spatial_clip_amd import net
    f = str(tmp_path / "lightning_like.ckpt")
    torch.save({"state_dict": {"logit_scale": torch.tensor(1.0)}, "callbacks": _NotATensor()}, f)
    monkeypatch.delenv("SC_UNSAFE_PICKLE", raising=False)
    with pytest.raises(RuntimeError, match="SC_UNSAFE_PICKLE"):
        net.read_checkpoint_file(f)
    got = net.read_checkpoint_file(f, unsafe_pickle=True)
    assert float(got["state_dict"]["logit_scale"]) == 1.0
    monkeypatch.setenv("SC_UNSAFE_PICKLE", "1")
    assert "state_dict" in net.read_checkpoint_file(f)
    sd = {"positional_embedding": torch.zeros(16, 8)}
    with pytest.raises(AssertionError, match="width changed"):
        net.resize_text_pos_embed(sd, 32, model_width=12)


def test_quickgelu_registry_entries():
    """model_configs/ViT-B-16-quickgelu.json, ViT-B-32-quickgelu.json, ViT-L-14-quickgelu.json: the base architecture with
    `quick_gelu: true`; the gene variants keep the flag for the image tower."""
    from spatial_clip_amd import model_configs as mc
    for base in ("ViT-B-16", "ViT-B-32", "ViT-L-14"):
        a, b = mc.get_model_config(base), mc.get_model_config(base + "-quickgelu")
        assert not a.quick_gelu and b.quick_gelu
        assert a.vision == b.vision and a.text == b.text and a.embed_dim == b.embed_dim
        assert mc.get_model_config(base + "-quickgelu-gene").quick_gelu
    assert "ViT-B-16-quickgelu" in mc.list_models()
