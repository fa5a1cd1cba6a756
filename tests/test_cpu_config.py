"""Hydra-surface tests (SURVEY.md section 4, item v): every experiment composes, interpolations resolve, reference
``_target_`` paths map onto this package's classes; the reference's own config tree composes too when present."""
import glob
import os

import pytest

import spatial_clip_amd  # noqa: F401
from spatial_clip_amd import hydra_lite as H

REF_CONFIGS = "/root/reference/configs"


def test_default_composition_and_interpolation(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    cfg = H.compose("train.yaml", [])
    assert cfg.model._target_ == "src.models.spatial_clip_module.SpatialClipLitModule"
    assert cfg.model.loss_fn._target_ == "src.models.components.losses.SpatialLoss"
    assert cfg.model.loss_fn.cap_logit_scale == 40.0 and cfg.model.loss_fn.temp_reg_weight == 0.05
    assert cfg.model.optimizer_cfg._partial_ is True and cfg.model.optimizer_cfg.betas == [0.9, 0.98]
    assert cfg.model.net.n_genes == cfg.data.n_genes == 20000
    assert cfg.data.data_dir == "/tmp/proj/data/"
    assert cfg.trainer.precision == "bf16-mixed" and cfg.trainer.gradient_clip_val == 1.0 and cfg.seed == 42


def test_overrides(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    cfg = H.compose("train.yaml", ["loss=clip", "trainer=ddp", "optimizer.lr=1e-4", "data.batch_size=32",
                                   "+trainer.max_steps=10", "model.net.model_name=ViT-L-14-gene", "~tags"])
    assert cfg.model.loss_fn._target_.endswith("ClipLoss") and "cap_logit_scale" not in cfg.model.loss_fn
    assert cfg.trainer.strategy == "ddp" and cfg.trainer.devices == 8 and cfg.trainer.precision == "bf16-mixed"
    assert cfg.model.optimizer_cfg.lr == 1e-4 and cfg.data.batch_size == 32 and cfg.trainer.max_steps == 10
    assert cfg.model.net.model_name == "ViT-L-14-gene" and "tags" not in cfg


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(H.OWN_CONFIG_DIR, "experiment", "*.yaml"))))
def test_every_own_experiment_composes(path, monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    name = os.path.splitext(os.path.basename(path))[0]
    cfg = H.compose("train.yaml", [f"experiment={name}"])
    assert "_target_" in cfg.model and "_target_" in cfg.trainer and "_target_" in cfg.data
    if name == "smoke_shards":
        assert cfg.data.batch_size == 8 and cfg.trainer.fast_dev_run is True and cfg.model.net.model_name == "ViT-Ti-16-gene"
    if name == "spatial_v1":
        assert cfg.model.optimizer_cfg.lr == 1e-4 and cfg.model.scheduler_cfg.num_warmup_steps == 1000


def test_instantiate_partials_and_targets(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    cfg = H.compose("train.yaml", ["experiment=smoke_shards"])
    from spatial_clip_amd import data, losses, optim, trainer
    dm = H.instantiate(cfg.data)
    assert isinstance(dm, data.SyntheticSpatialDataModule) and dm.batch_size == 8
    loss = H.instantiate(cfg.loss)
    assert isinstance(loss, losses.SpatialLoss) and loss.neighbor_alpha_scale == 0.5
    opt_partial = H.instantiate(cfg.optimizer)
    assert opt_partial.func is optim.FusedAdamW and opt_partial.keywords["weight_decay"] == 0.1
    sched_partial = H.instantiate(cfg.scheduler)
    assert sched_partial.func is optim.get_cosine_schedule_with_warmup
    tr = H.instantiate(cfg.trainer)
    assert isinstance(tr, trainer.Trainer) and tr.fast_dev_run is True and tr.gradient_clip_val == 1.0
    with pytest.raises(RuntimeError):
        H.instantiate({"_target_": "lightning.pytorch.Trainer", "accelerator": "cpu"})


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("exp", ["smoke_shards", "medium_normal", "medium_spatial", "spatial_v1",
                                 "spatial_v2_multi_gpu", "smoke_hugo", "compare_hugo_overlap", "compare_medium_overlap",
                                 "smoke_multitech", "spatial_v3_multigpu_moreepochs"])
def test_reference_config_tree_composes(exp, monkeypatch):
    """The reference's OWN configs/ directory drives the composer; its missing data group falls back to ours.  All ten
    files of the reference's configs/experiment/ (the parametrisation is checked against the directory listing)."""
    have = sorted(f[:-5] for f in os.listdir(os.path.join(REF_CONFIGS, "experiment")) if f.endswith(".yaml"))
    assert exp in have and len(have) == 10, have
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    cfg = H.compose("train.yaml", [f"experiment={exp}", "logger=csv"], config_dir=REF_CONFIGS)
    assert cfg.model._target_ == "src.models.spatial_clip_module.SpatialClipLitModule"
    assert cfg.model.net._target_ == "src.models.components.spatial_clip_net.SpatialClipNet"
    assert cfg.model.loss_fn._target_ in ("src.models.components.losses.SpatialLoss",
                                          "src.models.components.losses.ClipLoss")
    if exp == "medium_normal":
        assert cfg.model.loss_fn._target_.endswith("ClipLoss") and cfg.data.batch_size == 32
    if exp == "spatial_v2_multi_gpu":
        assert cfg.trainer.devices == 2 and cfg.data.batch_size == 1500
    assert H._locate(cfg.model.loss_fn._target_).__module__ == "spatial_clip_amd.losses"
    assert H._locate(cfg.trainer._target_).__name__ == "Trainer"


def test_eval_recipe_composes(monkeypatch):
    """configs/eval.yaml (the reference's evaluation recipe, configs/eval.yaml: task_name eval, mandatory ckpt_path)."""
    monkeypatch.setenv("PROJECT_ROOT", "/tmp")
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import hydra_lite
    cfg = hydra_lite.compose("eval.yaml", ["experiment=smoke_shards", "ckpt_path=/tmp/x.ckpt"])
    assert cfg.task_name == "eval" and cfg.ckpt_path == "/tmp/x.ckpt"
    assert cfg.model.net.model_name == "ViT-Ti-16-gene" and cfg.data.batch_size == 8


def test_configs4_experiment_carries_precision_and_recompute_keys(monkeypatch):
    """BASELINE configs[4] on the Hydra surface: model, per-GPU batch, fp8 trainer precision, activation recomputation."""
    monkeypatch.setenv("PROJECT_ROOT", "/tmp/proj")
    cfg = H.compose("train.yaml", ["experiment=vitl14_genetr_fp8_8gpu"])
    assert cfg.model.net.model_name == "ViT-L-14-genetr" and cfg.model.net.grad_checkpointing is True
    assert cfg.trainer.precision == "fp8-mixed" and cfg.trainer.devices == 8 and cfg.trainer.strategy == "ddp"
    assert cfg.data.batch_size == 1024 and cfg.data.k_neighbors == 8
    assert cfg.model.loss_fn._target_.endswith("SpatialLoss")
    one = H.compose("train.yaml", ["experiment=vitl14_genetr_b256", "trainer.precision=fp8-mixed"])
    assert one.trainer.devices == 1 and one.data.batch_size == 256 and one.trainer.precision == "fp8-mixed"
    from spatial_clip_amd import trainer
    tr = H.instantiate(one.trainer)
    assert isinstance(tr, trainer.Trainer) and tr.precision == "fp8"
    assert H.instantiate(H.compose("train.yaml", ["experiment=vitl14_genetr_b256"]).trainer).precision == "bf16"
    with pytest.raises(ValueError):
        H.instantiate({"_target_": "lightning.pytorch.Trainer", "precision": "16-mixed"})
