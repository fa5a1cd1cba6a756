"""Entry point with the reference's shape (``src/train.py:44-172``): compose config -> seed -> instantiate
datamodule / model / trainer -> handshake preprocess_fn + tokenizer -> fit -> test.

    python -m spatial_clip_amd.train experiment=smoke_shards trainer.max_epochs=1 [--config-dir /path/to/configs]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m spatial_clip_amd.train experiment=vitb16_gene_8gpu          (one process per GPU, RCCL over xGMI)
"""
from __future__ import annotations

import sys
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import comm, hydra_lite
from .trainer import _requested_world


def train(cfg) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    # One process per GPU (reference: Lightning's DDP strategy re-launches src/train.py per device; here the launcher is
    # torch.distributed.run).  Bind to cuda:LOCAL_RANK and join the RCCL group BEFORE the model allocates anything;
    # a trainer config that asks for more ranks than the launcher started raises (no silent single-GPU "DP8").
    tcfg = cfg.get("trainer") or {}
    comm.init_from_env(expect_world=_requested_world(tcfg.get("devices", "auto"), tcfg.get("num_nodes", 1))
                       if comm.env_world()[2] > 1 else None)
    if cfg.get("seed") is not None:
        torch.manual_seed(int(cfg.seed))                       # L.seed_everything (src/train.py:56-57)
    datamodule = hydra_lite.instantiate(cfg.data)
    # precision is a Trainer key in the reference (configs/trainer/default.yaml:15); here the GEMM operand type shapes the
    # net's weight copies, so an fp8 trainer precision is handed to the net before it is built
    if str(tcfg.get("precision", "bf16-mixed")).startswith("fp8") and isinstance(cfg.get("model", {}).get("net"), dict):
        cfg.model["net"].setdefault("precision", "fp8")
    model = hydra_lite.instantiate(cfg.model)
    model.hparams["optimized_metric"] = cfg.get("optimized_metric", "val/loss")
    datamodule.preprocess_fn = model.net.preprocess_train      # handshake, src/train.py:70-73
    datamodule.tokenizer = model.net.tokenizer
    if "save_ckpt" in cfg and isinstance(cfg.get("trainer"), dict):
        cfg.trainer["enable_checkpointing"] = bool(cfg.save_ckpt)          # src/train.py:92-99
    # callbacks stay configuration (model_checkpoint / early_stopping keys of configs/callbacks/*.yaml), read by the Trainer
    cbs = cfg.get("callbacks")
    trainer = hydra_lite.instantiate(cfg.trainer, callbacks=dict(cbs) if isinstance(cbs, dict) else None)
    objects = {"cfg": cfg, "datamodule": datamodule, "model": model, "trainer": trainer}
    metrics: Dict[str, Any] = {}
    if cfg.get("train", True):
        trainer.fit(model=model, datamodule=datamodule, ckpt_path=cfg.get("ckpt_path"))
        metrics.update(getattr(trainer, "callback_metrics", {}))
    if cfg.get("test", False):
        ckpt_path = None                                         # src/train.py:126-133: best checkpoint if one exists
        if trainer.checkpoint_callback is not None and trainer.checkpoint_callback.best_model_path:
            ckpt_path = trainer.checkpoint_callback.best_model_path
        datamodule.preprocess_fn = model.net.preprocess_val     # test-time handshake, src/train.py:136-138
        out = trainer.test(model=model, datamodule=datamodule, ckpt_path=ckpt_path)
        if out:
            metrics.update(out[0])
    return metrics, objects


def main(argv: Optional[List[str]] = None) -> Dict[str, Any]:
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir = None
    if "--config-dir" in argv:
        i = argv.index("--config-dir")
        config_dir = argv[i + 1]
        del argv[i:i + 2]
    cfg = hydra_lite.compose("train.yaml", argv, config_dir=config_dir)
    metrics, _ = train(cfg)
    if comm.world()[0] == 0:
        print({k: (round(v, 5) if isinstance(v, float) else v) for k, v in metrics.items()})
    comm.shutdown()
    return metrics


if __name__ == "__main__":
    main()
