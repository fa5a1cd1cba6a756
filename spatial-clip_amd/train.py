"""Entry point with the reference's shape (``src/train.py:44-172``): compose config -> seed -> instantiate
datamodule / model / trainer -> handshake preprocess_fn + tokenizer -> fit -> test.

    python -m spatial_clip_amd.train experiment=smoke_shards trainer.max_epochs=1 [--config-dir /path/to/configs]
"""
from __future__ import annotations

import sys
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import hydra_lite


def train(cfg) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    if cfg.get("seed") is not None:
        torch.manual_seed(int(cfg.seed))                       # L.seed_everything (src/train.py:56-57)
    datamodule = hydra_lite.instantiate(cfg.data)
    model = hydra_lite.instantiate(cfg.model)
    model.hparams["optimized_metric"] = cfg.get("optimized_metric", "val/loss")
    datamodule.preprocess_fn = model.net.preprocess_train      # handshake, src/train.py:70-73
    datamodule.tokenizer = model.net.tokenizer
    trainer = hydra_lite.instantiate(cfg.trainer)
    objects = {"cfg": cfg, "datamodule": datamodule, "model": model, "trainer": trainer}
    metrics: Dict[str, Any] = {}
    if cfg.get("train", True):
        trainer.fit(model=model, datamodule=datamodule, ckpt_path=cfg.get("ckpt_path"))
        metrics.update(getattr(trainer, "callback_metrics", {}))
    if cfg.get("test", False):
        out = trainer.test(model=model, datamodule=datamodule)
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
    print({k: (round(v, 5) if isinstance(v, float) else v) for k, v in metrics.items()})
    return metrics


if __name__ == "__main__":
    main()
