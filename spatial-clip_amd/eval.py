"""Evaluation entry point with the reference's shape (``src/eval.py:22-73``, ``configs/eval.yaml``): compose config ->
instantiate datamodule / model / trainer -> ``trainer.test(model, datamodule, ckpt_path=cfg.ckpt_path)`` -> the
trainer's metrics.  ``ckpt_path`` is mandatory, as in the reference.

    python -m spatial_clip_amd.eval experiment=smoke_shards ckpt_path=/path/to/checkpoints/last.ckpt
"""
from __future__ import annotations

import sys
from typing import Any, Dict, List, Optional, Tuple

from . import comm, hydra_lite
from .trainer import _requested_world


def evaluate(cfg) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    if not cfg.get("ckpt_path"):
        raise ValueError("evaluate: cfg.ckpt_path is required (configs/eval.yaml: `ckpt_path: ???`)")
    tcfg = cfg.get("trainer") or {}
    comm.init_from_env(expect_world=_requested_world(tcfg.get("devices", "auto"), tcfg.get("num_nodes", 1))
                       if comm.env_world()[2] > 1 else None)
    datamodule = hydra_lite.instantiate(cfg.data)
    model = hydra_lite.instantiate(cfg.model)
    datamodule.preprocess_fn = model.net.preprocess_val        # the test-time handshake of src/train.py:136-138
    datamodule.tokenizer = model.net.tokenizer
    trainer = hydra_lite.instantiate(cfg.trainer)
    objects = {"cfg": cfg, "datamodule": datamodule, "model": model, "trainer": trainer}
    trainer.test(model=model, datamodule=datamodule, ckpt_path=cfg.ckpt_path)
    return dict(trainer.callback_metrics), objects


def main(argv: Optional[List[str]] = None) -> Dict[str, Any]:
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir = None
    if "--config-dir" in argv:
        i = argv.index("--config-dir")
        config_dir = argv[i + 1]
        del argv[i:i + 2]
    cfg = hydra_lite.compose("eval.yaml", argv, config_dir=config_dir)
    metrics, _ = evaluate(cfg)
    if comm.world()[0] == 0:
        print({k: (round(v, 5) if isinstance(v, float) else v) for k, v in metrics.items()})
    comm.shutdown()
    return metrics


if __name__ == "__main__":
    main()
