"""SpatialClipLitModule: the reference's task module (``src/models/spatial_clip_module.py:17-158``) without Lightning.

Same constructor kwargs, same hook names and return contracts: ``forward``, ``model_step`` (signature-filtered
loss kwargs, ``:44,55-61``), ``training_step(batch, batch_idx) -> loss``, ``validation_step``, ``test_step``,
``configure_optimizers() -> {"optimizer", "lr_scheduler": {...}}``.  ``self.log`` / ``self.log_dict`` collect into
``self.logged`` (device scalars, no host sync) for the trainer's logger."""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, Optional

import torch

from . import ops
from .metrics import ContrastiveMetrics
from .net import SpatialClipNet


class _HParams(dict):
    __getattr__ = dict.get


class SpatialClipLitModule(torch.nn.Module):
    def __init__(self, net: SpatialClipNet, loss_fn: torch.nn.Module, optimizer_cfg: Callable, scheduler_cfg: Callable,
                 train_metrics: Optional[ContrastiveMetrics] = None, val_metrics: Optional[ContrastiveMetrics] = None,
                 test_metrics: Optional[ContrastiveMetrics] = None, global_hvg_path: Optional[str] = None):
        super().__init__()
        self.hparams = _HParams(optimizer_cfg=optimizer_cfg, scheduler_cfg=scheduler_cfg,
                                global_hvg_path=global_hvg_path)      # save_hyperparameters(ignore=[net, loss_fn])
        self.net = net
        self.loss_fn = loss_fn
        self.train_metrics = train_metrics or ContrastiveMetrics("train/")
        self.val_metrics = val_metrics or ContrastiveMetrics("val/")
        self.test_metrics = test_metrics or ContrastiveMetrics("test/")
        self.global_hvg_path = global_hvg_path
        self.zero_shot_metric = None            # validation-only, not on the training path
        self.gene_bank_embeddings = None
        self.trainer = None
        self.logged: Dict[str, Any] = {}
        # spatial_clip_module.py:44 -- cache the kwarg names the loss accepts, once
        self._loss_fn_arg_names = set(inspect.signature(self.loss_fn.forward).parameters.keys())

    @property
    def device(self) -> torch.device:
        return self.net.device_

    def log(self, name: str, value, **kw) -> None:
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def log_dict(self, metrics, **kw) -> None:
        self.logged["__metrics__" + metrics.prefix] = metrics

    def forward(self, images: torch.Tensor, texts: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.net(images, texts)

    def model_step(self, batch: Dict[str, Any], metrics: Optional[ContrastiveMetrics] = None) -> Dict[str, torch.Tensor]:
        features = self.forward(batch["images"], batch["texts"])
        available_data = {**features, **batch}
        loss_input = {k: v for k, v in available_data.items() if k in self._loss_fn_arg_names}
        fused_hits = metrics is not None and hasattr(self.loss_fn, "recall_hits")
        if fused_hits:
            metrics._ensure(self.device)
            self.loss_fn.recall_hits = metrics.hits      # R@k hit counting rides on the loss kernels' z matrix
        loss_dict = self.loss_fn(**loss_input)
        if fused_hits:
            metrics.add_hits(features["image_features"].shape[0])
            self.loss_fn.recall_hits = None
        output = {"loss": loss_dict["contrastive_loss"], "image_features": features["image_features"],
                  "text_features": features["text_features"], "logit_scale": features["logit_scale"]}
        return output

    @staticmethod
    def local_logits(output: Dict[str, torch.Tensor]) -> torch.Tensor:
        """``image_features @ text_features.T * logit_scale`` (spatial_clip_module.py:68), on demand."""
        f_i = output["image_features"].detach().contiguous()
        f_t = output["text_features"].detach().contiguous()
        B, D = f_i.shape
        z = torch.empty((B, B), dtype=torch.float32, device=f_i.device)
        ops.sgemm(f_i, D, 1, f_t, D, 1, z, B, B, B, D)
        return z * output["logit_scale"].detach()

    def training_step(self, batch: Dict[str, Any], batch_idx: int) -> torch.Tensor:
        output = self.model_step(batch, self.train_metrics)
        self.log("train/loss", output["loss"], on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        self.log_dict(self.train_metrics, on_step=False, on_epoch=True, sync_dist=True)
        return output["loss"]

    def validation_step(self, batch: Dict[str, Any], batch_idx: int) -> None:
        with torch.no_grad():
            output = self.model_step(batch, self.val_metrics)
        self.log("val/loss", output["loss"], on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)
        self.log_dict(self.val_metrics, on_step=False, on_epoch=True, sync_dist=True)

    def test_step(self, batch: Dict[str, Any], batch_idx: int) -> None:
        with torch.no_grad():
            output = self.model_step(batch, self.test_metrics)
        self.log("test/loss", output["loss"], on_step=False, on_epoch=True, sync_dist=True)
        self.log_dict(self.test_metrics, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self) -> Dict[str, Any]:
        optimizer = self.hparams.optimizer_cfg(params=self.parameters())
        if self.trainer is None:
            return {"optimizer": optimizer}
        if self.trainer.max_steps == -1:
            total_steps = self.trainer.estimated_stepping_batches if self.trainer.max_epochs is not None else 1_000_000
        else:
            total_steps = self.trainer.max_steps
        if total_steps == float("inf") or total_steps == -1:
            total_steps = 1_000_000
        scheduler = self.hparams.scheduler_cfg(optimizer=optimizer, num_training_steps=int(total_steps))
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "monitor": self.hparams.get("optimized_metric", "val/loss"),
                                 "interval": "step", "frequency": 1}}
