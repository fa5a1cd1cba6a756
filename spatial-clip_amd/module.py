"""SpatialClipLitModule: the reference's task module (``src/models/spatial_clip_module.py:17-158``) without Lightning.

Same constructor kwargs, same hook names and return contracts: ``forward``, ``model_step`` (signature-filtered
loss kwargs, ``:44,55-61``), ``training_step(batch, batch_idx) -> loss``, ``validation_step``, ``test_step``,
``configure_optimizers() -> {"optimizer", "lr_scheduler": {...}}``.  ``self.log`` / ``self.log_dict`` collect into
``self.logged`` (device scalars, no host sync) for the trainer's logger."""
from __future__ import annotations

import inspect
import os
from typing import Any, Callable, Dict, Optional

import torch

from . import comm, ops
from .metrics import ContrastiveMetrics, ZeroShotGeneExpressionMetric
from .net import SpatialClipNet


class _HParams(dict):
    __getattr__ = dict.get


class StepOutput(dict):
    """``model_step`` result with the reference's keys (``loss``, ``logits``, ``image_features``;
    spatial_clip_module.py:66-70).  ``logits`` -- the local [B,B] ``image_features @ text_features.T * logit_scale``
    the reference computes for its metrics -- is materialised on first access: R@k already rides on the loss
    kernels' similarity matrix here, so the training step itself never needs it."""

    _LAZY = "logits"

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        dict.__setitem__(self, self._LAZY, None)

    def _materialise(self):
        v = dict.__getitem__(self, self._LAZY)
        if v is None:
            v = SpatialClipLitModule.local_logits(self)
            dict.__setitem__(self, self._LAZY, v)
        return v

    def __getitem__(self, k):
        return self._materialise() if k == self._LAZY else dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


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
        # validation-only zero-shot path (spatial_clip_module.py:36-41)
        self.zero_shot_metric = ZeroShotGeneExpressionMetric(global_hvg_path=global_hvg_path) if global_hvg_path else None
        self.gene_bank_embeddings = None
        self.trainer = None
        self._feature_gather = None
        self.logged: Dict[str, Any] = {}
        self.synced = set()                    # names logged with sync_dist=True
        # spatial_clip_module.py:44 -- cache the kwarg names the loss accepts, once
        self._loss_fn_arg_names = set(inspect.signature(self.loss_fn.forward).parameters.keys())

    @property
    def device(self) -> torch.device:
        return self.net.device_

    def log(self, name: str, value, sync_dist: bool = False, **kw) -> None:
        """Collects into ``self.logged`` (device scalars, no host sync).  ``sync_dist=True`` (the reference sets it on
        every loss it logs, spatial_clip_module.py:105,107,113,121) marks the name in ``self.synced``: the trainer
        averages those values over the ranks when it turns them into host numbers (``comm.all_reduce_mean_scalars``,
        one collective per record)."""
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value
        if sync_dist:
            self.synced.add(name)

    def log_dict(self, metrics, **kw) -> None:
        self.logged["__metrics__" + metrics.prefix] = metrics

    def forward(self, images: torch.Tensor, texts: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.net(images, texts)

    def model_step(self, batch: Dict[str, Any], metrics: Optional[ContrastiveMetrics] = None) -> Dict[str, torch.Tensor]:
        # W > 1: let the net launch the feature all-gathers from inside its forward (text side under the vision tower,
        # image side under the first similarity GEMM); the loss picks the gathered tensors up by identity
        fg = None
        if comm.is_dist() and hasattr(self.loss_fn, "prefetched") and os.environ.get("SC_GATHER_OVERLAP", "1") != "0":
            fg = self._feature_gather
            if fg is None:
                fg = self._feature_gather = comm.FeatureGather(self.device)
            want_ids = "image_tile_ids" in self._loss_fn_arg_names
            fg.begin(batch.get("image_tile_ids") if want_ids else None, batch.get("text_tile_ids") if want_ids else None)
        self.net.feature_gather = fg
        if hasattr(self.loss_fn, "prefetched"):
            self.loss_fn.prefetched = fg
        features = self.forward(batch["images"], batch["texts"])
        self.net.feature_gather = None
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
        return StepOutput({"loss": loss_dict["contrastive_loss"], "image_features": features["image_features"],
                           "text_features": features["text_features"], "logit_scale": features["logit_scale"]})

    @staticmethod
    def local_logits(output: Dict[str, torch.Tensor]) -> torch.Tensor:
        """``image_features @ text_features.T * logit_scale`` (spatial_clip_module.py:68), on demand."""
        f_i = dict.__getitem__(output, "image_features").detach().contiguous()
        f_t = dict.__getitem__(output, "text_features").detach().contiguous()
        B, D = f_i.shape
        z = torch.empty((B, B), dtype=torch.float32, device=f_i.device)
        ops.sgemm(f_i, D, 1, f_t, D, 1, z, B, B, B, D)
        return z * dict.__getitem__(output, "logit_scale").detach()

    def root_gradient(self, loss: torch.Tensor) -> torch.Tensor:
        """The 1.0 that ``loss.backward()`` would materialise with an ATen fill on every step, kept resident instead:
        ``loss.backward(module.root_gradient(loss))`` leaves no ATen kernel in the training step."""
        g = getattr(self, "_root_grad", None)
        if g is None or g.device != loss.device or g.dtype != loss.dtype:
            g = self._root_grad = torch.ones((), dtype=loss.dtype, device=loss.device)
        return g

    def training_step(self, batch: Dict[str, Any], batch_idx: int) -> torch.Tensor:
        output = self.model_step(batch, self.train_metrics)
        self.log("train/loss", output["loss"], on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        self.log_dict(self.train_metrics, on_step=False, on_epoch=True, sync_dist=True)
        return output["loss"]

    def on_validation_start(self) -> None:
        """Build the gene bank once: text-tower embeddings of every gene name in ``global_hvg_path``
        (spatial_clip_module.py:73-103).  ``net.tokenizer`` must turn a list of gene names into token ids; the
        built-in one does not tokenise strings (BPE is CPU-side data preparation), so a data module that wants this
        metric installs its tokenizer on the net, as the reference's data module does."""
        if not self.zero_shot_metric or self.gene_bank_embeddings is not None:
            return
        path = self.global_hvg_path
        if not path or not os.path.exists(path):
            print(f"Warning: Global HVG path {path} not found.")
            return
        with open(path, "r") as f:
            gene_list = [line.strip() for line in f if line.strip()]
        if not gene_list:
            return
        embs = []
        with torch.no_grad():
            for i in range(0, len(gene_list), 256):
                tokens = self.net.tokenizer(gene_list[i:i + 256])
                tokens = tokens.to(self.device) if isinstance(tokens, torch.Tensor) else torch.as_tensor(tokens).to(self.device)
                embs.append(self.net.model.encode_text(tokens, normalize=True).float())
        self.gene_bank_embeddings = torch.cat(embs, dim=0).contiguous()

    def _zero_shot_update(self, batch: Dict[str, Any], output: Dict[str, torch.Tensor], name: str) -> None:
        if self.zero_shot_metric and self.gene_bank_embeddings is not None and "raw_text" in batch:
            f_i = output["image_features"].detach().float().contiguous()
            bank = self.gene_bank_embeddings
            B, D = f_i.shape
            n = bank.shape[0]
            logits = torch.empty((B, n), dtype=torch.float32, device=f_i.device)     # image_features @ bank.T
            ops.sgemm(f_i, D, 1, bank, D, 1, logits, n, B, n, D)
            self.zero_shot_metric.update(logits, batch["raw_text"])
            self.log(name, self.zero_shot_metric, on_step=False, on_epoch=True, sync_dist=True)

    def validation_step(self, batch: Dict[str, Any], batch_idx: int) -> None:
        with torch.no_grad():
            output = self.model_step(batch, self.val_metrics)
        self.log("val/loss", output["loss"], on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)
        self.log_dict(self.val_metrics, on_step=False, on_epoch=True, sync_dist=True)
        self._zero_shot_update(batch, output, "val/zero_shot_pcc")

    def test_step(self, batch: Dict[str, Any], batch_idx: int) -> None:
        with torch.no_grad():
            output = self.model_step(batch, self.test_metrics)
        self.log("test/loss", output["loss"], on_step=False, on_epoch=True, sync_dist=True)
        self.log_dict(self.test_metrics, on_step=False, on_epoch=True, sync_dist=True)
        self._zero_shot_update(batch, output, "test/zero_shot_pcc")

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
