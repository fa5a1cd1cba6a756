"""Minimal Lightning-free trainer honouring the ``lightning.pytorch.Trainer`` kwargs the reference's configs use
(configs/trainer/*.yaml): max_epochs / max_steps, precision ("bf16-mixed" is what the kernels implement),
gradient_clip_val, limit_*_batches, fast_dev_run, devices / strategy (one process per GPU launched by torchrun;
gradient reduction is this package's bucketed RCCL all-reduce, not a DDPStrategy).  Loop semantics per step
(SURVEY.md 3.2): training_step -> loss.backward() -> grad all-reduce (mean) -> clip_grad_norm_ -> AdamW.step ->
scheduler.step()."""
from __future__ import annotations

import math
import time
from typing import Any, Dict, Optional

import torch

from . import comm, streams


def _to_device(batch: Dict[str, Any], device) -> Dict[str, Any]:
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


class Trainer:
    def __init__(self, max_epochs: Optional[int] = 1, min_epochs: int = 1, max_steps: int = -1, accelerator: str = "auto",
                 devices: Any = "auto", precision: str = "bf16-mixed", gradient_clip_val: Optional[float] = None,
                 check_val_every_n_epoch: int = 1, log_every_n_steps: int = 50, fast_dev_run: Any = False,
                 limit_train_batches: float = 1.0, limit_val_batches: float = 1.0, limit_test_batches: float = 1.0,
                 deterministic: bool = False, strategy: str = "auto", num_nodes: int = 1, sync_batchnorm: bool = False,
                 default_root_dir: Optional[str] = None, callbacks=None, logger=None, **unused):
        if precision not in ("bf16-mixed", "bf16", "32", "32-true", 32):
            raise ValueError(f"precision {precision!r}: the HIP path computes bf16 GEMMs with fp32 accumulation")
        if accelerator == "cpu":
            raise RuntimeError("accelerator=cpu: this build has no CPU path (the oracle under oracle/ is test-only)")
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.gradient_clip_val = gradient_clip_val
        self.fast_dev_run = fast_dev_run
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.log_every_n_steps = log_every_n_steps
        self.global_step = 0
        self.estimated_stepping_batches = None
        self.history = []

    def _limit(self, n: int, frac) -> int:
        if self.fast_dev_run:
            return 1 if self.fast_dev_run is True else int(self.fast_dev_run)
        return n if frac == 1.0 else (int(frac) if frac > 1 else max(1, int(n * frac)))

    def fit(self, model, datamodule, ckpt_path: Optional[str] = None) -> None:
        datamodule.preprocess_fn = getattr(datamodule, "preprocess_fn", None) or model.net.preprocess_train
        datamodule.tokenizer = getattr(datamodule, "tokenizer", None) or model.net.tokenizer
        datamodule.setup("fit")
        train = datamodule.train_dataloader()
        n_train = self._limit(len(train), self.limit_train_batches)
        epochs = 1 if self.fast_dev_run else (self.max_epochs or 1)
        self.estimated_stepping_batches = n_train * epochs if self.max_steps == -1 else self.max_steps
        model.trainer = self
        cfg = model.configure_optimizers()
        opt = cfg["optimizer"]
        sched = cfg.get("lr_scheduler", {}).get("scheduler")
        rank, W = comm.world()
        reducer = comm.GradBucketReducer(model.net.store.grad)
        model.net.grad_bucket_hook = reducer.bucket_ready if W > 1 else None
        for epoch in range(epochs):
            model.train_metrics.reset()
            t0 = time.time()
            for i, batch in enumerate(train):
                if i >= n_train or (self.max_steps != -1 and self.global_step >= self.max_steps):
                    break
                batch = _to_device(batch, model.device)
                with streams.chain_stream():
                    loss = model.training_step(batch, i)
                    loss.backward()
                    reducer.finish()
                    opt.step(grad_scale=1.0 / W, max_norm=self.gradient_clip_val)
                if sched is not None:
                    sched.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or self.fast_dev_run:
                    self.history.append({"step": self.global_step, "train/loss": float(loss.detach())})
            torch.cuda.synchronize()
            rec = {"epoch": epoch, "time_s": time.time() - t0, **model.train_metrics.compute()}
            if self.history:
                rec["train/loss"] = self.history[-1]["train/loss"]
            val = datamodule.val_dataloader()
            n_val = self._limit(len(val), self.limit_val_batches)
            model.val_metrics.reset()
            model.on_validation_start()
            if model.zero_shot_metric:
                model.zero_shot_metric.reset()
            vl = []
            for i, batch in enumerate(val):
                if i >= n_val:
                    break
                model.validation_step(_to_device(batch, model.device), i)
                vl.append(model.logged["val/loss"])
            if vl:
                rec["val/loss"] = float(torch.stack(vl).mean())
                rec.update(model.val_metrics.compute())
                if model.zero_shot_metric and model.gene_bank_embeddings is not None:
                    rec["val/zero_shot_pcc"] = model.zero_shot_metric.compute()
            self.history.append(rec)
        self.callback_metrics = {k: v for k, v in self.history[-1].items() if isinstance(v, float)} if self.history else {}

    # ------------------------------------------------------------------ checkpoints (reference state_dict names)
    @staticmethod
    def save_checkpoint(path: str, model, optimizer=None, scheduler=None, global_step: int = 0) -> None:
        """``state_dict`` uses the reference ``CLIP.state_dict()`` key names, so the file loads into open_clip too."""
        ck = {"state_dict": {k: v.cpu() for k, v in model.net.state_dict().items()}, "global_step": global_step}
        if optimizer is not None:
            ck["optimizer"] = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in optimizer.state_dict().items()}
        if scheduler is not None:
            ck["scheduler_last_epoch"] = scheduler.last_epoch
        torch.save(ck, path)

    @staticmethod
    def load_checkpoint(path: str, model, optimizer=None, scheduler=None) -> int:
        ck = torch.load(path, map_location="cpu")
        model.net.load_state_dict(ck["state_dict"])
        if optimizer is not None and "optimizer" in ck:
            optimizer.load_state_dict({k: (v.to(model.device) if isinstance(v, torch.Tensor) else v)
                                       for k, v in ck["optimizer"].items()})
        if scheduler is not None and "scheduler_last_epoch" in ck:
            scheduler.last_epoch = int(ck["scheduler_last_epoch"])
            scheduler._apply()
        return int(ck.get("global_step", 0))

    def test(self, model, datamodule, ckpt_path: Optional[str] = None):
        loader = datamodule.test_dataloader()
        n = self._limit(len(loader), 1.0)
        model.test_metrics.reset()
        tl = []
        for i, batch in enumerate(loader):
            if i >= n:
                break
            model.test_step(_to_device(batch, model.device), i)
            tl.append(model.logged["test/loss"])
        out = {"test/loss": float(torch.stack(tl).mean()), **model.test_metrics.compute()} if tl else {}
        return [out]
