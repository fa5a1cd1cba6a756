"""Minimal Lightning-free trainer honouring the ``lightning.pytorch.Trainer`` kwargs the reference's configs use
(configs/trainer/*.yaml): max_epochs / max_steps, precision ("bf16-mixed" is what the kernels implement),
gradient_clip_val, limit_*_batches, fast_dev_run, devices / strategy / num_nodes, default_root_dir,
enable_checkpointing.

Multi-GPU (reference: ``strategy: ddp``, configs/trainer/ddp.yaml; Lightning re-launches the script per GPU): here
the launcher is ``python -m torch.distributed.run --nproc-per-node N``; the Trainer joins the RCCL process group the
launcher describes (``comm.init_from_env``), REFUSES to run when ``devices x num_nodes`` disagrees with the world
the launcher started, and reduces gradients with this package's bucketed all-reduce (no DDPStrategy).

Loop semantics per step: training_step -> loss.backward() -> grad all-reduce (mean) ->
clip_grad_norm_ -> AdamW.step -> scheduler.step().  Checkpoints (reference: ModelCheckpoint monitor ``val/R@1``,
``save_last``; configs/callbacks/default.yaml:8-14, gated by ``save_ckpt`` -> ``enable_checkpointing``,
src/train.py:77-99) are written by rank 0 under ``default_root_dir/checkpoints``; ``fit(ckpt_path=...)`` resumes
weights, optimiser moments, LR schedule and the step counter (src/train.py:119)."""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Optional

import torch

from . import comm, graph, streams


def _to_device(batch: Dict[str, Any], device) -> Dict[str, Any]:
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _requested_world(devices: Any, num_nodes: int) -> Optional[int]:
    """Ranks the config asks for, or None when it leaves the choice to the launcher ("auto", -1)."""
    if devices in ("auto", None, -1, "-1"):
        return None
    if isinstance(devices, (list, tuple)):
        n = len(devices)
    else:
        n = int(devices)
    return n * max(1, int(num_nodes))


class CheckpointCallback:
    """The slice of ``lightning.pytorch.callbacks.ModelCheckpoint`` that src/train.py reads back."""

    def __init__(self, dirpath: str, monitor: str = "val/R@1", mode: str = "max"):
        self.dirpath, self.monitor, self.mode = dirpath, monitor, mode
        self.best_model_path = ""
        self.last_model_path = ""
        self.best_model_score: Optional[float] = None

    def is_better(self, value: float) -> bool:
        if self.best_model_score is None:
            return True
        return value > self.best_model_score if self.mode == "max" else value < self.best_model_score


class Trainer:
    def __init__(self, max_epochs: Optional[int] = 1, min_epochs: int = 1, max_steps: int = -1, accelerator: str = "auto",
                 devices: Any = "auto", precision: str = "bf16-mixed", gradient_clip_val: Optional[float] = None,
                 check_val_every_n_epoch: int = 1, log_every_n_steps: int = 50, fast_dev_run: Any = False,
                 limit_train_batches: float = 1.0, limit_val_batches: float = 1.0, limit_test_batches: float = 1.0,
                 deterministic: bool = False, strategy: str = "auto", num_nodes: int = 1, sync_batchnorm: bool = False,
                 default_root_dir: Optional[str] = None, enable_checkpointing: bool = False, callbacks=None, logger=None,
                 val_check_interval: Any = 1.0, **unused):
        if precision not in ("bf16-mixed", "bf16", "fp8-mixed", "fp8", "32", "32-true", 32):
            raise ValueError(f"precision {precision!r}: 'bf16-mixed' (bf16 MFMA GEMMs, fp32 accumulation / residual stream "
                             "/ master weights) or 'fp8-mixed' (e4m3 MFMA GEMMs in the transformer blocks on the same policy)")
        # the GEMM operand type is a property of the net (its weight copies are laid out for it at construction):
        # train.train copies trainer.precision into model.net.precision, and fit() refuses a net built for another one
        self.precision = "fp8" if str(precision).startswith("fp8") else "bf16"
        if accelerator == "cpu":
            raise RuntimeError("accelerator=cpu: this build has no CPU path (the oracle under oracle/ is test-only)")
        if strategy not in ("auto", "ddp", "ddp_find_unused_parameters_false", "ddp_find_unused_parameters_true", None):
            raise ValueError(f"strategy {strategy!r}: data parallel (one process per GPU, 'ddp') is the only strategy")
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.gradient_clip_val = gradient_clip_val
        self.fast_dev_run = fast_dev_run
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.limit_test_batches = limit_test_batches
        self.log_every_n_steps = log_every_n_steps
        # Lightning semantics: validate after every n-th epoch; val_check_interval = fraction of an epoch (float <= 1.0) or a
        # number of training batches (int) between validation runs inside an epoch (tests/test_train.py of the reference)
        self.check_val_every_n_epoch = max(1, int(check_val_every_n_epoch or 1))
        if isinstance(val_check_interval, float) and not 0.0 < val_check_interval <= 1.0:
            raise ValueError(f"val_check_interval={val_check_interval}: a float must lie in (0, 1]")
        self.val_check_interval = val_check_interval
        self.devices, self.strategy, self.num_nodes = devices, strategy, num_nodes
        self.default_root_dir = default_root_dir
        self.global_step = 0
        self.current_epoch = 0
        self.estimated_stepping_batches = None
        self.history = []
        self.callback_metrics: Dict[str, float] = {}
        # ---- distributed: join the group the launcher described, or fail loudly
        want = _requested_world(devices, num_nodes)
        env_rank, env_local, env_W = comm.env_world()
        if want is not None and want > 1 and env_W == 1 and not comm.is_dist():
            raise RuntimeError(
                f"trainer.devices={devices} x num_nodes={num_nodes} asks for {want} ranks but this process was started "
                f"alone (WORLD_SIZE unset).  Launch one process per GPU: `python -m torch.distributed.run --nnodes=1 "
                f"--nproc-per-node {want} --master-addr 127.0.0.1 -m spatial_clip_amd.train ...`")
        if env_W > 1 or comm.is_dist():
            comm.init_from_env(expect_world=want)
        self.rank, self.world_size = comm.world()
        self.is_global_zero = self.rank == 0
        # callbacks: the reference instantiates Lightning callbacks from cfg.callbacks (src/train.py:77-99,
        # configs/callbacks/default.yaml).  Here the two that change what a run produces are read from the same config
        # keys: model_checkpoint {dirpath, monitor, mode} and early_stopping {monitor, mode, patience, min_delta};
        # model_summary / rich_progress_bar are presentation only and ignored.
        cbs = callbacks if isinstance(callbacks, dict) else {}
        mc = cbs.get("model_checkpoint") if isinstance(cbs.get("model_checkpoint"), dict) else None
        es = cbs.get("early_stopping") if isinstance(cbs.get("early_stopping"), dict) else None
        self.checkpoint_callback: Optional[CheckpointCallback] = None
        if enable_checkpointing and not fast_dev_run and (default_root_dir or (mc and mc.get("dirpath"))):
            dirpath = (mc or {}).get("dirpath") or os.path.join(str(default_root_dir), "checkpoints")
            self.checkpoint_callback = CheckpointCallback(str(dirpath), monitor=(mc or {}).get("monitor") or "val/R@1",
                                                          mode=(mc or {}).get("mode") or "max")
        self.early_stopping = None
        if es and es.get("monitor") and not fast_dev_run:
            if es.get("mode", "min") not in ("min", "max"):
                raise ValueError(f"early_stopping.mode={es.get('mode')!r}: 'min' or 'max'")
            self.early_stopping = {"monitor": es["monitor"], "mode": es.get("mode", "min"),
                                   "patience": int(es.get("patience", 3)), "min_delta": float(es.get("min_delta", 0.0)),
                                   "best": None, "wait": 0}
        self.should_stop = False

    def _validate(self, model, datamodule) -> Dict[str, float]:
        """One pass over the validation loader: val/loss, the module's retrieval metrics, the zero-shot metric if a
        gene bank is configured (src/models/spatial_clip_module.py:104-136)."""
        val = datamodule.val_dataloader()
        n_val = self._limit(len(val), self.limit_val_batches)
        model.val_metrics.reset()
        model.on_validation_start()
        if model.zero_shot_metric:
            model.zero_shot_metric.reset()
        vl, out = [], {}
        for i, batch in enumerate(val):
            if i >= n_val:
                break
            model.validation_step(_to_device(batch, model.device), i)
            vl.append(model.logged["val/loss"])
        if vl:
            out["val/loss"] = self._synced(model, "val/loss", float(torch.stack(vl).mean()))
            out.update(model.val_metrics.compute())
            if model.zero_shot_metric and model.gene_bank_embeddings is not None:
                out["val/zero_shot_pcc"] = model.zero_shot_metric.compute()
        return out

    @staticmethod
    def _synced(model, name: str, value: float) -> float:
        """Mean over ranks of a value the module logged with ``sync_dist=True`` (rank-local mean otherwise)."""
        if name in getattr(model, "synced", ()):
            return comm.all_reduce_mean_scalars([value])[0]
        return value

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
        net_prec = getattr(model.net, "precision", "bf16")
        if net_prec != self.precision:
            raise RuntimeError(f"trainer.precision asks for {self.precision} GEMMs but the net was built with "
                               f"precision={net_prec!r}: set model.net.precision (spatial_clip_amd.train does it from "
                               "trainer.precision)")
        cfg = model.configure_optimizers()
        opt = cfg["optimizer"]
        sched = cfg.get("lr_scheduler", {}).get("scheduler")
        self.optimizer, self.scheduler = opt, sched
        start_epoch = 0
        rank, W = comm.world()
        # gradient exchange of the group (None without one): the bucketed all-reduce by default, the sharded reduce-scatter /
        # all-gather exchange with SC_GRAD_EXCHANGE=sharded (comm.make_grad_exchange).  Attached before a checkpoint is
        # loaded: the sharded optimiser keeps only this rank's piece of the moments.
        reducer = comm.make_grad_exchange(model.net.store)
        model.net.grad_bucket_hook = reducer.bucket_ready if reducer is not None else None
        if hasattr(opt, "attach_exchange"):
            opt.attach_exchange(reducer)
        if ckpt_path:
            if not os.path.isfile(ckpt_path):
                raise FileNotFoundError(f"ckpt_path {ckpt_path!r} does not exist")
            self.global_step = self.load_checkpoint(ckpt_path, model, opt, sched)
            start_epoch = self.global_step // max(n_train, 1)
        # single process, bf16, FusedAdamW: the step can run as ONE hipGraph (SC_GRAPH=auto: only where the first eager steps
        # show the host as the bottleneck -- the reference's ViT-B-32 / batch-32 experiments; =1 always; =0 never)
        gstep = None
        if reducer is None and graph.graph_mode() != "0" and hasattr(opt, "step_captured") and torch.cuda.is_available():
            gstep = graph.GraphedTrainStep(model, opt, max_norm=self.gradient_clip_val, grad_scale=1.0 / W,
                                           policy="always" if graph.graph_mode() == "1" else "host_bound")
        self.graphed_step = gstep
        vci = self.val_check_interval
        val_every = 0 if self.fast_dev_run else (int(vci) if isinstance(vci, int) and not isinstance(vci, bool)
                                                 else (max(1, int(n_train * vci)) if vci < 1.0 else 0))
        self.val_runs = 0
        for epoch in range(start_epoch, epochs):
            self.current_epoch = epoch
            if hasattr(datamodule, "set_epoch"):         # fresh shuffle + augmentation draws, also after a resume
                datamodule.set_epoch(epoch)
            model.train_metrics.reset()
            t0 = time.time()
            skip = self.global_step - epoch * n_train if epoch == start_epoch else 0     # mid-epoch resume
            for i, batch in enumerate(train):
                if i >= n_train or (self.max_steps != -1 and self.global_step >= self.max_steps):
                    break
                if i < skip:
                    continue
                batch = _to_device(batch, model.device)
                if gstep is not None:           # one hipGraph per step where the host is the bottleneck (graph.py); eager otherwise
                    loss = gstep(batch)
                else:
                    with streams.chain_stream():
                        loss = model.training_step(batch, i)
                        loss.backward(model.root_gradient(loss))
                        if reducer is not None:
                            reducer.finish()
                        opt.step(grad_scale=1.0 / W, max_norm=self.gradient_clip_val)
                if sched is not None:
                    sched.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or self.fast_dev_run:
                    self.history.append({"step": self.global_step,
                                         "train/loss": self._synced(model, "train/loss", float(loss.detach()))})
                if val_every and (i + 1) % val_every == 0 and (i + 1) < n_train \
                        and (epoch + 1) % self.check_val_every_n_epoch == 0:
                    self.history.append({"step": self.global_step, "epoch": epoch, **self._validate(model, datamodule)})
                    self.val_runs += 1
            torch.cuda.synchronize()
            rec = {"epoch": epoch, "time_s": time.time() - t0, **model.train_metrics.compute()}
            last_train = next((h["train/loss"] for h in reversed(self.history) if "train/loss" in h), None)
            if last_train is not None:
                rec["train/loss"] = last_train
            if (epoch + 1) % self.check_val_every_n_epoch == 0:
                rec.update(self._validate(model, datamodule))
                self.val_runs += 1
            self.history.append(rec)
            self._checkpoint_epoch(model, opt, sched, rec)
            if self._early_stop(rec):
                break
        self.callback_metrics = {k: v for k, v in self.history[-1].items() if isinstance(v, float)} if self.history else {}
        model.net.store.wait_all()      # the last optimiser update may run behind the (absent) next forward: fit returns finished weights

    # ------------------------------------------------------------------ checkpoints (reference state_dict names)
    def _early_stop(self, rec: Dict[str, Any]) -> bool:
        """lightning.pytorch.callbacks.EarlyStopping on the epoch's validation record: stop after `patience` validated
        epochs without an improvement of more than min_delta.  The monitored value is the same on every rank when it is
        an all-reduced metric (R@k) or a loss logged with sync_dist=True; rank 0's decision is broadcast regardless, so
        that a rank-local monitor can never leave part of the group blocked in the next collective."""
        es = self.early_stopping
        if es is None or es["monitor"] not in rec:
            return False
        v = float(rec[es["monitor"]])
        better = es["best"] is None or (v > es["best"] + es["min_delta"] if es["mode"] == "max"
                                        else v < es["best"] - es["min_delta"])
        if better:
            es["best"], es["wait"] = v, 0
        else:
            es["wait"] += 1
        self.should_stop = comm.broadcast_flag(es["wait"] >= es["patience"], src=0)
        return self.should_stop

    def _checkpoint_epoch(self, model, opt, sched, rec: Dict[str, Any]) -> None:
        """Best / last bookkeeping runs on EVERY rank (Lightning broadcasts ``best_model_path``: src/train.py:122-128
        reads it back on all ranks for ``trainer.test(ckpt_path=...)``); only the file writes / removals are rank 0's.
        The paths are deterministic and BOTH the decision and the score are rank 0's, broadcast together (one float64
        collective), so every rank names the same file and compares later epochs against the same best score -- also with a
        rank-local monitor (advisor, round 3: only the flag used to travel)."""
        cb = self.checkpoint_callback
        if cb is None:
            return
        last = os.path.join(cb.dirpath, "last.ckpt")
        # the optimiser state of a sharded optimiser is gathered by a collective: every rank forms it, rank 0 writes it
        # (with the replicated optimiser only rank 0 forms it: the other ranks would clone 8 bytes per parameter to discard them)
        sharded = getattr(opt, "exchange", None) is not None and hasattr(opt.exchange, "gather_moments")
        opt_state = opt.state_dict() if opt is not None and (self.is_global_zero or sharded) else None
        if self.is_global_zero:
            os.makedirs(cb.dirpath, exist_ok=True)
            self.save_checkpoint(last, model, opt, sched, self.global_step, optimizer_state=opt_state)
        cb.last_model_path = last
        score = rec.get(cb.monitor)
        improved = isinstance(score, float) and cb.is_better(score)
        improved, score0 = comm.broadcast_flag_and_value(improved, score if isinstance(score, float) else float("nan"), src=0)
        if improved:
            score = score0
            best = os.path.join(cb.dirpath, f"epoch_{int(rec['epoch']):03d}.ckpt")
            if self.is_global_zero:
                self.save_checkpoint(best, model, opt, sched, self.global_step, optimizer_state=opt_state)
                if cb.best_model_path and cb.best_model_path != best and os.path.exists(cb.best_model_path):
                    os.remove(cb.best_model_path)
            cb.best_model_path = best
            cb.best_model_score = float(score) if isinstance(score, float) else cb.best_model_score
        if comm.is_dist():
            torch.distributed.barrier()

    @staticmethod
    def save_checkpoint(path: str, model, optimizer=None, scheduler=None, global_step: int = 0, optimizer_state=None) -> None:
        """``state_dict`` uses the reference ``CLIP.state_dict()`` key names, so the file loads into open_clip too.
        ``optimizer_state``: what ``optimizer.state_dict()`` returned on THIS rank (with the sharded optimiser that call is a
        collective, so a caller that writes on one rank only passes the result in)."""
        ck = {"state_dict": {k: v.cpu() for k, v in model.net.state_dict().items()}, "global_step": global_step}
        if optimizer is not None:
            osd = optimizer_state if optimizer_state is not None else optimizer.state_dict()
            ck["optimizer"] = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in osd.items()}
        if scheduler is not None:
            ck["scheduler_last_epoch"] = scheduler.last_epoch
        fp8 = model.net.fp8_scaling_state() if hasattr(model.net, "fp8_scaling_state") else None
        if fp8 is not None:
            ck["fp8_scaling"] = fp8         # delayed e4m3 scales + amax history: a resumed run continues with them
        torch.save(ck, path)

    @staticmethod
    def load_checkpoint(path: str, model, optimizer=None, scheduler=None) -> int:
        """Accepts this trainer's files and the reference's Lightning / DDP / plain-CLIP layouts (key prefixes
        ``net.model.``, ``module.`` are stripped: net.strip_checkpoint_prefix)."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        model.net.load_checkpoint_state_dict(ck, source=path)          # resets the fp8 scaling history ...
        if hasattr(model.net, "load_fp8_scaling_state"):
            model.net.load_fp8_scaling_state(ck.get("fp8_scaling"))     # ... and restores it when the file carries one
        if optimizer is not None and isinstance(ck.get("optimizer"), dict) and "exp_avg" in ck["optimizer"]:
            optimizer.load_state_dict({k: (v.to(model.device) if isinstance(v, torch.Tensor) else v)
                                       for k, v in ck["optimizer"].items()})
        if scheduler is not None and "scheduler_last_epoch" in ck:
            scheduler.last_epoch = int(ck["scheduler_last_epoch"])
            scheduler._apply()
        return int(ck.get("global_step", 0))

    def test(self, model, datamodule, ckpt_path: Optional[str] = None):
        if ckpt_path:
            if not os.path.isfile(ckpt_path):
                raise FileNotFoundError(f"ckpt_path {ckpt_path!r} does not exist")
            self.load_checkpoint(ckpt_path, model)
        if getattr(datamodule, "preprocess_fn", None) is None:
            datamodule.preprocess_fn = model.net.preprocess_val
            datamodule.tokenizer = model.net.tokenizer
            datamodule.setup("test")
        loader = datamodule.test_dataloader()
        n = self._limit(len(loader), self.limit_test_batches)
        model.test_metrics.reset()
        tl = []
        for i, batch in enumerate(loader):
            if i >= n:
                break
            model.test_step(_to_device(batch, model.device), i)
            tl.append(model.logged["test/loss"])
        out = {"test/loss": self._synced(model, "test/loss", float(torch.stack(tl).mean())),
               **model.test_metrics.compute()} if tl else {}
        self.callback_metrics = {**self.callback_metrics, **{k: v for k, v in out.items() if isinstance(v, float)}}
        return [out]
