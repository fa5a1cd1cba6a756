"""Two data-parallel ranks on ONE GPU (gloo backend with device tensors) against a single process on the concatenated
batch: exercises the product's whole multi-GPU code path (packed feature all-gather, reduce-scatter of the gather
gradient, bucketed gradient all-reduce launched from backward, 1/world_size folded into AdamW) on real kernels.
RCCL itself cannot be exercised on a one-GPU box; the collectives differ only in the backend string."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(rank, world, port, steps, B, ret, exchange="sharded"):
    import sys
    sys.path.insert(0, ROOT)
    import functools
    import torch.distributed as dist
    os.environ["SC_GRAD_EXCHANGE"] = exchange
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, model_configs as mc, module, net, optim
    cfg = mc.ModelCfg(embed_dim=64, vision=mc.VisionCfg(32, 8, 64, 2, 32), text=None, gene=mc.GeneCfg(200, 64))
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=7)
    loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.0,
                                 neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 10, None, 10
    m.trainer = T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.make_grad_exchange(n.store, bucket_floats=20000)          # None in the single process
    assert (reducer is None) == (world == 1)
    if world > 1:
        assert isinstance(reducer, comm.ShardedGradExchange if exchange == "sharded" else comm.GradBucketReducer)
    n.grad_bucket_hook = reducer.bucket_ready if reducer is not None else None
    opt.attach_exchange(reducer)
    if world > 1 and exchange == "sharded":
        assert opt.exp_avg.numel() == n.store.total // world and len(reducer.buckets) >= 3       # Adam moments: this rank's 1/W
    losses_out = []
    for s in range(steps):
        if world > 1:
            b = data.synthetic_batch(B, 32, 200, 4, s, rank, world)
        else:       # the concatenation of what the two ranks see
            parts = [data.synthetic_batch(B // 2, 32, 200, 4, s, r, 2) for r in range(2)]
            b = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        loss = m.training_step({k: v.cuda() for k, v in b.items()}, s)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(grad_scale=1.0 / world, max_norm=1.0)
        sched.step()
        losses_out.append(float(loss.detach()))
    torch.cuda.synchronize()
    osd = opt.state_dict()                       # collective with the sharded optimiser: full flat-layout moments
    key = (world, rank) if exchange == "sharded" else (world, rank, exchange)
    ret[key] = {"loss": losses_out, "master": n.store.master.detach().cpu(), "exp_avg": osd["exp_avg"].cpu(),
                "exp_avg_sq": osd["exp_avg_sq"].cpu(), "proj": n.store.p("visual.proj").cpu(),
                          "fc2": n.store.p("gene.fc2.weight").cpu(), "qkv": n.store.p(
                              "visual.transformer.resblocks.0.attn.in_proj_weight").cpu()}
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_match_single_process():
    """Two ranks (default gradient exchange: per-bucket reduce-scatter -> AdamW on the rank's pieces -> all-gather behind the
    next forward) against one process on the concatenated batch, and against the SAME two ranks on the rounds-1-4 route
    (bucketed SUM all-reduce + replicated AdamW): weights and Adam moments bit-identical after 3 steps."""
    mp.set_start_method("spawn", force=True)
    steps, B = 3, 8
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_run, args=(1, 0, steps, 2 * B, ret), nprocs=1, join=True)
        mp.spawn(_run, args=(2, 29713, steps, B, ret), nprocs=2, join=True)
        mp.spawn(_run, args=(2, 29715, steps, B, ret, "allreduce"), nprocs=2, join=True)
        res = dict(ret)
    one, r0, r1 = res[(1, 0)], res[(2, 0)], res[(2, 1)]
    a0, a1 = res[(2, 0, "allreduce")], res[(2, 1, "allreduce")]
    for k in ("master", "exp_avg", "exp_avg_sq"):
        assert torch.equal(r0[k], r1[k]), f"sharded route: ranks diverged on {k}"
        assert torch.equal(a0[k], a1[k]), f"all-reduce route: ranks diverged on {k}"
        assert torch.equal(r0[k], a0[k]), f"the two gradient-exchange routes differ on {k}"
    assert r0["loss"] == a0["loss"] and r1["loss"] == a1["loss"]
    for k in ("proj", "fc2", "qkv"):
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"          # same reduced gradients -> same weights
        assert float((r0[k] - one[k]).abs().max()) < 3e-3, k                  # and they follow the single-process run
    for s in range(steps):
        mean2 = 0.5 * (r0["loss"][s] + r1["loss"][s])                         # mean of the per-rank local losses
        assert abs(mean2 - one["loss"][s]) < (4e-3 if s < 2 else 2e-2), (s, mean2, one["loss"][s])


# ----------------------------------------------------------------------------------------------------------------------
# every (local_loss, gather_with_grad) layout of ClipLoss against the REFERENCE's own 2-rank gloo run
def _layout_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import losses
    z = np.load(os.path.join(ROOT, "tests", "golden", "loss_w2.npz"))
    img, txt, scale = torch.from_numpy(z["img"]), torch.from_numpy(z["txt"]), float(z["scale"])
    B = img.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    out = {}
    for name, (local_loss, gwg) in {"bg_grad": (True, True), "gg_grad": (False, True), "gg_nograd": (False, False),
                                    "bg_nograd": (True, False)}.items():
        i = img[sl].cuda().requires_grad_(True)
        t = txt[sl].cuda().requires_grad_(True)
        s = torch.tensor(scale, device="cuda", requires_grad=True)
        crit = losses.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, cache_labels=True, rank=rank, world_size=world)
        l = crit(i, t, s)["contrastive_loss"]
        l.backward()
        out[name] = {"loss": float(l), "gimg": i.grad.cpu().numpy(), "gtxt": t.grad.cpu().numpy(), "gscale": float(s.grad)}
    ret[rank] = out
    dist.destroy_process_group()


def test_clip_loss_layouts_two_ranks_match_reference():
    import numpy as np
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_layout_worker, args=(2, 29721, ret), nprocs=2, join=True)
        res = dict(ret)
    zl = np.load(os.path.join(ROOT, "tests", "golden", "loss_w2_layouts.npz"))
    z0 = np.load(os.path.join(ROOT, "tests", "golden", "loss_w2.npz"))
    for r in range(2):
        for name in ("bg_grad", "gg_grad", "gg_nograd", "bg_nograd"):
            if name == "bg_grad":
                want = {k: z0[f"r{r}_clip_{k}"] for k in ("loss", "gimg", "gtxt", "gscale")}
            else:
                want = {k: zl[f"r{r}_{name}_{k}"] for k in ("loss", "gimg", "gtxt", "gscale")}
            got = res[r][name]
            assert abs(got["loss"] - float(want["loss"])) < 5e-6, (r, name)
            assert abs(got["gscale"] - float(want["gscale"])) < 5e-6, (r, name)
            np.testing.assert_allclose(got["gimg"], want["gimg"], atol=5e-6, err_msg=f"{r} {name} gimg")
            np.testing.assert_allclose(got["gtxt"], want["gtxt"], atol=5e-6, err_msg=f"{r} {name} gtxt")


# ----------------------------------------------------------------------------------------------------------------------
# the PRODUCT entry point (train.train: hydra config -> comm.init_from_env -> Trainer.fit) under two ranks, with the
# feature all-gather on the communication stream vs the synchronous gather: bit-identical weights
def _entry_worker(rank, world, port, overlap, ret):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world), "SC_DIST_BACKEND": "gloo", "SC_GATHER_OVERLAP": "1" if overlap else "0",
                       "PROJECT_ROOT": ROOT})
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, hydra_lite, train
    cfg = hydra_lite.compose("train.yaml", ["experiment=smoke_shards", "trainer=ddp", f"trainer.devices={world}",
                                            "trainer.fast_dev_run=false", "trainer.max_epochs=1", "loss=spatial",
                                            "data.steps_per_epoch=3", "data.batch_size=8", "test=false",
                                            "model.net.model_name=ViT-Ti-16-gene", "data.n_genes=512"])
    metrics, obj = train.train(cfg)
    m = obj["model"]
    assert comm.world() == (rank, world)
    assert obj["trainer"].world_size == world
    fg = m._feature_gather
    m.net.store.wait_all()        # an optimiser update may still be running behind the forward (communication stream)
    ret[(overlap, rank)] = {"w": m.net.store.master.detach().cpu(), "loss": metrics.get("val/loss"),
                            "gathers": 0 if fg is None else fg.launched, "device": torch.cuda.current_device()}
    comm.shutdown()


def test_train_entry_point_two_ranks_overlapped_gather_is_bit_identical():
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_entry_worker, args=(2, 29731, True, ret), nprocs=2, join=True)
        mp.spawn(_entry_worker, args=(2, 29741, False, ret), nprocs=2, join=True)
        res = dict(ret)
    assert res[(True, 0)]["gathers"] >= 6 and res[(False, 0)]["gathers"] == 0     # 3 train steps x (text, image) [+ val]
    for r in range(2):
        assert torch.equal(res[(True, r)]["w"], res[(False, r)]["w"]), f"rank {r}: overlapped gather changed the result"
    assert torch.equal(res[(True, 0)]["w"], res[(True, 1)]["w"]), "ranks diverged"
    assert torch.isfinite(res[(True, 0)]["w"]).all()


# ----------------------------------------------------------------------------------------------------------------------
# RCCL itself: a ONE-rank nccl group with the distributed code path forced on (SC_FORCE_DIST=1) runs every collective
# of the step -- packed all-gathers on the communication stream, reduce_scatter_tensor, bucketed async all-reduce -- on
# the real RCCL backend; at world size 1 they are identities, so the result must equal the plain single-process run
def _rccl_worker(rank, world, port, force, ret, native=False, exchange="sharded"):
    import sys
    sys.path.insert(0, ROOT)
    import functools
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "LOCAL_RANK": "0",
                       "WORLD_SIZE": "1", "SC_FORCE_DIST": "1" if force else "0", "SC_COMM_NATIVE": "1" if native else "0",
                       "SC_GRAD_EXCHANGE": exchange})
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, model_configs as mc, module, net, optim
    comm.init_from_env()
    if force:
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_backend() == "nccl" and comm.is_dist()
    cfg = mc.ModelCfg(embed_dim=64, vision=mc.VisionCfg(32, 8, 64, 2, 32), text=None, gene=mc.GeneCfg(200, 64))
    n = net.SpatialClipNet("custom", None, model_cfg=cfg, seed=3)
    loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                                 neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=1))

    class T:
        max_steps, max_epochs, estimated_stepping_batches = 10, None, 10
    m.trainer = T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.make_grad_exchange(n.store, bucket_floats=20000)
    assert (reducer is not None) == bool(force)
    n.grad_bucket_hook = reducer.bucket_ready if reducer is not None else None
    opt.attach_exchange(reducer)
    ls = []
    for s in range(3):
        b = data.synthetic_batch(16, 32, 200, 4, s)
        loss = m.training_step({k: v.cuda() for k, v in b.items()}, s)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(grad_scale=1.0, max_norm=1.0)
        sched.step()
        ls.append(float(loss.detach()))
    torch.cuda.synchronize()
    nat = comm.native()
    assert (nat is not None) == bool(native and force)
    key = ("native" if native else force) if exchange == "sharded" else exchange
    n.store.wait_all()
    ret[key] = {"w": n.store.master.detach().cpu(), "loss": ls,
                "gathers": 0 if m._feature_gather is None else m._feature_gather.launched,
                "native_calls": 0 if nat is None else nat.launched, "stats": {k: list(v) for k, v in comm.STATS.items()},
                "info": comm.describe() if force else None}
    comm.shutdown()
    assert comm.native() is None


def nat_info(res):
    return res["native"]["info"]["inplace_collectives_verified"]


def test_rccl_one_rank_group_runs_every_collective_and_changes_nothing():
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_worker, args=(1, 29751, True, ret), nprocs=1, join=True)
        mp.spawn(_rccl_worker, args=(1, 29752, False, ret), nprocs=1, join=True)
        mp.spawn(_rccl_worker, args=(1, 29753, True, ret, True), nprocs=1, join=True)
        mp.spawn(_rccl_worker, args=(1, 29754, True, ret, False, "allreduce"), nprocs=1, join=True)
        res = dict(ret)
    assert res[True]["gathers"] == 6 and res[False]["gathers"] == 0
    assert res[True]["loss"] == res[False]["loss"]
    assert torch.equal(res[True]["w"], res[False]["w"])
    # the default gradient exchange ran on RCCL: reduce-scatters and weight all-gathers, no bucket all-reduce; the rounds-1-4
    # route (SC_GRAD_EXCHANGE=allreduce) the other way round; same weights
    st = res[True]["stats"]
    assert st["reduce_scatter(gradient bucket)"][0] >= 9 and st["all_gather(weights bucket)"][0] == st["reduce_scatter(gradient bucket)"][0]
    assert "all_reduce(gradient bucket)" not in st
    assert res[True]["info"]["backend"] == "nccl" and res[True]["info"]["rccl_ranks"] == 1 and res[True]["info"]["grad_exchange"] == "sharded"
    assert res[True]["info"]["inplace_collectives_verified"] is True and nat_info(res) is True
    ar = res["allreduce"]
    assert ar["stats"]["all_reduce(gradient bucket)"][0] >= 3 and "all_gather(weights bucket)" not in ar["stats"]
    assert ar["loss"] == res[False]["loss"] and torch.equal(ar["w"], res[False]["w"])
    # the same step with the collectives issued by the kernel library's own RCCL entry points (SC_COMM_NATIVE=1:
    # sc_comm_init + sc_allgather_feats_async / sc_reduce_scatter_grads_async / sc_allreduce_sum_async on explicit streams)
    nat = res["native"]
    assert nat["gathers"] == 6 and nat["native_calls"] >= 6 + 3 + 2 * 9      # gathers + reduce-scatters + per step >= 3 gradient buckets, each a reduce-scatter and an all-gather
    assert nat["loss"] == res[False]["loss"] and torch.equal(nat["w"], res[False]["w"])


# ----------------------------------------------------------------------------------------------------------------------
# bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N):
# two ranks share the one GPU of the test box (LOCAL_RANK % device_count) over gloo; rank 0 prints ONE JSON line that
# carries the true world size and the global batch, and a --gpus / WORLD_SIZE mismatch exits non-zero
def test_bench_under_torchrun_two_ranks_prints_one_line():
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SC_DIST_BACKEND="gloo", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    args = ["--steps", "2", "--warmup", "1", "--model", "ViT-Ti-16-gene", "--batch", "16", "--n-genes", "512",
            "--no-cpu-baseline", "--no-loss-delta", "--no-kernel-events"]
    r = subprocess.run(base + ["--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_gpus_live"] == 2 and out["config"]["global_batch"] == 32
    assert out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak" and out["value"] > 0
    # the line explains its own communication (round-5 verdict item 6): exposed waits per step, every collective's
    # issue-to-completion time and payload rate, host enqueue time -- from the probe pass behind the timed region
    c = out["comm"]
    assert c["exposed_comm_ms_per_step"] >= 0.0 and c["host_enqueue_ms_per_step"] > 0.0
    assert any(k.startswith("gradient all-reduce") for k in c["exposed_by_wait_ms_per_step"]), c
    assert any(k.startswith("feature all-gather") for k in c["exposed_by_wait_ms_per_step"]), c
    ag = c["collectives"]["all_gather(features|ids)"]
    assert ag["launches_per_step"] == 2 and ag["avg_issue_to_done_ms"] > 0 and ag["payload_gbytes_per_s_lower_bound"] > 0
    assert c["collectives"]["all_reduce(gradient bucket)"]["launches_per_step"] >= 1
    assert c["grad_exchange"] == "allreduce"            # the default route is the reference's DDP shape
    r = subprocess.run(base + ["--gpus", "4"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (WORLD_SIZE unset): bench.spawn_ranks counts the devices in a
    throw-away child, starts torch.distributed.run itself before touching the GPU and returns the ranks' exit code.
    The two ranks share this box's one GPU (explicit opt-in SC_BENCH_SHARE_GPU=1, gloo).  Without the opt-in the same
    command refuses (exit 2) instead of oversubscribing the device."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SC_DIST_BACKEND="gloo", SC_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model",
           "ViT-Ti-16-gene", "--batch", "16", "--n-genes", "512", "--no-cpu-baseline", "--no-loss-delta", "--no-kernel-events"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_gpus_live"] == 2 and out["config"]["global_batch"] == 32 and out["value"] > 0
    if torch.cuda.device_count() < 2:
        env.pop("SC_BENCH_SHARE_GPU")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 2 and "GPU(s) are visible" in r.stderr
