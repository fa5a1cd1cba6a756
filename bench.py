#!/usr/bin/env python3
"""Headline benchmark: tile-gene pairs/sec of the full Spatial-CLIP train step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = forward (ViT image tower + gene tower) + global-batch contrastive loss + backward + gradient
all-reduce (mean) + grad-norm clip 1.0 + AdamW + cosine-warmup scheduler step -- the reference's training_step
plus what Lightning does after it.  Workload = BASELINE.json configs[1]/[2]: ViT-B/16 + gene-MLP
(20000 -> 512 -> 512), local batch 256, ClipLoss over the RCCL-gathered global batch 256*N, synthetic 224x224
tiles already resident in HBM.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, Chip-level parameters)


def flops_per_pair(cfg, G: int) -> float:
    """Algorithmic train FLOPs per pair (2 FLOP/MAC, train = 3x forward, no padding/recompute)."""
    v = cfg.vision
    L, d, mlp, D = v.tokens, v.width, int(v.width * v.mlp_ratio), cfg.embed_dim
    kp = 3 * v.patch_size ** 2
    fwd = 2.0 * (L - 1) * kp * d
    per_layer = 2.0 * L * d * 3 * d + 2 * (2.0 * L * L * d) + 2.0 * L * d * d + 2 * (2.0 * L * d * mlp)
    fwd += v.layers * per_layer + 2.0 * d * D
    if cfg.gene is not None and cfg.gene.kind == "transformer":
        g = cfg.gene
        gl, gd, gm = g.tokens, g.width, int(g.width * g.mlp_ratio)
        fwd += 2.0 * (gl - 1) * g.patch * gd + 2.0 * gd * D
        fwd += g.layers * (2.0 * gl * gd * 3 * gd + 2 * (2.0 * gl * gl * gd) + 2.0 * gl * gd * gd + 2 * (2.0 * gl * gd * gm))
    elif cfg.gene is not None:
        fwd += 2.0 * cfg.gene.n_genes * cfg.gene.hidden + 2.0 * cfg.gene.hidden * D
    elif cfg.text is not None:      # the reference's CLIP text tower (+5.960 GFLOP forward at d = 512)
        t = cfg.text
        tl, td, tm = t.context_length, t.width, int(t.width * t.mlp_ratio)
        fwd += t.layers * (2.0 * tl * td * 3 * td + 2 * (2.0 * tl * tl * td) + 2.0 * tl * td * td + 2 * (2.0 * tl * td * tm)) + 2.0 * td * D
    fwd += 4.0 * G * D
    return 3.0 * fwd


def skipped_flops_per_pair(cfg) -> float:
    """FLOPs of the reference formula that this build does NOT execute: in the last ViT block only the CLS token is
    consumed downstream (pool 'tok'), so attention output, out_proj and the MLP run for 1 of the L rows."""
    if os.environ.get("SC_CLS_ONLY", "1") == "0":
        return 0.0
    v = cfg.vision
    L, d, mlp = v.tokens, v.width, int(v.width * v.mlp_ratio)
    fwd = (L - 1) * (2.0 * d * d + 2 * (2.0 * d * mlp)) + 2 * (2.0 * (L - 1) * L * d)
    if os.environ.get("SC_CLS_Q", "1") != "0":
        fwd += (L - 1) * 2.0 * d * d        # ... and, of its qkv projection, q for that one row (bf16 path)
    return 3.0 * fwd


def host_cpu_share() -> int:
    """Cores this process may really use: min(affinity, cgroup quota, SC_CPU_THREADS or 16 -- the GPU box's share)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SC_CPU_THREADS", "16"))))


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def _oracle_cfg(cfg):
    from oracle import spatial_clip_oracle as O
    v = cfg.vision
    g, t = cfg.gene, cfg.text
    vis = O.VisionCfg(v.image_size, v.patch_size, v.width, v.layers, v.head_width)
    if g is None:               # the reference's own pairing: CLIP text tower on token ids
        return O.ModelCfg(cfg.embed_dim, vis, O.TextCfg(t.context_length, t.vocab_size, t.width, t.heads, t.layers, t.mlp_ratio),
                          None, quick_gelu=bool(getattr(cfg, "quick_gelu", False)))
    return O.ModelCfg(cfg.embed_dim, vis, None,
                      O.GeneCfg(g.n_genes, g.hidden, g.kind, g.patch, g.width, g.layers, g.head_width, g.mlp_ratio))


def make_batch(cfg, B, n_genes, step=0, rank=0, world=1, rates=None):
    """One synthetic batch of the reference's batch contract for this model: gene matrix in the ``texts`` slot for the gene
    towers, BPE-shaped token ids for the reference's text tower (data.synthetic_captions)."""
    from spatial_clip_amd import data
    b = data.synthetic_batch(B, cfg.vision.image_size, n_genes if cfg.gene is not None else 64, 8, step, rank, world,
                             rates if cfg.gene is not None else None)
    if cfg.gene is None:
        b["texts"] = data.synthetic_captions(B, cfg.text.context_length, cfg.text.vocab_size, seed=4321 + 1000 * step + rank)
    return b


def cpu_baseline_leg(model_name: str, n_genes: int, B: int, steps: int = 3, loss: str = "clip"):
    """The oracle's fp32 train step (fwd + loss + bwd + clip + AdamW + scheduler) timed on the host cores."""
    import torch
    from oracle import spatial_clip_oracle as O
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import data, model_configs as mc
    cfg = mc.get_model_config(model_name, n_genes)
    ocfg = _oracle_cfg(cfg)
    cores = host_cpu_share()
    torch.set_num_threads(cores)
    O.USE_ATEN_KERNELS = True       # timed form: same maths through the ATen kernels the reference runs (oracle header)
    tr = O.OracleTrainer(ocfg, O.init_params(ocfg, seed=0), loss=loss, lr=1e-3, warmup=2000)
    rates = data.make_gene_rates(n_genes)
    times = []
    for s in range(steps + 1):
        batch = make_batch(cfg, B, n_genes, s, rates=rates)
        t0 = time.time()
        tr.training_step(batch)
        times.append(time.time() - t0)
    timed = sorted(times[1:])
    O.USE_ATEN_KERNELS = False
    return {"workload": f"{model_name}, n_genes={n_genes}, B={B}", "pairs_per_s_best": round(B / timed[0], 3),
            "pairs_per_s_median": round(B / timed[len(timed) // 2], 3), "step_s": [round(t, 3) for t in times]}


def cpu_baseline(model_name: str, n_genes: int, B: int = 32):
    """Reported baseline only (SURVEY.md 8d / BASELINE.md section 4): the oracle restatement of the identical
    training_step on the GPU box's host cores, cfg [0] (ViT-Ti/16 + 2-layer gene-MLP, B = 8) and the bench model at a
    reduced batch (B = 32), 1 warm-up + 3 timed steps each."""
    legs = [cpu_baseline_leg("ViT-Ti-16-gene", n_genes, 8), cpu_baseline_leg(model_name, n_genes, min(32, B))]
    head = legs[1]
    return {"value": head["pairs_per_s_best"], "unit": "pairs/s", "cores": host_cpu_share(), "kind": "port",
            "cpu_model": cpu_model_string(),
            "sample": f"oracle fp32 train step ({head['workload']}), 1 warm-up + 3 timed steps, best; "
                      f"configs[0] leg ({legs[0]['workload']}) alongside",
            "legs": legs}


# Stated tolerances against the fp32 oracle: spatial_clip_amd/parity.py -- the north-star's 1e-3 on the
# loss everywhere; features 5e-3 (bf16) / 8e-3 (e4m3 operands) at the initial weights; at trained weights the feature bound is
# TRAINED_POINT_NOISE_FACTOR x the reference policy's own noise there (bf16 autocast over the fp32 oracle, same weights and
# batch), never below the initial-weights bound.  ONE pass / fail per point covers the loss AND the features.
PARITY_TRAINED_STEPS = 60           # the trained-weights point: exactly this many optimiser steps from the initial weights,
                                    # whatever --steps / --warmup / the schedule-selection setup ran before


def _tolerances():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import parity
    return parity


def loss_delta_vs_oracle(m, n, cfg, batch_cpu, loss_kind: str, dtype: str = "bf16", point: str = "", trained: bool = False):
    """Half of BASELINE's metric: |loss(HIP) - loss(fp32 oracle)| on the SAME batch and the SAME weights at the
    benchmark's own size, outside the timed region (oracle forward only: ~10 s of host time at B = 256).  ``trained``: also
    run the oracle under torch.autocast(bf16) -- the reference's own precision policy -- and bound the HIP features by its
    distance from the fp32 oracle at these weights (parity.trained_point_feature_bound)."""
    import torch
    from oracle import spatial_clip_oracle as O
    par = _tolerances()
    torch.set_num_threads(host_cpu_share())
    policy_noise = None
    with torch.no_grad():
        out = m.model_step({k: v.cuda() for k, v in batch_cpu.items()})
        torch.cuda.synchronize()
        hip_loss = float(out["loss"])
        f_i, f_t = out["image_features"].float().cpu(), out["text_features"].float().cpu()
        p = {k: v.detach().cpu() for k, v in n.state_dict().items()}
        f = O.net_forward(batch_cpu["images"], batch_cpu["texts"], p, _oracle_cfg(cfg))
        if loss_kind == "clip":
            ref = O.clip_loss(f["image_features"], f["text_features"], f["logit_scale"])
        else:
            ref = O.spatial_loss(f["image_features"], f["text_features"], f["logit_scale"], batch_cpu["image_tile_ids"],
                                 batch_cpu["text_tile_ids"], batch_cpu["neighbor_tile_ids"], batch_cpu["neighbor_alphas"])
        if trained:
            O.USE_ATEN_KERNELS = True
            O.REFERENCE_AUTOCAST_STREAM = (n.residual_stream == "bf16" and os.environ.get("SC_RES_STREAM", "bf16") == "bf16")
            try:        # the reference's policy with the residual stream it really carries (oracle.REFERENCE_AUTOCAST_STREAM)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    fa = O.net_forward(batch_cpu["images"], batch_cpu["texts"], p, _oracle_cfg(cfg))
            finally:
                O.USE_ATEN_KERNELS = False
                O.REFERENCE_AUTOCAST_STREAM = False
            policy_noise = max(float((fa["image_features"].float() - f["image_features"]).abs().max()),
                               float((fa["text_features"].float() - f["text_features"]).abs().max()))
    d_img, d_txt = float((f_i - f["image_features"]).abs().max()), float((f_t - f["text_features"]).abs().max())
    dfeat = max(d_img, d_txt)
    ftol = par.trained_point_feature_bound(policy_noise, dtype) if trained else par.FEATURE_TOLERANCE[dtype]
    dl = abs(hip_loss - float(ref))
    res = {"loss_hip": hip_loss, "loss_oracle_fp32": float(ref), "loss_delta_vs_oracle": dl, "max_abs_feature_delta": dfeat,
           "max_abs_image_feature_delta": d_img, "max_abs_second_tower_feature_delta": d_txt,
           "batch": int(batch_cpu["images"].shape[0]), "point": point, "tolerance": par.LOSS_TOLERANCE[dtype],
           "feature_tolerance": ftol,
           "loss_within_tolerance": bool(dl <= par.LOSS_TOLERANCE[dtype]),
           "features_within_tolerance": bool(dfeat <= ftol),
           "within_tolerance": bool(dl <= par.LOSS_TOLERANCE[dtype] and dfeat <= ftol)}
    if trained:
        res["reference_policy_feature_noise"] = policy_noise
        res["feature_tolerance_rule"] = (f"max({par.FEATURE_TOLERANCE[dtype]:g}, {par.TRAINED_POINT_NOISE_FACTOR:g} x max-abs feature "
                                         "distance of torch.autocast(bf16) over the fp32 oracle from the fp32 oracle, same weights and batch)")
    return res


PMC_TRAFFIC_FILES = ("r06_pmc_traffic_summary.json", "r05_pmc_traffic_summary.json", "r04_pmc_traffic_summary.json")


def pmc_traffic_nt():
    """(HBM bytes per NT-GEMM launch, file) from the newest committed rocprofv3 PMC passes of this command
    (profiles/README.md): counters cannot be read from inside the process, so this number is NOT measured by the run that
    prints it -- the line says so (``traffic_measured_by_this_run``, ``traffic_from_profile``).  (None, None) if absent."""
    for name in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        tot = n = 0
        for k, v in d.items():
            if k.startswith("gemm8p_kernel") or k.startswith("gemm8pp_kernel"):
                tot += (v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"]
                n += v["launches"]
        if n:
            return round(tot / n), "profiles/" + name
    return None, None


def visible_gpu_count() -> int:
    """GPUs the ranks will see, counted in a THROW-AWAY child process: this process must not touch HIP before it starts
    the ranks (a parent that has initialised the GPU may not start or exec other GPU programs on this pool), and whether
    ``torch.cuda.device_count()`` initialises the runtime depends on the build (without amdsmi it falls through to
    hipGetDeviceCount).  -1 = could not tell (the ranks themselves then report a missing device)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return -1


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU)
    BEFORE this process makes any GPU call, and return their exit code.  Rank 0 of the children prints the JSON line.
    SC_BENCH_SHARE_GPU=1 (with SC_DIST_BACKEND=gloo: RCCL ranks cannot share a device) lets the ranks share the visible
    GPUs round-robin -- the rehearsal of this launch path on a 1-GPU box (tests/test_gpu_ddp.py)."""
    import socket
    import subprocess
    have = visible_gpu_count()
    share = os.environ.get("SC_BENCH_SHARE_GPU", "0") == "1"
    if have < 0:
        have = n                               # unknown: let the ranks find out
    if have < n and not (share and have >= 1):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    if have < n and os.environ.get("SC_DIST_BACKEND", "nccl") == "nccl":
        print("bench.py: SC_BENCH_SHARE_GPU=1 needs SC_DIST_BACKEND=gloo (two RCCL ranks cannot share one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def note(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.time()


def main():
    import faulthandler
    faulthandler.dump_traceback_later(300, repeat=True, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="local batch per GPU")
    ap.add_argument("--model", default="ViT-B-16-gene")
    ap.add_argument("--n-genes", type=int, default=20000)
    ap.add_argument("--loss", default="clip", choices=["clip", "spatial"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-loss-delta", action="store_true")
    ap.add_argument("--no-comm-probe", action="store_true",
                    help="N > 1: skip the extra pass that times the waits on communication (exposed_comm_ms) and every collective")
    ap.add_argument("--grad-checkpointing", action="store_true",
                    help="activation recomputation (LayerNorm outputs, GELU output): configs[4] at 1024 pairs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: e4m3 forward GEMMs of the transformer blocks (BASELINE configs[4]); backward stays bf16")
    ap.add_argument("--graph", default=os.environ.get("SC_GRAPH", "auto"), choices=["auto", "on", "off", "1", "0"],
                    help="the step as ONE hipGraph (spatial_clip_amd/graph.py): auto = capture it, time replay against enqueueing "
                         "over a few untimed steps and keep the faster form; single process, bf16 only")
    ap.add_argument("--residual-stream", default="bf16", choices=["fp32", "bf16"],
                    help="the forward residual stream of the patch towers: bf16 like the reference's autocast (default) or fp32")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env}", file=sys.stderr)
        sys.exit(2)

    # stdout carries ONE JSON line.  Libraries print there too (RCCL's version banner at communicator creation, for one):
    # everything this process writes to file descriptor 1 goes to stderr until the line itself is printed.
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import functools
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm, data, losses, module, net, ops, optim, streams
    rank, local_rank, world = comm.init_from_env(expect_world=args.gpus)     # set_device(LOCAL_RANK) + RCCL group
    if world > 1 and dist.get_world_size() != args.gpus:
        print(f"bench.py: live process group has {dist.get_world_size()} ranks, wanted {args.gpus}", file=sys.stderr)
        sys.exit(2)

    # What the collectives of this run really are, on rank 0's stderr and in the JSON line: backend, RCCL version, the number
    # of ranks RCCL itself reaches (a SUM all-reduce of ones over the communicator the step uses), which code enqueues the
    # data-path collectives and how the gradients are exchanged -- an 8-GPU run evidences itself.
    comm_info = comm.describe()
    note(f"process group: backend={comm_info['backend']} world_size={comm_info['world_size']} rccl_ranks={comm_info['rccl_ranks']} "
         f"rccl_version={comm_info['rccl_version']} route={comm_info['route']} grad_exchange={comm_info['grad_exchange']}")
    if world > 1 and comm_info["backend"] == "nccl" and comm_info["rccl_ranks"] != args.gpus:
        print(f"bench.py: RCCL reaches {comm_info['rccl_ranks']} ranks, wanted {args.gpus}", file=sys.stderr)
        sys.exit(2)
    n = net.SpatialClipNet(args.model, None, n_genes=args.n_genes, seed=0, precision=args.dtype,
                           residual_stream=args.residual_stream)
    cfg = n.cfg
    if args.grad_checkpointing:
        n.model.set_grad_checkpointing(True)
    if args.loss == "clip":
        loss_fn = losses.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True)
    else:
        loss_fn = losses.SpatialLoss(local_loss=True, gather_with_grad=True, cap_logit_scale=40.0, temp_reg_weight=0.05,
                                     neighbor_alpha_scale=0.5, float32_logits=True)
    m = module.SpatialClipLitModule(
        n, loss_fn, functools.partial(optim.FusedAdamW, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1),
        functools.partial(optim.get_cosine_schedule_with_warmup, num_warmup_steps=2000))

    class _T:
        max_steps, max_epochs, estimated_stepping_batches = 1_000_000, None, 1_000_000
    m.trainer = _T()
    oc = m.configure_optimizers()
    opt, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
    reducer = comm.make_grad_exchange(n.store)          # None at N = 1; sharded reduce-scatter / all-gather exchange by default
    n.grad_bucket_hook = reducer.bucket_ready if reducer is not None else None
    opt.attach_exchange(reducer)
    if reducer is None:
        comm_info["grad_exchange"] = "none (one process)"
    else:                         # the exchange that was really built (the sharded route is taken only once its in-place collectives have reproduced an all-reduce on this group)
        comm_info["grad_exchange"] = "sharded" if isinstance(reducer, comm.ShardedGradExchange) else "allreduce"
        comm_info["inplace_collectives_verified"] = comm.describe_inplace_check()
        note(f"gradient exchange: {comm_info['grad_exchange']} (in-place collectives verified: {comm_info['inplace_collectives_verified']})")

    B = args.batch
    rates = data.make_gene_rates(args.n_genes)
    batches = []
    for s in range(2):          # synthetic inputs resident in HBM before the timed region
        b = make_batch(cfg, B, args.n_genes, s, rank, world, rates)
        batches.append({k: v.cuda() for k, v in b.items()})

    from spatial_clip_amd import graph as sc_graph
    gmode = {"1": "on", "0": "off"}.get(args.graph, args.graph)
    gstep = sc_graph.GraphedTrainStep(m, opt, max_norm=1.0, grad_scale=1.0 / world) if (world == 1 and gmode != "off") else None
    use_graph = [False]
    graph_info = {"mode": gmode, "used": False, "why": "off" if gstep is None else None}

    def step(i, eager=False):
        if use_graph[0] and not eager:
            loss = gstep(batches[i % 2])          # static-buffer copy + 12-byte hyper-parameter copy + ONE graph launch
            sched.step()
            return loss
        with streams.chain_stream():       # same stream role as Trainer.fit (high-priority chain, side-stream wgrads)
            return step_(i)

    def step_(i):
        loss = m.training_step(batches[i % 2], i)
        loss.backward(m.root_gradient(loss))
        if reducer is not None:
            reducer.finish()
        opt.step(grad_scale=1.0 / world, max_norm=1.0)
        sched.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Parity half of the metric, at the INITIAL weights (the meaningful point: nothing is memorised yet) ...
    delta_init = None
    if rank == 0 and world == 1 and not args.no_loss_delta:
        init_sd = n.state_dict()
        note("loss delta vs the fp32 oracle at the initial weights (oracle forward on the host)")
        delta_init = loss_delta_vs_oracle(m, n, cfg, make_batch(cfg, B, args.n_genes, 0, 0, 1, rates),
                                          args.loss, args.dtype, "initial weights, benchmark batch 0")
        note(f"init-point loss delta {delta_init['loss_delta_vs_oracle']:.2e}, max feature delta "
             f"{delta_init['max_abs_feature_delta']:.2e} (bound {delta_init['tolerance']:g})")
    # Setup before the warm-up: with SC_OVERLAP unset ("auto") every transformer stack times its own backward with the
    # weight gradients on the side stream and on the chain stream (towers.TransformerStack._overlap_auto: 2 + 4 + 4 calls, one more
    # to decide) and keeps the faster schedule -- the same bits either way.  Like a kernel autotuner, this runs once per
    # process and batch shape, and is not part of the W warm-up or the K timed steps.
    setup_steps = 0
    if os.environ.get("SC_OVERLAP", "auto") not in ("0", "1"):
        from spatial_clip_amd import towers as _towers
        setup_steps = sum(_towers.TransformerStack.OVERLAP_TRIAL_CALLS[1:]) * 2 + _towers.TransformerStack.OVERLAP_TRIAL_CALLS[0] + 1
        note(f"schedule selection ({setup_steps} steps)")
        for i in range(setup_steps):
            step(i)
        torch.cuda.synchronize()
        note(f"weight-gradient side stream per stack: {n.side_stream_choice()}")
    # The step as one hipGraph (graph.py): captured once the schedule is fixed; `auto` keeps it only where replaying beats
    # enqueueing -- it does when the host is the bottleneck (small batches, the two-tower text models), not when the GPU is.
    if gstep is not None:
        why = gstep.capturable()
        if why is None:
            def timed_ms(k, eager):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for i in range(k):
                    step(setup_steps + i, eager=eager)
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / k * 1e3
            use_graph[0] = True
            step(0)                               # (a shape is captured after it has run eagerly once through the wrapper)
            step(1)                               # capture + first replay
            torch.cuda.synchronize()
            if gstep.graph is None:
                use_graph[0] = False
                graph_info["why"] = f"capture failed: {gstep.failed}"
            else:
                ms_g = min(timed_ms(3, False), timed_ms(3, False))
                ms_e = min(timed_ms(3, True), timed_ms(3, True))
                graph_info.update(replay_ms=round(ms_g, 3), enqueue_ms=round(ms_e, 3))
                use_graph[0] = gmode == "on" or ms_g < ms_e * 0.995
                graph_info["why"] = "replay is faster" if use_graph[0] else "enqueueing is at least as fast (GPU-bound step)"
                setup_steps += 14
        else:
            graph_info["why"] = why
        graph_info["used"] = use_graph[0]
        note(f"hipGraph step: {graph_info}")
    note(f"model + {len(batches)} batches resident; warm-up ({args.warmup} steps)")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    fence()
    comm.reset_stats()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the timed steps (GPU-bound if < total)
    fence()
    dt = time.perf_counter() - t0
    coll = {k: {"launches_per_step": round(v[0] / args.steps, 2), "mbytes_per_step": round(v[1] / args.steps / 1e6, 3)}
            for k, v in comm.STATS.items()}
    for k, v in coll.items():
        note(f"collective per step and rank: {k}: {v['launches_per_step']} launches, {v['mbytes_per_step']} MB")
    note(f"timed region done: {dt / args.steps * 1e3:.2f} ms/step (host enqueue {t_host / args.steps * 1e3:.2f} ms/step)")
    # Per-kernel durations: the SAME K steps again with every GEMM launch bracketed by HIP events on its launch stream.
    # Kept out of the timed region above because the 2 x ~150 event markers per step perturb the GPU pipeline (~+10 %).
    # The weight-gradient GEMMs normally run on a side stream beside the chain, which stretches every kernel that
    # shares the chip with them; the kernel's own duration is taken with the side stream off (SC_OVERLAP=0, read by
    # towers at every backward), and the same measurement with it on is reported next to it.
    # N > 1: the same K steps once more with the communication probe on (comm.CommProbe): HIP timing events around every point
    # where the compute stream waits for a collective (-> exposed communication per step) and from the issue of every
    # collective to its completion (-> a lower bound of its achieved bandwidth).  Outside the timed region: the markers are
    # launches of their own.  All ranks run it (the collectives are collective); rank 0 reports its own numbers.
    comm_probe = None
    if world > 1 and not args.no_comm_probe:
        comm.PROBE = comm.CommProbe()
        tp = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + args.steps + i, eager=True)
        t_host_probe = time.perf_counter() - tp
        fence()
        comm_probe = comm.PROBE.summary(args.steps)
        comm.PROBE = None
        comm_probe["probe_pass_host_enqueue_ms_per_step"] = round(t_host_probe / args.steps * 1e3, 3)
        note(f"communication probe: exposed {comm_probe['exposed_comm_ms_per_step']} ms/step; {comm_probe['collectives']}")
    events, dt_inst, events_ov, dt_ov = None, None, None, None
    if not args.no_kernel_events:
        def instrumented(overlap):
            prev = os.environ.get("SC_OVERLAP")
            prev_b = os.environ.get("SC_ADAMW_BEHIND")
            prev_t = os.environ.get("SC_TOWER_OVERLAP")
            os.environ["SC_OVERLAP"] = "1" if overlap else "0"
            if not overlap:                     # ... and nothing else beside a launch either: the optimiser as one launch in front
                os.environ["SC_ADAMW_BEHIND"] = "0"      # of the forward instead of bucket by bucket behind it (read at every step)
                os.environ["SC_TOWER_OVERLAP"] = "0"     # ... and the two towers one after the other (read at every forward / backward)
            for i in range(2):
                step(args.warmup + args.steps + i, eager=True)
            fence()
            ops.KERNEL_EVENTS = []
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + args.steps + 2 + i, eager=True)      # event markers per launch: the enqueued form
            fence()
            d = time.perf_counter() - t1
            ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            if prev is None:
                os.environ.pop("SC_OVERLAP", None)
            else:
                os.environ["SC_OVERLAP"] = prev
            if prev_b is None:
                os.environ.pop("SC_ADAMW_BEHIND", None)
            else:
                os.environ["SC_ADAMW_BEHIND"] = prev_b
            if prev_t is None:
                os.environ.pop("SC_TOWER_OVERLAP", None)
            else:
                os.environ["SC_TOWER_OVERLAP"] = prev_t
            return ev, d
        events, dt_inst = instrumented(False)
        note(f"instrumented pass (single stream) done: {dt_inst / args.steps * 1e3:.2f} ms/step")
        events_ov, dt_ov = instrumented(True)
        note(f"instrumented pass (side stream on) done: {dt_ov / args.steps * 1e3:.2f} ms/step")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if world == 1 or dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    pairs = B * world * args.steps
    G = B * world
    fpp = flops_per_pair(cfg, G)
    value = pairs / dt

    def aggregate(evs):
        agg = {}
        for rec in evs:
            name, fl, (e0, e1) = rec[:3]
            a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
            a[3] += rec[3] if len(rec) > 3 else 0.0
        return agg

    roofline = None
    if events:
        agg = aggregate(events)
        dom = "gemm_nt"
        if "gemm_nt_fp8" in agg:                 # fp8 run: forward launches are the e4m3 kernel; report them beside the bf16 ones
            f8, s8, c8 = agg["gemm_nt_fp8"][:3]
        fl, sec, cnt, alg_bytes = agg[dom]
        traffic, traffic_file = pmc_traffic_nt()
        ach = fl / sec / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm8p_kernel / gemm8pp_kernel (NT: forward + dgrad GEMMs)", "achieved": round(ach, 1),
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "traffic_measured_by_this_run": False, "traffic_from_profile": traffic_file,
                    "algorithmic_bytes_per_launch": round(alg_bytes / cnt),
                    "traffic_over_algorithmic": None if not traffic else round(traffic / (alg_bytes / cnt), 3),
                    "launches_per_step": cnt // args.steps,
                    "avg_launch_us": round(sec / cnt * 1e6, 1), "flops_per_launch_avg": fl / cnt,
                    "share_of_step_time": round(sec / dt_inst, 3),
                    "measured_in": "extra pass over K steps with HIP events around every GEMM launch on its launch stream, "
                                   "weight-gradient side stream OFF (SC_OVERLAP=0), the optimiser as one launch (SC_ADAMW_BEHIND=0) and the towers in sequence (SC_TOWER_OVERLAP=0) so that a launch has the chip to itself",
                    "instrumented_ms_per_step": round(dt_inst / args.steps * 1e3, 3)}
        if events_ov:
            ao = aggregate(events_ov)
            fo, so, co = ao[dom][:3]
            roofline["with_side_stream"] = {"achieved": round(fo / so / 1e12, 1), "avg_launch_us": round(so / co * 1e6, 1),
                                            "instrumented_ms_per_step": round(dt_ov / args.steps * 1e3, 3),
                                            "note": "same launches while the weight-gradient GEMMs share the chip (SC_OVERLAP=1: the schedule "
                                                    "SC_OVERLAP=auto keeps only where it is faster, see side_stream)"}
        if "gemm_nt_fp8" in agg:
            roofline["forward_fp8"] = {"achieved": round(f8 / s8 / 1e12, 1), "peak": 5000.0, "unit": "TFLOP/s",
                                       "frac": round(f8 / s8 / 1e12 / 5000.0, 4), "launches_per_step": c8 // args.steps,
                                       "share_of_step_time": round(s8 / dt_inst, 3),
                                       "note": "e4m3 NT GEMMs on v_mfma_f32_16x16x128_f8f6f4 (dense fp8 peak ~5 PFLOP/s)"}
        if "gemm_tn" in agg:
            fl2, sec2, cnt2 = agg["gemm_tn"][:3]
            roofline["wgrad_tn"] = {"achieved": round(fl2 / sec2 / 1e12, 1), "share_of_step_time": round(sec2 / dt_inst, 3),
                                    "note": "weight-gradient GEMMs (split-K slabs + fused bias-gradient column sums)"}
        step_tflops = value / world * fpp / 1e12
        exe = fpp - skipped_flops_per_pair(cfg)
        roofline["whole_step"] = {"achieved": round(step_tflops, 1), "frac": round(step_tflops / PEAK_BF16_TFLOPS, 4),
                                  "flops_per_pair": fpp, "executed_flops_per_pair": exe,
                                  "achieved_executed": round(value / world * exe / 1e12, 1),
                                  "note": "flops_per_pair = reference-algorithmic (SURVEY 8d); executed excludes the "
                                          "last block's dead non-CLS token work that this build skips"}

    delta = None
    if rank == 0 and world == 1 and not args.no_loss_delta:
        # ... and again at a TRAINED point that does not depend on how many steps this process happened to run: back to the
        # initial weights and optimiser state, then exactly PARITY_TRAINED_STEPS steps over the two resident batches
        note(f"trained-weights parity point: initial weights + {PARITY_TRAINED_STEPS} optimiser steps, then the fp32 oracle and the "
             "oracle under bf16 autocast (the reference's own policy) on the host")
        n.load_state_dict(init_sd)
        opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.step_count = 0
        sched.last_epoch = 0; sched._apply()
        for i in range(PARITY_TRAINED_STEPS):
            step(i)
        torch.cuda.synchronize()
        delta = loss_delta_vs_oracle(m, n, cfg, make_batch(cfg, B, args.n_genes, 0, 0, 1, rates), args.loss, args.dtype,
                                     f"initial weights + {PARITY_TRAINED_STEPS} optimiser steps over the two resident batches, "
                                     "benchmark batch 0", trained=True)
        note(f"trained point: loss delta {delta['loss_delta_vs_oracle']:.2e}, max feature delta {delta['max_abs_feature_delta']:.2e} "
             f"(reference policy's own noise {delta['reference_policy_feature_noise']:.2e}, bound {delta['feature_tolerance']:.2e})")
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle baseline (bounded sample)")
            cpu = cpu_baseline(args.model, args.n_genes, B)
            note("cpu baseline done")
        out = {"metric": "tile-gene pairs/sec (train step)", "value": round(value, 2), "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_steps": setup_steps,
               "side_stream": n.side_stream_choice() if setup_steps else os.environ.get("SC_OVERLAP"),
               "side_stream_trial_ms": n.side_stream_timing() if setup_steps else None,
               "hip_graph": graph_info, "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3),
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{args.model} image tower + " + (
                          f"CLIP text tower(vocab {cfg.text.vocab_size}, {cfg.text.layers} x {cfg.text.width}, context "
                          f"{cfg.text.context_length}, embed {cfg.embed_dim}), " if cfg.gene is None else
                          f"gene-MLP({args.n_genes}->{cfg.gene.hidden}->{cfg.embed_dim}), " if cfg.gene.kind == "mlp" else
                          f"{cfg.gene.layers}-layer gene transformer({args.n_genes} genes -> {cfg.gene.tokens} tokens x "
                          f"{cfg.gene.width}, embed {cfg.embed_dim}), ") + (
                                      f"{cfg.vision.image_size}x{cfg.vision.image_size} tiles, local batch {B}, "
                                      f"{'ClipLoss' if args.loss == 'clip' else 'SpatialLoss(k=8)'} over global batch {G}, "
                                      "fwd+bwd+grad-allreduce+clip+AdamW" +
                                      (", activation recomputation (LN outputs, GELU output)" if args.grad_checkpointing else "")),
                          "global_batch": G, "residual_stream": n.residual_stream if not os.environ.get("SC_RES_STREAM")
                          else os.environ["SC_RES_STREAM"],
                          "parallelism": f"dp{world}", "loss": float(loss.detach())},
               "n_gpus_live": dist.get_world_size() if world > 1 else 1,
               "rccl_ranks": comm_info["rccl_ranks"],
               "comm": {**comm_info, "collectives_per_step_and_rank": coll,
                        "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3), **(comm_probe or {})},
               "hbm_peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
               "loss_delta_vs_oracle": None if delta_init is None else delta_init["loss_delta_vs_oracle"],
               "max_abs_feature_delta": None if delta_init is None else delta_init["max_abs_feature_delta"],
               "loss_tolerance": _tolerances().LOSS_TOLERANCE[args.dtype],
               "parity": None if delta_init is None else {
                   "initial_weights": delta_init, "after_training_steps": delta,
                   "within_tolerance": bool(delta_init["within_tolerance"] and delta["within_tolerance"])},
               "roofline": roofline, "cpu_baseline": cpu}
        sys.stdout.flush()
        os.dup2(_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    comm.shutdown()


if __name__ == "__main__":
    main()
