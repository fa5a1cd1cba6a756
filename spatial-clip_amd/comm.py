"""Data-parallel exchange steps of the training step over ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for the world_size-2 correctness tests).  One process per GPU.

Process bootstrap (reference: Lightning ``strategy: ddp`` creates the group inside ``trainer.fit``,
configs/trainer/ddp.yaml:4, src/train.py:102,119; legacy loop src/open_clip_train/distributed.py:93-195):
  init_from_env()         reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun), binds this process to
                          cuda:LOCAL_RANK *before* anything allocates, and creates the RCCL process group.

Collectives of the path:
  C1+C3  FeatureGather     the overlapped form used by the product: the second tower's features | tile ids are packed
                           by one HIP kernel and all-gathered on a dedicated communication stream while the vision
                           tower still runs; the image features follow and travel while the first similarity GEMM
                           (which does not need them) computes.  Event hand-off, no host synchronisation.
         gather_packed     the synchronous one-collective form (tests, and the fallback when nothing was prefetched)
                           (reference: 2x torch.distributed.nn.all_gather + 2x dist.all_gather,
                           src/open_clip/loss.py:50-52, src/models/components/losses.py:63-68)
  C1'    reduce_scatter_sum  autograd of the feature all-gather: sum over ranks of d(all_features), keep own rows
  C4     GradBucketReducer   bucketed SUM all-reduce of the flat fp32 gradient buffer, launched per layer while
                           backward is still running (RCCL runs on its own stream; the 1/world_size of DDP's mean
                           is folded into the optimiser's grad_scale)
``SC_COMM_NATIVE=1`` routes the three data-path collectives through the kernel library's own RCCL entry points
(``sc_allgather_feats_async`` / ``sc_reduce_scatter_grads_async`` / ``sc_allreduce_sum_async`` on explicit HIP streams,
include/spatial_clip_hip.h) instead of ``torch.distributed``; the process group is then only the bootstrap channel for the
128-byte communicator id.
All functions are device-agnostic (CUDA/HIP or CPU tensors) and degrade to no-ops at world_size 1
(``SC_FORCE_DIST=1`` keeps the collective code path alive on a 1-rank group: used to exercise RCCL on a 1-GPU box)."""
from __future__ import annotations

import os
import sys
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def _force() -> bool:
    return os.environ.get("SC_FORCE_DIST", "0") == "1"


# Per-process collective counters (data path only): kind -> [launches, payload bytes this rank contributes / receives].
# bench.py resets them before the timed steps and prints per-step figures on rank 0, so that a multi-GPU run shows on its
# own stderr what travelled over xGMI and by which route.
STATS = {}


def _count(kind: str, nbytes: int) -> None:
    c = STATS.setdefault(kind, [0, 0])
    c[0] += 1
    c[1] += int(nbytes)


def reset_stats() -> None:
    STATS.clear()


class CommProbe:
    """bench.py's instrumented pass at N > 1 (round-5 verdict item 6): where the compute stream WAITS for communication, and
    how long each collective takes from the point it is issued.

    ``stall(kind, device, wait_fn)``: ``wait_fn`` makes the current stream wait for a collective (an event wait, a Work.wait,
    or a collective that runs on the critical path itself); two timing events on the current stream bracket it -- the time
    between them is communication the step did NOT hide (exposed communication).
    ``issued(kind, nbytes, device)`` -> token; ``completed(token, waiter)``: a timing event on the issuing stream in front of
    the collective and one on the probe's own stream behind ``waiter`` (which makes the probe stream wait for the collective):
    issue-to-completion time, an upper bound of the collective's duration (it includes any queueing in front of it), hence a
    LOWER bound of its achieved bandwidth.  Off (``PROBE is None``) in the timed region: the markers cost launches."""

    def __init__(self):
        self.stalls, self.colls = [], []
        self._stream = {}

    def stream(self, device):
        key = torch.device(device).index
        s = self._stream.get(key)
        if s is None:
            s = self._stream[key] = torch.cuda.Stream(device=device)
        return s

    def summary(self, steps: int) -> dict:
        torch.cuda.synchronize()
        ex, co = {}, {}
        for kind, e0, e1 in self.stalls:
            ex[kind] = ex.get(kind, 0.0) + e0.elapsed_time(e1)
        for kind, nbytes, e0, e1 in self.colls:
            c = co.setdefault(kind, [0, 0.0, 0.0])
            c[0] += 1
            c[1] += e0.elapsed_time(e1)
            c[2] += nbytes
        return {"exposed_comm_ms_per_step": round(sum(ex.values()) / max(steps, 1), 4),
                "exposed_by_wait_ms_per_step": {k: round(v / max(steps, 1), 4) for k, v in ex.items()},
                "collectives": {k: {"launches_per_step": round(c[0] / max(steps, 1), 2), "avg_issue_to_done_ms": round(c[1] / max(c[0], 1), 4),
                                    "payload_gbytes_per_s_lower_bound": round(c[2] / max(c[1], 1e-9) / 1e6, 2)} for k, c in co.items()}}


PROBE: Optional[CommProbe] = None


def _stalled(kind: str, device, wait_fn) -> None:
    """Run ``wait_fn`` (which makes the current stream wait for communication); with the probe on, bracket it with timing
    events on the current stream."""
    if PROBE is None or torch.device(device).type != "cuda":
        wait_fn()
        return
    cur = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    wait_fn()
    e1.record(cur)
    PROBE.stalls.append((kind, e0, e1))


def _issued(kind: str, nbytes: int, device, stream=None):
    if PROBE is None or torch.device(device).type != "cuda":
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(stream if stream is not None else torch.cuda.current_stream(device))
    return (kind, int(nbytes), e0, device)


def _completed(token, waiter) -> None:
    """``waiter()`` is called under the probe's own stream and must make THAT stream wait for the collective."""
    if token is None or PROBE is None:
        return
    kind, nbytes, e0, device = token
    ps = PROBE.stream(device)
    with torch.cuda.stream(ps):
        waiter()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(ps)
    PROBE.colls.append((kind, nbytes, e0, e1))


def route() -> str:
    """Which code enqueues the data-path collectives: 'native' (the kernel library's own RCCL calls on explicit HIP streams,
    SC_COMM_NATIVE=1) or 'torch.distributed' (ProcessGroupNCCL = RCCL on ROCm, or gloo in the CPU / shared-GPU rehearsals)."""
    return "native" if _native is not None else "torch.distributed"


def describe() -> dict:
    """Facts about the live group for the bench's stderr / JSON line: backend, ranks the RCCL communicator spans (0 = no RCCL
    communicator exists: plain single-process run or gloo), RCCL version, route, gradient-exchange mode."""
    info = {"backend": None, "world_size": 1, "rccl_ranks": 0, "rccl_version": None, "route": route(),
            "grad_exchange": grad_exchange_mode(), "inplace_collectives_verified": None}
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend()
        info["inplace_collectives_verified"] = describe_inplace_check()
        info["world_size"] = dist.get_world_size()
        if info["backend"] == "nccl":
            # a SUM all-reduce of ones over the communicator the step uses: the number of ranks RCCL itself reaches
            # (SC_COMM_NATIVE=1: through the library's own communicator -- THAT is the one the step's collectives use)
            t = torch.ones(4, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
            if _native is not None:
                _native.all_reduce(t, torch.cuda.current_stream(t.device))
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            info["rccl_ranks"] = int(round(float(t.cpu()[0])))
            try:
                info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                info["rccl_version"] = "unknown"
    return info


def describe_inplace_check() -> Optional[bool]:
    """Verdict of ``inplace_collectives_verified`` on the live group (None: not run -- no group, or the all-reduce route was asked for)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    checks = [v for (bk, _), v in _INPLACE_CHECK.items() if bk == dist.get_backend()]
    return all(checks) if checks else None


def grad_exchange_mode() -> str:
    """SC_GRAD_EXCHANGE = 'allreduce' (default: bucketed SUM all-reduce of the flat gradient overlapped with backward, AdamW
    replicated on every rank -- the reference's DDP shape) or 'sharded' (opt-in: per-bucket reduce-scatter -> AdamW on this
    rank's 1/W of every bucket -> all-gather of the refreshed weights behind the next forward).  Same weights
    either way (tests/test_gpu_ddp.py).  The sharded route was the default in round 5; it has never run on more than one real
    RCCL rank (no multi-GPU box is available to this build), so the route that ships by default is the one whose collective
    -- a plain out-of-place-free all-reduce -- carries no aliasing assumption (advisor, round 5)."""
    return os.environ.get("SC_GRAD_EXCHANGE", "allreduce")


def is_dist() -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _force()


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as the launcher exported them (torchrun / torch.distributed.run)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend: Optional[str] = None, expect_world: Optional[int] = None) -> Tuple[int, int, int]:
    """Bind this process to its GPU and join the job's process group; idempotent.  Returns (rank, local_rank, world).

    Must run before the model is built: ``SpatialClipNet`` allocates on the *current* device.  ``expect_world``
    (e.g. ``trainer.devices * num_nodes``) must match what the launcher started, otherwise this raises instead of
    silently training W independent replicas."""
    rank, local_rank, W = env_world()
    if expect_world is not None and expect_world != W:
        raise RuntimeError(
            f"the configuration asks for {expect_world} ranks but the launcher started WORLD_SIZE={W}: launch with "
            f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {expect_world} --master-addr 127.0.0.1 ...` "
            "(one process per GPU)")
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        n_dev = torch.cuda.device_count()
        # several ranks may share one device only in the gloo rehearsal tests (SC_DIST_BACKEND=gloo)
        torch.cuda.set_device(local_rank % max(n_dev, 1))
    if W > 1 or _force():
        if not dist.is_initialized():
            backend = backend or os.environ.get("SC_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl":
                if local_rank >= torch.cuda.device_count():
                    raise RuntimeError(f"LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPUs are visible")
                kw["device_id"] = torch.device(f"cuda:{local_rank}")      # eager RCCL communicator on this GPU
                # RCCL's kernels on high-priority streams: a bucket all-reduce / feature gather launched beside the
                # backward should get its few CUs as soon as a GEMM workgroup retires, not after the queued chain
                # kernels (SC_RCCL_HIGH_PRIO=0 keeps the default)
                if os.environ.get("SC_RCCL_HIGH_PRIO", "1") == "1":
                    try:
                        opts = dist.ProcessGroupNCCL.Options()
                        opts.is_high_priority_stream = True
                        kw["pg_options"] = opts
                    except (AttributeError, TypeError):
                        pass
            dist.init_process_group(backend, rank=rank, world_size=W, **kw)
            if backend == "nccl" and os.environ.get("SC_COMM_NATIVE", "0") == "1":
                global _native
                _native = NativeComm(rank, W, torch.device(f"cuda:{local_rank}"))
        elif dist.get_world_size() != W or dist.get_rank() != rank:
            raise RuntimeError(f"live process group is rank {dist.get_rank()}/{dist.get_world_size()} but the "
                               f"environment says {rank}/{W}")
    return rank, local_rank, W


class NativeComm:
    """RCCL communicator owned through the C ABI (sc_comm_*): collectives are enqueued on a caller-chosen HIP stream by
    the kernel library itself, no torch.distributed object on the data path."""

    def __init__(self, rank: int, world_size: int, device: torch.device):
        import ctypes
        from . import _lib
        self._l, self._check = _lib.lib(), _lib.check
        self.rank, self.world_size, self.device = rank, world_size, device
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self._l.sc_comm_unique_id(buf), "sc_comm_unique_id")
        box = [buf.raw if rank == 0 else None]
        if world_size > 1:
            dist.broadcast_object_list(box, src=0)          # bootstrap: the existing process group carries the id
        with torch.cuda.device(device):
            self.handle = self._l.sc_comm_init(box[0], rank, world_size)
        if not self.handle:
            raise RuntimeError("sc_comm_init failed: " + self._l.sc_last_error().decode())
        self.stream = torch.cuda.Stream(device=device, priority=-1)     # gradient buckets travel here
        self.launched = 0

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor, stream) -> None:
        self._check(self._l.sc_allgather_feats_async(self.handle, send.data_ptr(), recv.data_ptr(),
                                                     send.numel() * send.element_size(), stream.cuda_stream),
                    "sc_allgather_feats_async")
        self.launched += 1

    def reduce_scatter(self, send: torch.Tensor, recv: torch.Tensor, stream) -> None:
        self._check(self._l.sc_reduce_scatter_grads_async(self.handle, send.data_ptr(), recv.data_ptr(), recv.numel(),
                                                          stream.cuda_stream), "sc_reduce_scatter_grads_async")
        self.launched += 1

    def all_reduce(self, buf: torch.Tensor, stream) -> None:
        self._check(self._l.sc_allreduce_sum_async(self.handle, buf.data_ptr(), buf.numel(), stream.cuda_stream),
                    "sc_allreduce_sum_async")
        self.launched += 1

    def destroy(self) -> None:
        if self.handle:
            torch.cuda.synchronize(self.device)
            self._l.sc_comm_destroy(self.handle)
            self.handle = 0


_native: Optional[NativeComm] = None


def native() -> Optional[NativeComm]:
    return _native


def shutdown() -> None:
    global _native
    if _native is not None:
        _native.destroy()
        _native = None
    _INPLACE_CHECK.clear()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- logged scalars
def _scalar_device() -> torch.device:
    """Where a scalar collective's tensor must live: the GPU for nccl (RCCL), anything for gloo."""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_reduce_mean_scalars(values: List[float]) -> List[float]:
    """Mean over ranks of a few host scalars in ONE collective (``self.log(..., sync_dist=True)``,
    src/models/spatial_clip_module.py:105,107: Lightning reduces logged values with mean over the group).  Every rank
    must call it with the same number of values.  No-op without a group."""
    if not is_dist() or not values:
        return list(values)
    _, W = world()
    t = torch.tensor(values, dtype=torch.float64, device=_scalar_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) / W for v in t.cpu()]


def broadcast_flag(flag: bool, src: int = 0) -> bool:
    """Rank ``src``'s decision for everybody (early stopping: all ranks must leave the epoch loop together, or the ones
    that stay block in the next collective)."""
    if not is_dist():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_scalar_device())
    dist.broadcast(t, src=src)
    return bool(int(t.cpu()[0]))


def broadcast_flag_and_value(flag: bool, value: float, src: int = 0):
    """Rank ``src``'s (decision, score) for everybody -- one float64 collective.  The best-checkpoint bookkeeping must
    compare later epochs against the SAME best score on every rank, also when the monitored value is rank-local."""
    if not is_dist():
        return bool(flag), float(value)
    t = torch.tensor([1.0 if flag else 0.0, float(value)], dtype=torch.float64, device=_scalar_device())
    dist.broadcast(t, src=src)
    t = t.cpu()
    return bool(t[0] != 0.0), float(t[1])


# ---------------------------------------------------------------------------------------------- packing
def _pack(feat: torch.Tensor, ids_a: Optional[torch.Tensor], ids_b: Optional[torch.Tensor], out: torch.Tensor) -> None:
    """out[B, D (+4)] = feat | ids_a (int64 as 2 floats) | ids_b.  Device tensors take the HIP pack kernel."""
    B, D = feat.shape
    if feat.is_cuda:
        from . import ops
        ops.pack_rows(feat, ids_a, ids_b, out)
        return
    out[:, :D] = feat
    if ids_a is not None:
        out[:, D:D + 2].view(torch.int64).copy_(ids_a.view(B, 1))
        out[:, D + 2:D + 4].view(torch.int64).copy_(ids_b.view(B, 1))


def _ids_view(buf: torch.Tensor, D: int, which: int) -> torch.Tensor:
    """int64 [G] tile ids out of the packed [G, D+4] fp32 gather buffer (no copy: strided int64 view)."""
    G, cols = buf.shape
    flat = buf.view(-1).view(torch.int64)          # cols is even (D is a multiple of 2, +4)
    return flat.as_strided((G,), (cols // 2,), (D // 2 + which))


def gather_packed(image_features: torch.Tensor, text_features: torch.Tensor,
                  image_tile_ids: Optional[torch.Tensor] = None, text_tile_ids: Optional[torch.Tensor] = None):
    """Rank-major concatenation of every rank's rows: returns (all_image, all_text, all_image_ids, all_text_ids).
    Equal local batch on every rank is assumed, as by the reference (loss.py:96, losses.py:94).  Synchronous form:
    ONE collective on the caller's stream."""
    if not is_dist():
        return image_features, text_features, image_tile_ids, text_tile_ids
    _, W = world()
    B, D = image_features.shape
    with_ids = image_tile_ids is not None
    cols = 2 * D + (4 if with_ids else 0)
    packed = torch.empty((B, cols), dtype=torch.float32, device=image_features.device)
    packed[:, :D] = image_features
    packed[:, D:2 * D] = text_features
    if with_ids:
        packed[:, 2 * D:2 * D + 2].view(torch.int64).copy_(image_tile_ids.view(B, 1))
        packed[:, 2 * D + 2:2 * D + 4].view(torch.int64).copy_(text_tile_ids.view(B, 1))
    out = torch.empty((W * B, cols), dtype=torch.float32, device=packed.device)
    _count("all_gather(features|ids)", out.numel() * 4)
    if _native is not None and packed.is_cuda:
        _native.all_gather(packed, out, torch.cuda.current_stream(packed.device))
    else:
        dist.all_gather_into_tensor(out, packed)
    all_i = out[:, :D].contiguous()
    all_t = out[:, D:2 * D].contiguous()
    if with_ids:
        ids_i = out[:, 2 * D:2 * D + 2].contiguous().view(torch.int64).view(-1)
        ids_t = out[:, 2 * D + 2:2 * D + 4].contiguous().view(torch.int64).view(-1)
        return all_i, all_t, ids_i, ids_t
    return all_i, all_t, None, None


class FeatureGather:
    """Overlapped global-batch formation (north-star: "RCCL all-gather ... overlapped with the final encoder block on
    a side HIP stream").

    Per step:  ``begin(ids_i, ids_t)`` -> ``put("text", f_t)`` as soon as the second tower has produced its features
    (its rows + both id vectors are packed and all-gathered on the communication stream while the vision tower runs)
    -> ``put("image", f_i)`` after the vision tower -> ``take("text")`` / ``take("image")`` make the *current* stream
    wait for the matching gather (an event wait, never the host) and return the gathered tensors.

    The communication stream owns persistent send / receive buffers, so nothing allocated on the compute stream is
    ever touched by another stream (no caching-allocator hazards), and the result is bit-identical to
    ``gather_packed``: an all-gather moves bytes."""

    def __init__(self, device: torch.device):
        self.device = device
        self.cuda = device.type == "cuda"
        self.stream = None
        if self.cuda:
            from . import streams
            self.stream = streams.comm_stream(device)          # shared with the sharded optimiser's weight gathers
        self._send, self._recv, self._done, self._src = {}, {}, {}, {}
        self._ids = (None, None)
        self.launched = 0          # number of collectives issued (tests)

    def begin(self, ids_i: Optional[torch.Tensor], ids_t: Optional[torch.Tensor]) -> None:
        self._ids = (ids_i, ids_t)
        self._done.clear()
        self._src.clear()

    def _buf(self, table, key, shape):
        t = table.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            table[key] = t
        return t

    def put(self, which: str, feat: torch.Tensor) -> None:
        if not is_dist():
            return
        _, W = world()
        B, D = feat.shape
        ids_i, ids_t = self._ids if which == "text" else (None, None)
        cols = D + (4 if ids_i is not None else 0)
        send = self._buf(self._send, which, (B, cols))
        recv = self._buf(self._recv, which, (W * B, cols))
        self._src[which] = (feat.data_ptr(), D, ids_i is not None)
        _count("all_gather(features|ids)", recv.numel() * 4)
        if not self.cuda:
            _pack(feat, ids_i, ids_t, send)
            dist.all_gather_into_tensor(recv, send)
            self._done[which] = None
            self.launched += 1
            return
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.stream.wait_event(ready)               # the features are final on the compute stream
        with torch.cuda.stream(self.stream):
            _pack(feat, ids_i, ids_t, send)
            tok = _issued("all_gather(features|ids)", recv.numel() * 4, self.device, self.stream)
            if _native is not None:
                _native.all_gather(send, recv, self.stream)     # RCCL's kernel is enqueued on the communication stream itself
            else:
                work = dist.all_gather_into_tensor(recv, send, async_op=True)
                work.wait()                         # nccl: the communication stream waits for RCCL's stream (no host block)
            done = torch.cuda.Event()
            done.record(self.stream)
        _completed(tok, lambda: torch.cuda.current_stream(self.device).wait_event(done))
        self._done[which] = done
        self.launched += 1

    def has(self, which: str, feat: torch.Tensor) -> bool:
        src = self._src.get(which)
        return src is not None and src[0] == feat.data_ptr() and which in self._done

    def with_ids(self, which: str) -> bool:
        src = self._src.get(which)
        return src is not None and src[2]

    def take(self, which: str):
        """-> (all_features [G, D] (row stride D or D+4), all_ids_i, all_ids_t)  (ids only for "text")."""
        done = self._done[which]
        if done is not None:
            _stalled(f"feature all-gather ({which})", self.device, lambda: torch.cuda.current_stream(self.device).wait_event(done))
        _, D, with_ids = self._src[which]
        recv = self._recv[which]
        feats = recv[:, :D]
        if with_ids:
            return feats, _ids_view(recv, D, 0), _ids_view(recv, D, 1)
        return feats, None, None


def reduce_scatter_sum(full: torch.Tensor) -> torch.Tensor:
    """full: [W*B, C] per-rank contribution to every rank's rows -> this rank's [B, C] summed over ranks."""
    if not is_dist():
        return full
    rank, W = world()
    B = full.shape[0] // W
    full = full.contiguous()
    _count("reduce_scatter(d features)", full.numel() * full.element_size())
    if dist.get_backend() == "gloo":          # gloo has no reduce_scatter: all-reduce and slice
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        return full[rank * B:(rank + 1) * B].clone()
    out = torch.empty((B,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)

    def run():
        if _native is not None and full.is_cuda and full.dtype == torch.float32:
            _native.reduce_scatter(full, out, torch.cuda.current_stream(full.device))
        else:
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM)
    _stalled("reduce_scatter(d features): on the critical path", full.device, run)
    return out


class GradBucketReducer:
    """Overlapped gradient reduction.  ``bucket_ready(lo, hi)`` is called by backward as soon as the flat-gradient
    range [lo, hi) is final; ranges are coalesced until ``bucket_floats`` is reached and then all-reduced
    asynchronously.  ``finish()`` flushes and waits (on the compute stream, no host sync for nccl)."""

    def __init__(self, flat_grad: torch.Tensor, bucket_floats: int = 16 * 1024 * 1024):
        self.flat = flat_grad
        self.bucket_floats = bucket_floats
        self.pending: Optional[Tuple[int, int]] = None
        self.works: List = []
        self.launched: List[Tuple[int, int]] = []

    def bucket_ready(self, lo: int, hi: int) -> None:
        if not is_dist():
            return
        if self.pending is None:
            self.pending = (lo, hi)
        else:
            plo, phi = self.pending
            if hi == plo or lo == phi or (lo <= phi and hi >= plo):      # adjacent / overlapping: merge
                self.pending = (min(lo, plo), max(hi, phi))
            else:
                self._launch(*self.pending)
                self.pending = (lo, hi)
        plo, phi = self.pending
        # Backward completes the flat buffer back to front, so what is announced LAST (the first blocks and the stem, offsets
        # below one bucket) is the part of the exchange nothing is left to hide behind: it travels in quarter-size buckets, so
        # that all but the last ~4 M floats are already on the wire when backward ends (the exposed tail of the all-reduce).
        thresh = self.bucket_floats if plo >= self.bucket_floats else max(1, self.bucket_floats // 4)
        if phi - plo >= thresh:
            self._launch(plo, phi)
            self.pending = None

    def _reduce(self, lo: int, hi: int):
        """Enqueue the SUM all-reduce of flat[lo:hi] behind the caller's stream; returns what finish() waits on."""
        _count("all_reduce(gradient bucket)", (hi - lo) * 4)
        if _native is not None and self.flat.is_cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            _native.stream.wait_event(ready)
            _native.all_reduce(self.flat[lo:hi], _native.stream)
            done = torch.cuda.Event()
            done.record(_native.stream)
            return done
        return dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)

    def _launch(self, lo: int, hi: int) -> None:
        self.launched.append((lo, hi))
        tok = _issued("all_reduce(gradient bucket)", (hi - lo) * 4, self.flat.device) if self.flat.is_cuda else None
        w = self._reduce(lo, hi)
        self.works.append(w)
        if tok is not None:
            _completed(tok, (lambda: torch.cuda.current_stream(self.flat.device).wait_event(w)) if isinstance(w, torch.cuda.Event)
                       else (lambda: w.wait()))

    def finish(self) -> None:
        if not is_dist():
            return
        _stalled("gradient all-reduce (finish)", self.flat.device, self._finish)

    def _finish(self) -> None:
        if self.pending is not None:
            self._launch(*self.pending)
            self.pending = None
        covered = sum(hi - lo for lo, hi in self.launched)
        if covered < self.flat.numel():
            # anything backward did not announce (e.g. logit_scale): reduce the complement in one go
            done = sorted(self.launched)
            pos = 0
            for lo, hi in done + [(self.flat.numel(), self.flat.numel())]:
                if lo > pos:
                    self.works.append(self._reduce(pos, lo))
                pos = max(pos, hi)
        for w in self.works:
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream(self.flat.device).wait_event(w)
            else:
                w.wait()
        self.works, self.launched = [], []


# ---------------------------------------------------------------------------------------------- sharded gradient exchange
def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ShardedGradExchange:
    """gradient **reduce-scatter** per bucket (overlapped with backward) -> grad-norm + AdamW on this rank's
    1/W of every bucket -> **all-gather** of the refreshed fp32 masters per bucket, overlapped with the NEXT forward, which
    waits per bucket (``ParamStore.wait_range``).  Replaces Lightning's DDP mean all-reduce + replicated optimiser
    (configs/trainer/ddp.yaml:4, src/open_clip_train/main.py:300-310) with the same arithmetic: SUM over ranks, 1/W folded
    into the clip / AdamW kernels, so the weights equal the all-reduce route's (tests/test_gpu_ddp.py).

    Layout: the flat buffers are cut into static buckets of ``bucket_floats`` (rounded to a multiple of 64 W); rank r owns the
    r-th of the W equal pieces of every bucket (both collectives run IN PLACE on the flat gradient / master buffers: the
    reduced piece lands where the AdamW kernel reads it, the updated piece is gathered from where AdamW wrote it -- on one
    rank both are no-ops).  Per step and rank: (W-1)/W of the gradient bytes out and in for the reduce-scatter, the same for
    the all-gather -- the bytes of one all-reduce -- but AdamW, the norm and the Adam moments shrink to 1/W and the second
    half travels behind the next forward instead of in front of the optimiser.
    What is NOT done: gathering the bf16 mirror instead of the fp32 masters (half of the all-gather bytes).  The masters are
    read in fp32 by LayerNorm / bias / embedding kernels, by the K-padded weight copies, by the e4m3 weight quantisers and by
    every checkpoint; keeping them whole on every rank keeps all of those exact and collective-free."""

    def __init__(self, store, bucket_floats: int = 16 * 1024 * 1024):
        self.store = store
        self.flat = store.grad
        self.rank, self.W = world()
        U = 64 * self.W
        if store.total % U:
            raise RuntimeError(f"the parameter store ({store.total} floats) was not padded for {self.W} ranks: build the model "
                               "after comm.init_from_env()")
        size = _round_up(max(int(bucket_floats), U), U)
        edges = list(range(0, store.total, size)) + [store.total]
        if len(edges) > 2 and edges[-1] - edges[-2] < size // 2:      # a short tail joins the bucket in front of it
            edges.pop(-2)
        self.buckets: List[Tuple[int, int]] = [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]
        self.cuda = self.flat.is_cuda
        self.stream = None
        if self.cuda:
            from . import streams
            self.stream = streams.comm_stream(self.flat.device)     # the step's one communication stream (streams.comm_stream)
        # order in which the next forward consumes the buckets: the second tower runs first and its parameters sit at the
        # END of the flat buffers (forward order of the vision tower, then the second tower, then logit_scale)
        first_second = min((s.offset for s in store.specs if not s.name.startswith("visual.")), default=0)
        k0 = next(k for k, (lo, hi) in enumerate(self.buckets) if lo <= first_second < hi)
        self.order = list(range(k0, len(self.buckets))) + list(range(0, k0))
        # a Linear weight's derived copies are rebuilt with the LATEST-arriving bucket it touches
        pos = {k: i for i, k in enumerate(self.order)}
        self._copies: List[list] = [[] for _ in self.buckets]
        for c in store.copies.values():
            sp = store.by_name[c.name]
            touched = [k for k, (lo, hi) in enumerate(self.buckets) if lo < sp.offset + sp.numel and sp.offset < hi]
            self._copies[max(touched, key=lambda k: pos[k])].append(c)
        self._plans = [store._transpose_plan(cs) for cs in self._copies]
        self._ready: List[Tuple[int, int]] = []
        self._launched = [False] * len(self.buckets)
        self._works: List = []
        self.rs_launched = 0
        self.ag_launched = 0

    # ---- this rank's piece of bucket k, as offsets into the flat buffers
    def piece(self, k: int) -> Tuple[int, int]:
        lo, hi = self.buckets[k]
        c = (hi - lo) // self.W
        return lo + self.rank * c, lo + (self.rank + 1) * c

    def shard_floats(self) -> int:
        return self.store.total // self.W

    # ---- backward side: reduce-scatter
    def bucket_ready(self, lo: int, hi: int) -> None:
        """Called by backward when flat_grad[lo:hi] is final (same contract as GradBucketReducer.bucket_ready)."""
        if not is_dist():
            return
        self._ready.append((lo, hi))
        self._launch_covered()

    def _covered(self, lo: int, hi: int) -> bool:
        pos = lo
        for a, b in sorted(self._ready):
            if a > pos:
                break
            pos = max(pos, b)
            if pos >= hi:
                return True
        return pos >= hi

    def _launch_covered(self, force: bool = False) -> None:
        for k, (lo, hi) in enumerate(self.buckets):
            if not self._launched[k] and (force or self._covered(lo, hi)):
                self._launched[k] = True
                self._works.append(self._reduce_scatter(k))

    def _reduce_scatter(self, k: int):
        lo, hi = self.buckets[k]
        a, b = self.piece(k)
        _count("reduce_scatter(gradient bucket)", (hi - lo) * 4)
        self.rs_launched += 1
        if _native is not None and self.cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            _native.stream.wait_event(ready)
            _native.reduce_scatter(self.flat[lo:hi], self.flat[a:b], _native.stream)
            done = torch.cuda.Event()
            done.record(_native.stream)
            return done
        if dist.get_backend() == "gloo":          # gloo has no reduce_scatter: all-reduce the bucket, the own piece is part of it
            return dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
        if os.environ.get("SC_SHARD_DEBUG_SKIP_RS") == "1" and self.W == 1:      # A/B probe (one rank: the collective is an identity)
            return torch.cuda.Event()
        return dist.reduce_scatter_tensor(self.flat[a:b], self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self) -> None:
        """Flush what backward did not announce and make the current stream wait for every reduce-scatter."""
        if not is_dist():
            return
        _stalled("gradient reduce-scatter (finish)", self.flat.device, self._finish)

    def _finish(self) -> None:
        self._launch_covered(force=True)
        for w in self._works:
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream(self.flat.device).wait_event(w)
            else:
                w.wait()
        self._works, self._ready = [], []
        self._launched = [False] * len(self.buckets)

    # ---- optimiser side
    def all_reduce_partials(self, partial: torch.Tensor) -> None:
        """SUM over ranks of the fp64 sum-of-squares partials (a few KiB) on the current stream."""
        _count("all_reduce(grad-norm partials)", partial.numel() * 8)
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)

    def gather_bucket(self, k: int) -> None:
        """All-gather the updated masters of bucket k (this rank's piece was just written by AdamW on the current stream) on
        the communication stream, rebuild the bf16 operands that depend on it there, and leave an event for the forward."""
        st = self.store
        lo, hi = self.buckets[k]
        a, b = self.piece(k)
        _count("all_gather(weights bucket)", (hi - lo) * 4)
        self.ag_launched += 1
        if not self.cuda:
            dist.all_gather_into_tensor(st.master[lo:hi], st.master[a:b].clone())
            st.refresh_range(lo, hi, self._copies[k], self._plans[k], fresh=(a, b))
            return
        cur = torch.cuda.current_stream(self.flat.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            if _native is not None:
                _native.all_gather(st.master[a:b], st.master[lo:hi], self.stream)
            elif dist.get_backend() == "gloo":
                dist.all_gather_into_tensor(st.master[lo:hi], st.master[a:b].clone())
            else:
                work = dist.all_gather_into_tensor(st.master[lo:hi], st.master[a:b], async_op=True)
                work.wait()                         # the communication stream waits for RCCL's stream (no host block)
            st.refresh_range(lo, hi, self._copies[k], self._plans[k], fresh=(a, b))
            done = torch.cuda.Event()
            done.record(self.stream)
        st.add_pending(lo, hi, done)

    def gather_moments(self, shard: torch.Tensor) -> torch.Tensor:
        """A full flat-layout copy of a sharded optimiser-state buffer (checkpoints): collective, every rank calls it."""
        full = torch.zeros(self.store.total, dtype=shard.dtype, device=shard.device)
        off = 0
        for k, (lo, hi) in enumerate(self.buckets):
            c = (hi - lo) // self.W
            dist.all_gather_into_tensor(full[lo:hi], shard[off:off + c].clone())
            off += c
        return full

    def scatter_moments(self, full: torch.Tensor, shard: torch.Tensor) -> None:
        off = 0
        for k in range(len(self.buckets)):
            a, b = self.piece(k)
            shard[off:off + (b - a)].copy_(full[a:b])
            off += b - a


_INPLACE_CHECK: dict = {}


INPLACE_CHECK_FLOATS = 4 * 1024 * 1024        # per rank piece of the start-up check (16 MB): a real bucket piece, not a toy


def inplace_collectives_verified(device) -> bool:
    """Start-up check of the two IN-PLACE collectives the sharded exchange relies on (reduce-scatter whose output is the rank's
    piece of its input; all-gather whose input is the rank's piece of its output) against a plain SUM all-reduce, issued THE WAY
    THE STEP ISSUES THEM (advisor, round 5): bucket-sized buffers (W pieces of ``INPLACE_CHECK_FLOATS``), the reduce-scatter as
    an async call from the current stream with a producer kernel right in front of it, the all-gather on the communication
    stream (``streams.comm_stream``) behind an event, as ``ShardedGradExchange._reduce_scatter`` / ``gather_bucket`` do.  Any
    exception -- not only RuntimeError -- and any mismatch sends every rank to the all-reduce route.  Collective: every rank
    calls it; the verdict is the MIN over ranks.  Runs once per process group and device; the result is in ``describe()``."""
    key = (dist.get_backend(), str(device))
    if key in _INPLACE_CHECK:
        return _INPLACE_CHECK[key]
    rank, W = world()
    on_gpu = torch.device(device).type == "cuda"
    c = INPLACE_CHECK_FLOATS if on_gpu else 256
    n = c * W
    base = (torch.arange(n, dtype=torch.float32, device=device) % 97) * 0.25
    mine = base * float(rank + 1) + float(rank)
    want = mine.clone()
    dist.all_reduce(want, op=dist.ReduceOp.SUM)
    a, b = rank * c, (rank + 1) * c
    ok = True
    try:
        buf = torch.empty_like(mine)
        buf.copy_(mine)                         # the producer kernel, on the current stream, right in front of the collective
        if _native is not None and buf.is_cuda:
            cur = torch.cuda.current_stream(buf.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            _native.stream.wait_event(ready)
            _native.reduce_scatter(buf, buf[a:b], _native.stream)
            done = torch.cuda.Event()
            done.record(_native.stream)
            cur.wait_event(done)
        elif dist.get_backend() == "gloo":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True).wait()
        else:
            dist.reduce_scatter_tensor(buf[a:b], buf, op=dist.ReduceOp.SUM, async_op=True).wait()
        ok = ok and bool(torch.equal(buf[a:b], want[a:b]))
        buf = torch.full_like(mine, -1.0)
        buf[a:b] = want[a:b]
        if buf.is_cuda:
            from . import streams
            cur = torch.cuda.current_stream(buf.device)
            cs = streams.comm_stream(buf.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            cs.wait_event(ready)
            with torch.cuda.stream(cs):
                if _native is not None:
                    _native.all_gather(buf[a:b], buf, cs)
                elif dist.get_backend() == "gloo":
                    dist.all_gather_into_tensor(buf, buf[a:b].clone())
                else:
                    dist.all_gather_into_tensor(buf, buf[a:b], async_op=True).wait()
                done = torch.cuda.Event()
                done.record(cs)
            cur.wait_event(done)
        elif dist.get_backend() == "gloo":
            dist.all_gather_into_tensor(buf, buf[a:b].clone())
        else:
            dist.all_gather_into_tensor(buf, buf[a:b])
        ok = ok and bool(torch.equal(buf, want))
    except Exception as e:                  # a backend that refuses aliased buffers says so here, not in the first training step
        sys.stderr.write(f"[spatial_clip_amd.comm] rank {rank}: in-place collective refused: {type(e).__name__}: {e}\n")
        ok = False
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _INPLACE_CHECK[key] = bool(flag.item() > 0.5)
    return _INPLACE_CHECK[key]


def make_grad_exchange(store, bucket_floats: int = 16 * 1024 * 1024):
    """The gradient exchange of this process group: ``GradBucketReducer`` (default: bucketed all-reduce + replicated AdamW, the
    reference's DDP shape) or, with SC_GRAD_EXCHANGE=sharded, ``ShardedGradExchange``; None without a group.  The sharded route
    is taken only (a) when the parameter store was padded for this world size -- a store built before ``comm.init_from_env``
    (a library user) is not: that warns and takes the all-reduce route instead of raising -- and (b) after its two in-place
    collectives have reproduced a plain all-reduce on this group (``inplace_collectives_verified``); otherwise every rank says so
    on stderr and takes the all-reduce route -- same weights."""
    if not is_dist():
        return None
    if grad_exchange_mode() != "sharded":
        return GradBucketReducer(store.grad, bucket_floats)
    _, W = world()
    if store.total % (64 * W):
        sys.stderr.write(f"[spatial_clip_amd.comm] the parameter store ({store.total} floats) was built before the process group and is "
                         f"not padded for {W} ranks: gradient exchange falls back to bucketed all-reduce + replicated AdamW "
                         "(build the model after comm.init_from_env() for the sharded route)\n")
        os.environ["SC_GRAD_EXCHANGE"] = "allreduce"
        return GradBucketReducer(store.grad, bucket_floats)
    if not inplace_collectives_verified(store.grad.device):
        sys.stderr.write("[spatial_clip_amd.comm] in-place reduce-scatter / all-gather did not reproduce the all-reduce on this "
                         "process group: gradient exchange falls back to bucketed all-reduce + replicated AdamW\n")
        os.environ["SC_GRAD_EXCHANGE"] = "allreduce"
        return GradBucketReducer(store.grad, bucket_floats)
    return ShardedGradExchange(store, bucket_floats)
