"""world_size-2 gloo tests of the data-parallel exchange layer (spatial-clip_amd/comm.py) on CPU tensors.

The per-rank maths is supplied by the oracle; what is under test is the product's collective plumbing: the packed
feature/id all-gather (rank-major order, int64 ids through the float payload), the reduce-scatter that is the
autograd of the gather, and the bucketed gradient all-reduce.  Expected values are the reference's own 2-rank gloo
run (tests/golden/loss_w2.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm
    from oracle import spatial_clip_oracle as O
    z = np.load(os.path.join(GOLDEN, "loss_w2.npz"))
    z = {k: torch.from_numpy(z[k]) for k in z.files}
    G = z["img"].shape[0]
    B = G // world
    sl = slice(rank * B, (rank + 1) * B)
    out = {}
    for which in ("spatial", "clip"):
        img = z["img"][sl].clone().requires_grad_(True)
        txt = z["txt"][sl].clone().requires_grad_(True)
        s = torch.tensor(float(z["scale"]), requires_grad=True)
        ids = z["ids"][sl].clone()
        all_i, all_t, ids_i, ids_t = comm.gather_packed(img.detach(), txt.detach(), ids, ids)
        assert torch.equal(ids_i, z["ids"]) and torch.equal(ids_t, z["ids"])
        assert torch.equal(all_i, z["img"]) and torch.equal(all_t, z["txt"])
        ai = all_i.clone().requires_grad_(True)
        at = all_t.clone().requires_grad_(True)
        if which == "spatial":
            loss = O.spatial_loss(img, txt, s, ids, ids, z["nb"][sl], z["alpha"][sl], ai, at, ids_i, ids_t, rank=rank)
        else:
            loss = O.clip_loss(img, txt, s, ai, at, rank=rank)
        loss.backward()
        both = comm.reduce_scatter_sum(torch.cat([ai.grad, at.grad], dim=1))
        D = img.shape[1]
        out[f"{which}_loss"] = float(loss)
        out[f"{which}_gimg"] = (img.grad + both[:, :D]).numpy()
        out[f"{which}_gtxt"] = (txt.grad + both[:, D:]).numpy()
    # bucketed gradient all-reduce: ranges announced back to front, with a hole left to finish()
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = comm.GradBucketReducer(flat, bucket_floats=256)
    for lo, hi in ((900, 1000), (700, 900), (300, 700), (100, 300)):
        red.bucket_ready(lo, hi)
    red.finish()
    out["flat"] = flat.numpy()
    # the tail of the exchange (offsets below one bucket: what backward announces last) travels in quarter-size buckets
    flat2 = torch.ones(4096, dtype=torch.float32)
    red2 = comm.GradBucketReducer(flat2, bucket_floats=1024)
    sizes = []
    for lo in range(4096 - 128, -1, -128):
        red2.bucket_ready(lo, lo + 128)
        sizes = [hi - lo_ for lo_, hi in red2.launched]
    assert sizes[:3] == [1024, 1024, 1024] and all(sz == 256 for sz in sizes[3:]) and len(sizes) == 3 + 4, sizes
    red2.finish()
    assert torch.equal(flat2, torch.full((4096,), float(world)))
    ret[rank] = out
    dist.destroy_process_group()


def test_two_rank_exchange_matches_reference_run():
    world = 2
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29611, ret), nprocs=world, join=True)
        res = dict(ret)
    z = np.load(os.path.join(GOLDEN, "loss_w2.npz"))
    for r in range(world):
        for which in ("spatial", "clip"):
            assert abs(res[r][f"{which}_loss"] - float(z[f"r{r}_{which}_loss"])) < 2e-6
            np.testing.assert_allclose(res[r][f"{which}_gimg"], z[f"r{r}_{which}_gimg"], atol=1e-6)
            np.testing.assert_allclose(res[r][f"{which}_gtxt"], z[f"r{r}_{which}_gtxt"], atol=1e-6)
        np.testing.assert_allclose(res[r]["flat"], np.arange(1000, dtype=np.float32) * 3)


def test_single_process_is_identity():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm
    a, b = torch.randn(4, 8), torch.randn(4, 8)
    ids = torch.arange(4)
    ai, at, ii, it = comm.gather_packed(a, b, ids, ids)
    assert ai is a and at is b and ii is ids
    assert comm.reduce_scatter_sum(a) is a
    assert comm.world() == (0, 1)


def _fg_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world)})
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm
    assert comm.init_from_env() == (rank, rank, world)          # gloo on a CPU-only host
    assert comm.init_from_env(expect_world=world) == (rank, rank, world)   # idempotent
    try:
        comm.init_from_env(expect_world=world + 1)
        raise AssertionError("a world-size mismatch must raise")
    except RuntimeError as e:
        assert "WORLD_SIZE" in str(e)
    g = torch.Generator().manual_seed(100 + rank)
    B, D = 5, 8
    img, txt = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    ids = torch.arange(B) + 1000 * rank + (1 << 40)               # ids beyond 32 bits survive the float payload
    ref = comm.gather_packed(img, txt, ids, ids + 7)
    fg = comm.FeatureGather(torch.device("cpu"))
    fg.begin(ids, ids + 7)
    fg.put("text", txt)
    fg.put("image", img)
    assert fg.has("text", txt) and fg.has("image", img) and fg.with_ids("text") and not fg.with_ids("image")
    all_t, ids_i, ids_t = fg.take("text")
    all_i, none_a, none_b = fg.take("image")
    assert none_a is None and none_b is None
    assert torch.equal(all_i, ref[0]) and torch.equal(all_t, ref[1])
    assert torch.equal(ids_i, ref[2]) and torch.equal(ids_t, ref[3])
    assert all_t.stride(0) == D + 4 and all_i.stride(0) == D      # read in place from the receive buffers
    fg.begin(None, None)                                           # ClipLoss: no ids travel
    fg.put("text", txt)
    assert not fg.with_ids("text") and not fg.has("image", img)
    assert torch.equal(fg.take("text")[0], ref[1])
    ret[rank] = fg.launched
    comm.shutdown()


@pytest.mark.parametrize("world", [2, 4])
def test_feature_gather_matches_synchronous_gather(world):
    """Overlapped gather == synchronous packed gather on 2 and on 4 gloo ranks (rank-major row order, 64-bit ids)."""
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_fg_worker, args=(world, 29621 + world, ret), nprocs=world, join=True)
        assert dict(ret) == {r: 3 for r in range(world)}


# ---------------------------------------------------------------------------- logged scalars / callbacks at W = 2
class _StubNet:
    precision = "bf16"

    def state_dict(self):
        return {"w": torch.zeros(2)}

    def fp8_scaling_state(self):
        return None


class _StubModel:
    net = _StubNet()
    synced = {"val/loss"}


def _scalar_worker(rank, world, port, tmp, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), SC_DIST_BACKEND="gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm
    from spatial_clip_amd.trainer import Trainer
    out = {"mean": comm.all_reduce_mean_scalars([1.0 + rank, 10.0 * (rank + 1)]),
           "flag": [comm.broadcast_flag(rank == 0), comm.broadcast_flag(rank == 1)]}
    tr = Trainer(devices=world, strategy="ddp", enable_checkpointing=True, default_root_dir=tmp,
                 callbacks={"early_stopping": {"monitor": "val/loss", "mode": "min", "patience": 1}})
    model = _StubModel()
    # sync_dist: the rank-local validation losses 1.0 / 3.0 become the group mean 2.0 on both ranks
    out["synced"] = tr._synced(model, "val/loss", 1.0 + 2.0 * rank)
    out["unsynced"] = tr._synced(model, "val/other", 1.0 + 2.0 * rank)
    # rank-local monitor values that DISAGREE (rank 1 keeps improving, rank 0 does not): rank 0's decision wins everywhere
    stops = []
    for epoch, v in enumerate([(1.0, 1.0), (2.0, 0.5), (3.0, 0.2)]):
        stops.append(tr._early_stop({"epoch": epoch, "val/loss": v[rank]}))
    out["stops"] = stops
    # checkpoint bookkeeping runs on every rank; only rank 0 writes.  The monitored scores DISAGREE between the ranks (a
    # rank-local monitor): rank 0's decision AND score are what every rank must end up with (0.3 at epoch 1; rank 1's own
    # 0.9 at epoch 2 must not make it the best there, nor leave 0.9 behind as the score later epochs compare against)
    for epoch, score in enumerate([(0.1, 0.05), (0.3, 0.1), (0.2, 0.9)]):
        tr._checkpoint_epoch(model, None, None, {"epoch": epoch, "val/R@1": score[rank]})
    cb = tr.checkpoint_callback
    out["best"], out["last"], out["score"] = cb.best_model_path, cb.last_model_path, cb.best_model_score
    out["files"] = sorted(os.listdir(cb.dirpath))
    ret[rank] = out
    dist.destroy_process_group()


def test_sync_dist_early_stop_and_best_checkpoint_agree_on_every_rank(tmp_path):
    world, port = 2, 29000 + (os.getpid() * 7 + 3) % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_scalar_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
    assert r0["mean"] == r1["mean"] == [1.5, 15.0]
    assert r0["flag"] == r1["flag"] == [True, False]
    assert r0["synced"] == r1["synced"] == 2.0 and (r0["unsynced"], r1["unsynced"]) == (1.0, 3.0)
    assert r0["stops"] == r1["stops"] == [False, True, True]
    assert r0["best"] == r1["best"] and r0["best"].endswith("epoch_001.ckpt") and r0["score"] == r1["score"] == 0.3
    assert r0["last"] == r1["last"] and r0["last"].endswith("last.ckpt")
    assert r0["files"] == ["epoch_001.ckpt", "last.ckpt"]


# ---------------------------------------------------------------------------------------------- sharded gradient exchange
class _FakeSpec:
    def __init__(self, name, offset, numel):
        self.name, self.offset, self.numel = name, offset, numel


class _FakeStore:
    """The attributes comm.ShardedGradExchange reads of a ParamStore, on CPU tensors (the real one needs the HIP kernels):
    flat gradient / master buffers padded to 64 W, the spec list (which part belongs to the second tower), no derived copies."""

    def __init__(self, total, world, second_at):
        self.total = (total + 64 * world - 1) // (64 * world) * (64 * world)
        self.grad = torch.zeros(self.total)
        self.master = torch.zeros(self.total)
        self.specs = [_FakeSpec("visual.w", 0, second_at), _FakeSpec("gene.w", second_at, total - second_at)]
        self.by_name = {s.name: s for s in self.specs}
        self.copies = {}
        self.pending = []
        self.refreshed = []

    def _transpose_plan(self, copies):
        return None

    def refresh_range(self, lo, hi, copies, plan, fresh=None):
        assert fresh is not None and lo <= fresh[0] < fresh[1] <= hi
        self.refreshed.append((lo, hi))


def _sharded_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import comm
    total = 5000
    st = _FakeStore(total, world, second_at=3500)
    g = torch.Generator().manual_seed(7)
    base = torch.randn(st.total, generator=g)
    st.grad.copy_(base * (rank + 1))                         # sum over ranks = base * W (W + 1) / 2
    ex = comm.ShardedGradExchange(st, bucket_floats=1024)
    assert all((hi - lo) % (64 * world) == 0 for lo, hi in ex.buckets)
    assert ex.buckets[0][0] == 0 and ex.buckets[-1][1] == st.total
    assert all(ex.buckets[i][1] == ex.buckets[i + 1][0] for i in range(len(ex.buckets) - 1))
    k0 = next(k for k, (lo, hi) in enumerate(ex.buckets) if lo <= 3500 < hi)
    assert ex.order[0] == k0 and sorted(ex.order) == list(range(len(ex.buckets)))     # the second tower's bucket travels first
    # backward announces ranges back to front, in pieces that do not line up with the buckets, and forgets one
    for lo, hi in ((3500, st.total), (2100, 3500), (900, 2100), (64, 900)):
        ex.bucket_ready(lo, hi)
    early = ex.rs_launched
    assert 0 < early < len(ex.buckets)                       # complete buckets left during "backward", the first one could not
    ex.finish()
    assert ex.rs_launched == len(ex.buckets)
    want = base * (world * (world + 1) / 2)
    own = torch.zeros(st.total, dtype=torch.bool)
    for k in range(len(ex.buckets)):
        a, b = ex.piece(k)
        own[a:b] = True
        assert torch.allclose(st.grad[a:b], want[a:b], rtol=1e-6, atol=1e-6), k
    assert int(own.sum()) == st.total // world == ex.shard_floats()
    # "optimiser": every rank updates ITS pieces only, then the buckets are gathered in forward order
    st.master.fill_(-1.0)
    for k in ex.order:
        a, b = ex.piece(k)
        st.master[a:b] = 2.0 * st.grad[a:b] + 1.0
        ex.gather_bucket(k)
    assert torch.allclose(st.master, 2.0 * want + 1.0, rtol=1e-6, atol=1e-6)
    assert [r for r in st.refreshed] == [ex.buckets[k] for k in ex.order]
    # optimiser-state round trip through the full flat layout (checkpoints)
    shard = torch.arange(ex.shard_floats(), dtype=torch.float32) + 10000 * rank
    full = ex.gather_moments(shard)
    back = torch.zeros_like(shard)
    ex.scatter_moments(full, back)
    assert torch.equal(back, shard)
    # start-up check of the in-place collectives (what make_grad_exchange runs before it takes the sharded route), and the
    # route a group takes when the check fails: bucketed all-reduce, on every rank
    assert comm.inplace_collectives_verified(torch.device("cpu")) is True
    os.environ.pop("SC_GRAD_EXCHANGE", None)        # default route: the reference's DDP shape
    assert isinstance(comm.make_grad_exchange(st, bucket_floats=1024), comm.GradBucketReducer)
    os.environ["SC_GRAD_EXCHANGE"] = "sharded"
    assert isinstance(comm.make_grad_exchange(st, bucket_floats=1024), comm.ShardedGradExchange)
    assert comm.describe()["inplace_collectives_verified"] is True
    for key in list(comm._INPLACE_CHECK):
        comm._INPLACE_CHECK[key] = False
    assert isinstance(comm.make_grad_exchange(st, bucket_floats=1024), comm.GradBucketReducer)
    assert comm.grad_exchange_mode() == "allreduce"
    ret[rank] = {"master": st.master.numpy().copy(), "full": full.numpy().copy(), "stats": dict(comm.STATS)}
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_gradient_exchange_on_gloo(world):
    """per-bucket reduce-scatter -> update of the rank's pieces -> per-bucket all-gather, on 2 and 4 gloo
    ranks: every rank ends with the same, fully updated buffer, the same as a SUM all-reduce + replicated update gives."""
    mp.set_start_method("spawn", force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_worker, args=(world, 29650 + world, ret), nprocs=world, join=True)
        res = dict(ret)
    for r in range(1, world):
        np.testing.assert_array_equal(res[r]["master"], res[0]["master"])
        np.testing.assert_array_equal(res[r]["full"], res[0]["full"])
    assert res[0]["stats"]["reduce_scatter(gradient bucket)"][0] == res[0]["stats"]["all_gather(weights bucket)"][0]
