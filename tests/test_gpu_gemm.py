"""GPU parity of the MFMA GEMM (through the C ABI) against fp32 torch on the same bf16-rounded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import ops
    return ops


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def gelu(x):
    return torch.nn.functional.gelu(x)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 256, 128), (197 * 3, 192, 192), (1000, 576, 192),
                                   (50, 64, 64), (394, 2304, 768), (300, 768, 3072), (1024, 768, 768),
                                   (197 * 8, 3072, 768), (513, 200, 128)])
def test_nt_plain_and_bias(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    ref = a.float() @ b.float().t()
    ad, bd = a.cuda(), b.cuda()
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_BF16, ad, bd, out, M=M, N=N, K=K)
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), atol=2e-2, rtol=2e-2)
    ops.gemm(ops.NT, ops.EPI_BF16_BIAS, ad, bd, out, M=M, N=N, K=K, bias=bias.cuda())
    torch.testing.assert_close(out.float().cpu(), (ref + bias).to(torch.bfloat16).float(), atol=2e-2, rtol=2e-2)
    o32 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(ops.NT, ops.EPI_F32, ad, bd, o32, M=M, N=N, K=K)
    torch.testing.assert_close(o32.cpu(), ref, atol=1e-3, rtol=1e-3)


def test_nt_asymmetric_identity():
    """A = I with an asymmetric B catches transposed / permuted accumulator layouts exactly."""
    ops = _ops()
    M = N = K = 128
    a = torch.eye(M).to(torch.bfloat16)
    b = (torch.arange(N).view(N, 1) * 3 + torch.arange(K).view(1, K) * 0.5).to(torch.bfloat16)
    o32 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(ops.NT, ops.EPI_F32, a.cuda(), b.cuda(), o32, M=M, N=N, K=K)
    assert torch.equal(o32.cpu(), a.float() @ b.float().t())


@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 448, 576, 1024])
def test_nt_phase_interleaved_ring_exact(K):
    """The 256x256 phase-interleaved kernel keeps six half-tiles in flight in an 8-slot LDS ring; small-integer
    operands make every fp32 sum exact, so a fragment read from a stale or half-landed ring slot cannot hide in a
    tolerance.  K sweeps 1..16 K tiles (prologue-only, odd/even tile counts, ring wrap), M/N ragged, repeated with
    fresh data so an intermittent ordering bug has many chances to show."""
    ops = _ops()
    M, N = 256 * 3 + 40, 256 * 2 + 24
    g = torch.Generator().manual_seed(K)
    for rep in range(6):
        a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
        b = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
        ref = a.float() @ b.float().t()
        o32 = torch.full((M, N), 5.0, dtype=torch.float32, device="cuda")
        ops.gemm(ops.NT, ops.EPI_F32, a.cuda(), b.cuda(), o32, M=M, N=N, K=K)
        assert torch.equal(o32.cpu(), ref), (K, rep)


@pytest.mark.parametrize("K", [192, 256, 448])
def test_nt_persistent_tile_walk_exact(K):
    """More than 256 tiles + a store-only bf16 epilogue selects the persistent kernel (one workgroup per CU walks a
    tile list, the DMA ring runs across tile boundaries).  Exact on small integers; M is ragged so both the counted
    (interior tile) and the draining (edge tile) post-epilogue waits are exercised; several rounds of tiles."""
    ops = _ops()
    M, N = 256 * 64 + 72, 256 * 16
    g = torch.Generator().manual_seed(K)
    for rep in range(3):
        a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
        b = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
        bias = torch.randint(-4, 5, (N,), generator=g).float()
        ref = a.cuda().float() @ b.cuda().float().t()
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(ops.NT, ops.EPI_BF16, a.cuda(), b.cuda(), out, M=M, N=N, K=K)
        assert torch.equal(out, ref.to(torch.bfloat16)), (K, rep)
        ops.gemm(ops.NT, ops.EPI_BF16_BIAS, a.cuda(), b.cuda(), out, M=M, N=N, K=K, bias=bias.cuda())
        assert torch.equal(out, (ref + bias.cuda()).to(torch.bfloat16)), (K, rep, "bias")
        h = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
        a2 = (a.float() * 0.25).to(torch.bfloat16)               # keeps u in GELU's curved range; still exact sums
        ops.gemm(ops.NT, ops.EPI_GELU_PAIR, a2.cuda(), b.cuda(), out, M=M, N=N, K=K, bias=bias.cuda(), out2=h)
        u_ref = (a2.cuda().float() @ b.cuda().float().t() + bias.cuda()).to(torch.bfloat16)
        assert torch.equal(out, u_ref), (K, rep, "gelu pair u")
        torch.testing.assert_close(h.float(), gelu(u_ref.float()).to(torch.bfloat16).float(), atol=1e-2, rtol=1e-2)


def test_gelu_pair_epilogue_by_table_equals_the_formula_for_every_bf16(monkeypatch):
    """Round 4: the forward GELU epilogue (C = gelu'(u), C2 = gelu(u), u rounded to bf16 first) reads both values from a 6400-entry
    LDS table indexed by the bf16 bits of u (2^-20 <= |u| < 32; anything else takes the formula).  The table is filled by the
    formula, so the two paths must agree bit for bit -- checked here for EVERY finite bf16 value of u (u = a . 1 exactly), with
    the table on (default) and off (SC_GELU_LUT=0), and against the closed form."""
    ops = _ops()
    bits = torch.arange(65536, dtype=torch.int32)
    vals = bits.to(torch.int16).view(torch.bfloat16)
    finite = torch.isfinite(vals.float())
    vals = torch.where(finite, vals, torch.zeros_like(vals))
    M, N, K = 65536, 256, 64
    a = torch.zeros((M, K), dtype=torch.bfloat16)
    a[:, 0] = vals
    b = torch.zeros((N, K), dtype=torch.bfloat16)
    b[:, 0] = 1.0
    bias = torch.zeros(N)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SC_GELU_LUT", sw)
        gd = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
        h = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(ops.NT, ops.EPI_GELU_GRAD_PAIR, a.cuda(), b.cuda(), gd, M=M, N=N, K=K, bias=bias.cuda(), out2=h)
        outs[sw] = (gd.view(torch.int16).cpu(), h.view(torch.int16).cpu())
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    u = vals.float()
    h = outs["1"][1].view(torch.bfloat16).float()[:, 7]
    sane = u.abs() < 1e30                              # beyond that the fp32 pieces of the formula overflow (same bits on both paths)
    torch.testing.assert_close(h[sane], gelu(u)[sane], atol=4e-3, rtol=8e-3)
    # a tile that mixes in-table and out-of-table values in one 8-element chunk / one wave: the per-chunk fallback
    monkeypatch.setenv("SC_GELU_LUT", "1")
    g = torch.Generator().manual_seed(2)
    M2 = 1024
    a2 = _rand((M2, K), g)
    a2[::37, :] = 0                                    # exact zeros (outside the table)
    a2[5::91, 0] = 300.0                               # |u| far beyond 32
    b2 = _rand((N, K), g, 0.3)
    res = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SC_GELU_LUT", sw)
        gd = torch.empty((M2, N), dtype=torch.bfloat16, device="cuda")
        h2 = torch.empty((M2, N), dtype=torch.bfloat16, device="cuda")
        ops.gemm(ops.NT, ops.EPI_GELU_GRAD_PAIR, a2.cuda(), b2.cuda(), gd, M=M2, N=N, K=K, bias=bias.cuda(), out2=h2)
        res[sw] = (gd.clone(), h2.clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])


def test_nt_phase_interleaved_splitk_exact():
    ops = _ops()
    M, N, K = 512, 512, 64 * 13
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
    b = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
    for splitk in (1, 2, 4, 13):
        o32 = torch.full((M, N), 5.0, dtype=torch.float32, device="cuda")
        ops.gemm(ops.NT, ops.EPI_F32, a.cuda(), b.cuda(), o32, M=M, N=N, K=K, splitk=splitk)
        assert torch.equal(o32.cpu(), a.float() @ b.float().t()), splitk


@pytest.mark.parametrize("M,N,K", [(333, 192, 256), (197 * 4, 768, 192), (700, 264, 64)])
def test_nt_residual_gelu_dgelu(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.float() @ b.float().t()
    o32 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, a.cuda(), b.cuda(), o32, M=M, N=N, K=K, bias=bias.cuda(), res=res.cuda())
    torch.testing.assert_close(o32.cpu(), ref + bias + res, atol=1e-3, rtol=1e-3)
    u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_GELU_PAIR, a.cuda(), b.cuda(), u, M=M, N=N, K=K, bias=bias.cuda(), out2=h)
    uref = (ref + bias).to(torch.bfloat16)
    torch.testing.assert_close(u.float().cpu(), uref.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(h.float().cpu(), gelu(u.float().cpu()).to(torch.bfloat16).float(), atol=1e-2, rtol=1e-2)
    aux = _rand((M, N), g)
    d = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_BF16_DGELU, a.cuda(), b.cuda(), d, M=M, N=N, K=K, aux=aux.cuda())
    x = aux.float().requires_grad_(True)
    gelu(x).sum().backward()
    torch.testing.assert_close(d.float().cpu(), (ref * x.grad).to(torch.bfloat16).float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("M,N,K", [(333, 192, 256), (197 * 4, 768, 192), (700, 264, 64), (256, 3072, 768),
                                   (256 * 86 + 24, 3072, 192), (64, 256, 64)])
def test_nt_gelu_grad_pair_and_mul_aux(M, N, K):
    """Round 4: the forward c_fc epilogue stores gelu'(u) (SC_EPI_GELU_GRAD_PAIR) instead of u, and the c_proj data
    gradient multiplies by that stored factor (SC_EPI_BF16_MUL_AUX).  Checked (a) against torch's erf GELU and its
    autograd derivative on the kernel's own bf16 u, (b) for bit-identity with the u-storing pair that activation
    recomputation keeps (same h; dU from the stored factor == dU from the factor recomputed out of u), over the
    128x128 kernel, the 256x256 kernel and the persistent tile walk (> 1024 tiles)."""
    ops = _ops()
    g = torch.Generator().manual_seed(11 + M)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.15)
    bias = torch.randn(N, generator=g)
    ad, bd, biasd = a.cuda(), b.cuda(), bias.cuda()
    u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_GELU_PAIR, ad, bd, u, M=M, N=N, K=K, bias=biasd, out2=h)
    gd = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    h2 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_GELU_GRAD_PAIR, ad, bd, gd, M=M, N=N, K=K, bias=biasd, out2=h2)
    assert torch.equal(h, h2)                                   # same h whichever tensor travels beside it
    x = u.float().cpu().requires_grad_(True)
    y = gelu(x)
    y.sum().backward()
    torch.testing.assert_close(h.float().cpu(), y.detach().to(torch.bfloat16).float(), atol=8e-3, rtol=8e-3)
    # the stored factor: bf16 rounding of gelu'(u) (range [-0.13, 1.13]); 1 bf16 step = 2^-8 near 1
    torch.testing.assert_close(gd.float().cpu(), x.grad.to(torch.bfloat16).float(), atol=4e-3, rtol=8e-3)
    # backward: dY [M, K2] . W [K2 -> N]  x  factor
    K2 = 128
    dy, w = _rand((M, K2), g), _rand((N, K2), g, 0.1)
    d_mul = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    d_rec = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_BF16_MUL_AUX, dy.cuda(), w.cuda(), d_mul, M=M, N=N, K=K2, aux=gd)
    ops.gemm(ops.NT, ops.EPI_BF16_DGELU, dy.cuda(), w.cuda(), d_rec, M=M, N=N, K=K2, aux=u)
    assert torch.equal(d_mul, d_rec)                            # recomputation mode reproduces the default path bit for bit
    ref = (dy.float() @ w.float().t()) * x.grad
    torch.testing.assert_close(d_mul.float().cpu(), ref.to(torch.bfloat16).float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("M,N,K", [(333, 192, 256), (197 * 4, 768, 192), (700, 264, 64), (197 * 64, 768, 768), (64, 768, 3072)])
def test_nt_bf16_residual_epilogue(M, N, K):
    """SC_EPI_BF16_BIAS_RES: x_new = bf16(A.B^T + bias + x) with the residual x read as bf16 (strided rows included, as the
    class-token-only block passes them): equal to the fp32-residual epilogue fed with float(x), rounded once."""
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g)
    res_wide = torch.randn(M, 2 * N + 8, generator=g).to(torch.bfloat16).cuda()
    res = res_wide[:, 8:8 + N]                                        # row stride 2 N + 8 elements
    o16 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.NT, ops.EPI_BF16_BIAS_RES, a.cuda(), b.cuda(), o16, M=M, N=N, K=K, bias=bias.cuda(), res=res)
    o32 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, a.cuda(), b.cuda(), o32, M=M, N=N, K=K, bias=bias.cuda(), res=res.float().contiguous())
    want = o32.to(torch.bfloat16)
    diff = (o16.float() - want.float()).abs()
    # the two epilogues add bias and residual in a different order: a result on a rounding boundary may land one bf16 step apart
    assert float((diff > 0).float().mean()) < 2e-3 and float((diff / want.float().abs().clamp_min(1.0)).max()) <= 2 ** -7
    ref = (a.float() @ b.float().t() + bias + res.float().cpu())
    torch.testing.assert_close(o16.float().cpu(), ref, atol=3e-2, rtol=1e-2)


@pytest.mark.parametrize("M,N,K,splitk", [(128, 128, 64, 1), (192, 192, 197 * 2, 1), (768, 192, 1000, 4),
                                          (576, 192, 37, 1), (2304, 768, 197 * 8, 8), (512, 20032, 8, 1),
                                          (768, 768, 2048, 8), (2304, 768, 1024, 4), (768, 3072, 640, 1),
                                          (512, 20000, 256, 1), (776, 200, 128, 2)])
def test_tn_wgrad(M, N, K, splitk):
    ops = _ops()
    g = torch.Generator().manual_seed(K)
    at, bt = _rand((K, M), g), _rand((K, N), g)
    ref = at.float().t() @ bt.float()
    o32 = torch.full((M, N), -3.0, dtype=torch.float32, device="cuda")
    ops.gemm(ops.TN, ops.EPI_F32, at.cuda(), bt.cuda(), o32, M=M, N=N, K=K, splitk=splitk)
    torch.testing.assert_close(o32.cpu(), ref, atol=2e-3 * max(1, K ** 0.5 / 8), rtol=2e-3)


@pytest.mark.parametrize("K,splitk", [(64, 1), (128, 1), (192, 1), (320, 1), (576, 1), (1024, 1), (64 * 13, 4), (64 * 9, 9)])
def test_tn_phase_interleaved_ring_exact(K, splitk):
    """TN twin of the ring test above (transposed LDS reads, fused bias-gradient column sums): exact on small integers."""
    ops = _ops()
    M, N = 256 * 2 + 40, 256 * 3 + 24
    g = torch.Generator().manual_seed(K + splitk)
    for rep in range(4):
        dy = torch.randint(-3, 4, (K, M), generator=g).to(torch.bfloat16)
        x = torch.randint(-3, 4, (K, N), generator=g).to(torch.bfloat16)
        dw = torch.full((M, N), 9.0, dtype=torch.float32, device="cuda")
        db = torch.full((M,), 9.0, dtype=torch.float32, device="cuda")
        ops.gemm_wgrad_bias(dy.cuda(), x.cuda(), dw, db, M=M, N=N, K=K, splitk=splitk)
        assert torch.equal(dw.cpu(), dy.float().t() @ x.float()), (K, splitk, rep)
        assert torch.equal(db.cpu(), dy.float().sum(0)), (K, splitk, rep)
        o32 = torch.full((M, N), -3.0, dtype=torch.float32, device="cuda")
        ops.gemm(ops.TN, ops.EPI_F32, dy.cuda(), x.cuda(), o32, M=M, N=N, K=K, splitk=splitk)
        assert torch.equal(o32.cpu(), dy.float().t() @ x.float())


def test_tn_asymmetric_exact():
    ops = _ops()
    K, M, N = 64, 128, 128
    at = torch.zeros(K, M)
    at[torch.arange(K), torch.arange(K) * 2] = 1.0   # At^T picks rows of Bt
    bt = (torch.arange(K).view(K, 1) * 2.0 + torch.arange(N).view(1, N) * 0.25).to(torch.bfloat16)
    o32 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(ops.TN, ops.EPI_F32, at.to(torch.bfloat16).cuda(), bt.cuda(), o32, M=M, N=N, K=K)
    assert torch.equal(o32.cpu(), at.t() @ bt.float())


@pytest.mark.parametrize("M,N,K,splitk", [(768, 768, 2048, 8), (3072, 768, 1024, 4), (2304, 768, 197 * 8, 2),
                                          (512, 200, 256, 1), (768, 3072, 256, 1)])
def test_wgrad_with_fused_bias_grad(M, N, K, splitk):
    ops = _ops()
    g = torch.Generator().manual_seed(K + M)
    dy, x = _rand((K, M), g), _rand((K, N), g)
    dw = torch.full((M, N), 9.0, dtype=torch.float32, device="cuda")
    db = torch.full((M,), 9.0, dtype=torch.float32, device="cuda")
    ops.gemm_wgrad_bias(dy.cuda(), x.cuda(), dw, db, M=M, N=N, K=K, splitk=splitk)
    torch.testing.assert_close(dw.cpu(), dy.float().t() @ x.float(), atol=2e-3 * max(1, K ** 0.5 / 8), rtol=2e-3)
    torch.testing.assert_close(db.cpu(), dy.float().sum(0), atol=2e-3 * max(1, K ** 0.5 / 8), rtol=1e-4)


@pytest.mark.parametrize("K,splitk,shapes", [
    (64 * 13, 4, [(768, 768, False), (2304, 768, True)]),                       # out_proj + in_proj of a ViT-B block
    (64 * 9, 3, [(768, 3072, False), (3072, 768, True)]),                       # c_proj + c_fc
    (64 * 7, 7, [(768, 1024, False), (1024, 768, True), (512, 512, False), (1536, 512, True)]),   # four at once
    (64 * 5, 1, [(512, 512, True), (256, 768, True)]),                          # split-K 1: no slabs, bias sums only
    (200, 2, [(64, 64, True), (128, 64, False)]),                               # toy shapes: per-Linear fallback
    (64 * 6, 2, [(1024, 512, True)]),                                           # a group of one
])
def test_wgrad_group_exact(K, splitk, shapes):
    """sc_gemm_wgrad_group: several weight (+ bias) gradients over one token axis in ONE launch + one slab reduction
    (round 4).  Small-integer operands make every sum exact: each problem must equal dY^T . X and the column sums of dY
    bit for bit, whatever the group's common split-K, and must not touch the other problems' outputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(K + len(shapes))
    for rep in range(2):
        probs, want = [], []
        for (M, N, bias) in shapes:
            dy = torch.randint(-3, 4, (K, M), generator=g).to(torch.bfloat16)
            x = torch.randint(-3, 4, (K, N), generator=g).to(torch.bfloat16)
            dw = torch.full((M, N), 9.0, dtype=torch.float32, device="cuda")
            db = torch.full((M,), 9.0, dtype=torch.float32, device="cuda") if bias else None
            probs.append((dy.cuda(), x.cuda(), dw, db, M, N))
            want.append((dy.float().t() @ x.float(), dy.float().sum(0)))
        ops.gemm_wgrad_group(probs, K=K, splitk=splitk)
        for (dyd, xd, dw, db, M, N), (w, b) in zip(probs, want):
            assert torch.equal(dw.cpu(), w), (K, splitk, M, N, rep)
            if db is not None:
                assert torch.equal(db.cpu(), b), (K, splitk, M, N, rep, "bias")


@pytest.mark.parametrize("epi_name,M,N,K,gc", [("plain", 256 * 9 + 40, 256 * 5, 128, 2), ("plain", 256 * 17, 256 * 12, 64, 4),
                                               ("pair", 256 * 11 + 8, 256 * 7 + 64, 192, 3), ("mul_aux", 256 * 23, 256 * 12, 64, 6),
                                               ("bias_res", 256 * 9, 256 * 3, 256, 1), ("plain", 256 * 3, 256 * 4, 64, 3)])
def test_column_group_tile_walk_is_a_permutation_of_the_tiles(monkeypatch, epi_name, M, N, K, gc):
    """Round 5: SC_GEMM_COLGROUP walks the output tiles of the non-persistent 256-tile kernel in column groups inside per-XCD
    row bands.  Every tile must still be computed exactly once: results bit-identical to the row-major walk on ragged tile
    grids (rows not a multiple of the 8 bands, columns not a multiple of the group, fewer rows than bands)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    a, b = _rand((M, K), g), _rand((N, K), g, 0.2)
    epi = {"plain": ops.EPI_BF16, "pair": ops.EPI_GELU_GRAD_PAIR, "mul_aux": ops.EPI_BF16_MUL_AUX, "bias_res": ops.EPI_BF16_BIAS_RES}[epi_name]
    kw = {}
    if epi_name in ("pair", "bias_res"):
        kw["bias"] = torch.randn(N, generator=g).cuda()
    if epi_name == "mul_aux":
        kw["aux"] = _rand((M, N), g).cuda()
    if epi_name == "bias_res":
        kw["res"] = _rand((M, N), g).cuda()
    outs = []
    for sw in (f"{epi}:0", f"{epi}:{gc}"):
        monkeypatch.setenv("SC_GEMM_COLGROUP", sw)
        # (every shape here has < 1024 tiles: the launch takes the non-persistent kernel, where the walk lives)
        o = torch.full((M, N), 9.0, dtype=torch.bfloat16, device="cuda")
        o2 = torch.full((M, N), 9.0, dtype=torch.bfloat16, device="cuda") if epi_name == "pair" else None
        ops.gemm(ops.NT, epi, a.cuda(), b.cuda(), o, M=M, N=N, K=K, out2=o2, **kw)
        outs.append((o.clone(), None if o2 is None else o2.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])
    ref = a.float() @ b.float().t()
    if epi_name == "plain":
        torch.testing.assert_close(outs[1][0].float().cpu(), ref.to(torch.bfloat16).float(), atol=3e-2, rtol=2e-2)
