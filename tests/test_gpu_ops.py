"""GPU parity of the non-GEMM kernels (through the C ABI) against the CPU oracle's maths."""
import math

import os

import pytest
import torch

from oracle import spatial_clip_oracle as O

pytestmark = pytest.mark.gpu


def _ops():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import ops
    return ops


def bf(x):
    return x.to(torch.bfloat16)


def ref_attn(qkv, B, L, H, dh, causal):
    d = H * dh
    q, k, v = qkv.float().view(B, L, 3 * d).split(d, dim=-1)
    q = q.view(B, L, H, dh).transpose(1, 2)
    k = k.view(B, L, H, dh).transpose(1, 2)
    v = v.view(B, L, H, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    a = torch.softmax(s, -1)
    return (a @ v).transpose(1, 2).reshape(B * L, d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,L,H,dh,causal", [(2, 197, 3, 64, False), (3, 17, 2, 32, False), (2, 13, 2, 32, True),
                                             (2, 77, 8, 64, True), (1, 257, 2, 64, False), (4, 16, 2, 32, True),
                                             (2, 64, 1, 64, False), (1, 320, 2, 64, False), (1, 310, 1, 64, True)])
def test_attention_fwd_bwd(B, L, H, dh, causal):
    # L <= 304 (dh = 64) takes the fused backward kernel (Q, K, V, dO of a head in LDS together), longer sequences the
    # dq + dkv pair
    ops = _ops()
    g = torch.Generator().manual_seed(L + dh)
    d = H * dh
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    dout = bf(torch.randn(B * L, d, generator=g))
    x = qkv.float().requires_grad_(True)
    o_ref, lse_ref = ref_attn(x, B, L, H, dh, causal)
    o_ref.backward(dout.float())
    out, lse = ops.attn_fwd(qkv.cuda(), B, L, H, dh, causal)
    torch.testing.assert_close(out.float().cpu(), o_ref.detach(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(lse.cpu(), lse_ref.detach(), atol=2e-3, rtol=1e-3)
    dqkv = ops.attn_bwd(qkv.cuda(), out, dout.cuda(), lse, B, L, H, dh, causal)
    torch.testing.assert_close(dqkv.float().cpu(), x.grad, atol=4e-2, rtol=4e-2)


@pytest.mark.parametrize("B,L,H,causal,q_rows", [(64, 197, 12, False, 0), (110, 77, 8, True, 0), (70, 197, 12, False, 1),
                                                 (300, 33, 3, False, 0), (37, 224, 9, True, 0), (301, 33, 1, True, 0),
                                                 (130, 65, 3, True, 0), (90, 223, 3, False, 0), (280, 223, 1, True, 0),
                                                 (257, 65, 1, False, 1),
                                                 # round 5: 225..288 tokens (sc_attention_p2.hip: two query tiles per wave, one V image)
                                                 (40, 257, 16, False, 0), (300, 257, 1, False, 0), (70, 225, 4, True, 0),
                                                 (33, 288, 9, False, 0), (270, 240, 1, True, 0), (50, 257, 16, False, 1),
                                                 (20, 256, 16, False, 0), (40, 273, 7, False, 30)])
def test_attention_fwd_persistent_walks_many_heads(B, L, H, causal, q_rows):
    """More heads than CUs: every persistent workgroup walks several heads through its LDS double buffer (the DMA of
    head i+1 lands while head i computes; counted vmcnt past the previous head's stores).  Run twice: the second launch
    must reproduce the first bit for bit (no dependence on what an earlier head left in LDS)."""
    ops = _ops()
    dh = 64
    d = H * dh
    g = torch.Generator().manual_seed(B + L)
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    o_ref, lse_ref = ref_attn(qkv.float(), B, L, H, dh, causal)
    out = torch.full((B * L, d), 7.0, dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B, H, L), 7.0, device="cuda")
    qd = qkv.cuda()
    ops.attn_fwd(qd, B, L, H, dh, causal, out=out, lse=lse, q_rows=q_rows)
    out2 = torch.full_like(out, 3.0); lse2 = torch.full_like(lse, 3.0)
    ops.attn_fwd(qd, B, L, H, dh, causal, out=out2, lse=lse2, q_rows=q_rows)
    nq = q_rows if q_rows else L
    o = out.float().cpu().view(B, L, d); o2 = out2.float().cpu().view(B, L, d)
    torch.testing.assert_close(o[:, :nq], o_ref.view(B, L, d)[:, :nq], atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(lse.cpu()[:, :, :nq], lse_ref[:, :, :nq], atol=2e-3, rtol=1e-3)
    assert torch.equal(o[:, :nq], o2[:, :nq]) and torch.equal(lse.cpu()[:, :, :nq], lse2.cpu()[:, :, :nq])
    if nq < L:                                          # rows past q_rows are not written
        assert bool((o[:, nq:] == 7.0).all()) and bool((lse.cpu()[:, :, nq:] == 7.0).all())


@pytest.mark.parametrize("B,L,H,causal,q_rows", [(70, 197, 12, False, 0), (301, 33, 1, True, 0), (130, 65, 3, True, 0),
                                                 (90, 223, 3, False, 0), (70, 197, 12, False, 1),
                                                 (24, 257, 16, False, 0), (290, 257, 1, False, 0), (40, 241, 5, True, 0)])
def test_attention_fwd_persistent_vs_per_head_kernel(B, L, H, causal, q_rows, monkeypatch):
    """SC_ATTN_PERSIST is read per call: the same inputs through the persistent LDS-DMA kernel and through the
    one-workgroup-per-head kernel must agree to bf16 rounding of the output (different softmax schedule: two-pass vs
    online) -- the comparison tools/attn_fuzz.py makes over random shapes, pinned here on ragged and causal ones."""
    ops = _ops()
    dh, d = 64, H * 64
    g = torch.Generator().manual_seed(B * 3 + L)
    qd = bf(torch.randn(B * L, 3 * d, generator=g)).cuda()
    outs = []
    for persist in ("1", "0"):
        monkeypatch.setenv("SC_ATTN_PERSIST", persist)
        out = torch.full((B * L, d), 7.0, dtype=torch.bfloat16, device="cuda")
        lse = torch.full((B, H, L), 7.0, device="cuda")
        ops.attn_fwd(qd, B, L, H, dh, causal, out=out, lse=lse, q_rows=q_rows)
        outs.append((out.float().cpu().view(B, L, d), lse.cpu()))
    nq = q_rows if q_rows else L
    torch.testing.assert_close(outs[0][0][:, :nq], outs[1][0][:, :nq], atol=1.6e-2, rtol=1.6e-2)
    torch.testing.assert_close(outs[0][1][:, :, :nq], outs[1][1][:, :, :nq], atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("path", ["ring", "persistent", "single_pass", "per_head"])
@pytest.mark.parametrize("B,L,H,causal", [(26, 197, 12, False), (40, 77, 8, True), (300, 33, 1, False), (9, 224, 30, True),
                                          (3, 100, 2, False), (70, 161, 12, False), (260, 64, 3, False), (25, 20, 12, False)])
def test_attention_bwd_paths_walk_many_heads(B, L, H, causal, path, monkeypatch):
    """The four backward kernels behind sc_attn_bwd (selected per call by SC_ATTN_BWD3 / SC_ATTN_BWD1 / SC_ATTN_BWD2):
    ring (round 4, default: dS tiles in a three-block ring, dQ by one MFMA chain per block, rolling Q / dO refill;
    non-causal only, otherwise it declines and the next kernel runs), single-pass (dQ summed over the key waves in an
    fp32 LDS accumulator, fixed order; non-causal only), persistent two-pass with loader waves, one workgroup per head.
    More heads than CUs so that a persistent workgroup walks several heads; gradients against autograd and
    bit-identical across two launches (no float atomics)."""
    ops = _ops()
    monkeypatch.setenv("SC_ATTN_BWD3", "1" if path == "ring" else "0")             # tried first (round 4)
    monkeypatch.setenv("SC_ATTN_BWD1", "1" if path == "single_pass" else "0")      # tried second
    monkeypatch.setenv("SC_ATTN_BWD2", "1" if path == "persistent" else "0")       # tried third
    dh = 64
    d = H * dh
    g = torch.Generator().manual_seed(B * 7 + L)
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    dout = bf(torch.randn(B * L, d, generator=g))
    x = qkv.float().requires_grad_(True)
    o_ref, _ = ref_attn(x, B, L, H, dh, causal)
    o_ref.backward(dout.float())
    qd, gd = qkv.cuda(), dout.cuda()
    out, lse = ops.attn_fwd(qd, B, L, H, dh, causal)
    dq1 = torch.full((B * L, 3 * d), 7.0, dtype=torch.bfloat16, device="cuda")
    delta1 = torch.empty(B, H, L, device="cuda")
    ops.attn_bwd(qd, out, gd, lse, B, L, H, dh, causal, dqkv=dq1, delta=delta1)
    dq2 = torch.full_like(dq1, 3.0)
    ops.attn_bwd(qd, out, gd, lse, B, L, H, dh, causal, dqkv=dq2)
    assert torch.equal(dq1, dq2)
    torch.testing.assert_close(dq1.float().cpu(), x.grad, atol=4e-2, rtol=4e-2)
    want_delta = (dout.float() * out.float().cpu()).view(B, L, H, dh).sum(-1).permute(0, 2, 1)
    torch.testing.assert_close(delta1.cpu(), want_delta, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("B,L,H", [(24, 257, 16), (300, 257, 1), (40, 256, 3), (70, 225, 5), (33, 240, 8), (3, 257, 2),
                                   (260, 250, 1)])
def test_attention_bwd_ring8_for_225_to_257_tokens(B, L, H, monkeypatch):
    """Round 5 (sc_attention_bwd4.hip): the dS-ring backward with eight key waves and no helper wave, for ViT-L/14's 257
    tokens (the 257th key enters as rank-one terms computed by the reducers) and for 225..256 tokens.  More heads than CUs so
    that a workgroup walks several heads; against autograd, bit-identical across two launches, and against the
    one-workgroup-per-head kernel (SC_ATTN_BWD4=0) to bf16 rounding; delta = rowsum(dO O) is an output too."""
    ops = _ops()
    dh = 64
    d = H * dh
    g = torch.Generator().manual_seed(B * 5 + L)
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    dout = bf(torch.randn(B * L, d, generator=g))
    x = qkv.float().requires_grad_(True)
    o_ref, _ = ref_attn(x, B, L, H, dh, False)
    o_ref.backward(dout.float())
    qd, gd = qkv.cuda(), dout.cuda()
    out, lse = ops.attn_fwd(qd, B, L, H, dh, False)
    monkeypatch.setenv("SC_ATTN_BWD4", "1")
    dq1 = torch.full((B * L, 3 * d), 7.0, dtype=torch.bfloat16, device="cuda")
    delta1 = torch.full((B, H, L), 7.0, device="cuda")
    ops.attn_bwd(qd, out, gd, lse, B, L, H, dh, False, dqkv=dq1, delta=delta1)
    dq2 = torch.full_like(dq1, 3.0)
    ops.attn_bwd(qd, out, gd, lse, B, L, H, dh, False, dqkv=dq2)
    torch.cuda.synchronize()
    assert torch.equal(dq1, dq2)
    torch.testing.assert_close(dq1.float().cpu(), x.grad, atol=4e-2, rtol=4e-2)
    want_delta = (dout.float() * out.float().cpu()).view(B, L, H, dh).sum(-1).permute(0, 2, 1)
    torch.testing.assert_close(delta1.cpu(), want_delta, atol=2e-2, rtol=2e-2)
    monkeypatch.setenv("SC_ATTN_BWD4", "0")
    dq3 = torch.full_like(dq1, 5.0)
    ops.attn_bwd(qd, out, gd, lse, B, L, H, dh, False, dqkv=dq3)
    torch.testing.assert_close(dq1.float().cpu(), dq3.float().cpu(), atol=3e-2, rtol=3e-2)
    # the last token (row 256 at L = 257) is the stray key: its dK / dV rows on their own
    last = slice(L - 1, None, L)
    torch.testing.assert_close(dq1.float().cpu()[last], x.grad[last], atol=4e-2, rtol=4e-2)


@pytest.mark.parametrize("rows,d", [(50, 768), (197 * 2, 192), (33, 64), (7, 1024), (4500, 1024)])     # (4500 rows: more blocks than are resident at d = 1024)
def test_layernorm_fwd_bwd(rows, d):
    ops = _ops()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, d, generator=g) * 2 + 0.5
    gamma = torch.randn(d, generator=g)
    beta = torch.randn(d, generator=g)
    dy = bf(torch.randn(rows, d, generator=g))
    dres0 = torch.randn(rows, d, generator=g)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    y_ref = O.layer_norm(xr, gr, br)
    y_ref.backward(dy.float())
    dev = "cuda"
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
    mean = torch.empty(rows, device=dev)
    rstd = torch.empty(rows, device=dev)
    ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), y, mean, rstd, rows, d)
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), atol=3e-2, rtol=2e-2)
    for acc in (False, True):
        dres = dres0.clone().cuda()
        dbf = torch.empty(rows, d, dtype=torch.bfloat16, device=dev)
        dg = torch.empty(d, device=dev); db = torch.empty(d, device=dev); cs = torch.empty(d, device=dev)
        ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), dres, dbf, dg, db, cs, rows, d, accumulate=acc)
        want = xr.grad + (dres0 if acc else 0)
        torch.testing.assert_close(dres.cpu(), want, atol=2e-4, rtol=1e-4)
        torch.testing.assert_close(dbf.float().cpu(), want, atol=3e-2, rtol=2e-2)
        torch.testing.assert_close(dg.cpu(), gr.grad, atol=1e-3, rtol=1e-4)
        torch.testing.assert_close(db.cpu(), br.grad, atol=1e-3, rtol=1e-4)
        torch.testing.assert_close(cs.cpu(), want.sum(0), atol=2e-3, rtol=1e-4)
        # deferred column reductions (weight-gradient stream form): same bits as the one-call form
        dres2 = dres0.clone().cuda()
        dg2 = torch.full((d,), 9.0, device=dev); db2 = torch.full((d,), 9.0, device=dev); cs2 = torch.full((d,), 9.0, device=dev)
        ws = torch.empty(ops.layernorm_bwd_ws_floats(rows, d), device=dev)
        ops.layernorm_bwd(dy.cuda(), x.cuda(), mean, rstd, gamma.cuda(), dres2, dbf, dg2, db2, cs2, rows, d, accumulate=acc,
                          ws=ws, defer_reduce=True)
        assert float(dg2[0]) == 9.0                      # untouched until the reduce call
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.layernorm_bwd_reduce(ws, dg2, db2, cs2, rows, d)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(dres2, dres) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(cs2, cs)


@pytest.mark.parametrize("rows,d", [(50, 768), (197 * 2, 192), (33, 1024), (4500, 1024)])
def test_layernorm_bf16_rows_and_bf16_gradient_stream(rows, d):
    """sc_layernorm_fwd_x16 / _bwd_x16 / _bwd_g16: rows of the residual stream read as bf16 give the bits of the fp32 kernels
    fed with float(x) (the widening is exact); the bf16 gradient stream gout = bf16(float(gin) + LNbwd(dy)) equals the fp32
    buffer form started from float(gin); write_f32 = 0 leaves the fp32 buffer alone; the sparse form reads its class rows
    from the fp32 buffer."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + 1)
    dev = "cuda"
    x16 = bf(torch.randn(rows, d, generator=g) * 2 + 0.5).cuda()
    gamma, beta = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
    dy = bf(torch.randn(rows, d, generator=g)).cuda()
    gin = bf(torch.randn(rows, d, generator=g)).cuda()
    y_a, y_b = (torch.empty(rows, d, dtype=torch.bfloat16, device=dev) for _ in range(2))
    m_a, r_a, m_b, r_b = (torch.empty(rows, device=dev) for _ in range(4))
    ops.layernorm_fwd(x16, gamma, beta, y_a, m_a, r_a, rows, d)
    ops.layernorm_fwd(x16.float(), gamma, beta, y_b, m_b, r_b, rows, d)
    assert torch.equal(y_a, y_b) and torch.equal(m_a, m_b) and torch.equal(r_a, r_b)

    def bwd(x, **kw):
        dres = kw.pop("dres0").clone()
        dbf = torch.full((rows, d), 3.0, dtype=torch.bfloat16, device=dev)
        dg, db, cs = (torch.empty(d, device=dev) for _ in range(3))
        ops.layernorm_bwd(dy, x, m_a, r_a, gamma, dres, dbf, dg, db, cs, rows, d, **kw)
        return dres, dbf, dg, db, cs
    # fp32 buffer form on bf16 rows == on fp32 rows
    a = bwd(x16, dres0=gin.float(), accumulate=True)
    b = bwd(x16.float(), dres0=gin.float(), accumulate=True)
    assert all(torch.equal(p, q) for p, q in zip(a[:2], b[:2]))
    if rows <= 2000:
        assert all(torch.equal(p, q) for p, q in zip(a[2:], b[2:]))
    else:
        for p_, q_ in zip(a[2:], b[2:]):
            torch.testing.assert_close(p_, q_, atol=2e-3, rtol=1e-5)
    # bf16 stream (fp32 rows and bf16 rows): same outgoing gradient and column sums; the fp32 buffer only on request
    marker = torch.full((rows, d), 11.0, device=dev)
    os.environ["SC_LN_BWD_LEAN"] = "0"              # (d = 1024 on bf16 rows has a second, register-lean row body: below)

    def same_sums(ps, qs):
        # column sums: partial sums per BLOCK, and with more rows than resident blocks the instantiations (different register
        # counts, different resident grids) partition the rows differently -- same bits only while one block takes four rows
        if rows <= 2000:
            return all(torch.equal(p, q) for p, q in zip(ps, qs))
        for p, q in zip(ps, qs):
            torch.testing.assert_close(p, q, atol=2e-3, rtol=1e-5)
        return True
    try:
        for xx in (x16.float(), x16):
            c = bwd(xx, dres0=marker, accumulate=True, g16=True, g_in=gin, write_f32=False)
            assert torch.equal(c[0], marker) and torch.equal(c[1], a[1]) and same_sums(c[2:], a[2:])
            c = bwd(xx, dres0=marker, accumulate=True, g16=True, g_in=gin, write_f32=True)
            assert torch.equal(c[0], a[0]) and torch.equal(c[1], a[1])
    finally:
        os.environ.pop("SC_LN_BWD_LEAN")
    # The lean row body (default at d = 1024, SC_LN_BWD_LEAN=3 also at d = 768: packed bf16 inputs, dy * gamma and x_hat formed
    # twice) rounds its products where the other body lets the compiler contract them: same formula, differences of an fp32
    # ulp that move a bf16 result by one ulp on a few elements; column sums agree to fp32 rounding
    if d in (768, 1024):
        os.environ["SC_LN_BWD_LEAN"] = "3"
        try:
            c = bwd(x16, dres0=marker, accumulate=True, g16=True, g_in=gin, write_f32=True)
            c2 = bwd(x16, dres0=marker, accumulate=True, g16=True, g_in=gin, write_f32=True)
        finally:
            os.environ.pop("SC_LN_BWD_LEAN")
        assert all(torch.equal(p, q) for p, q in zip(c, c2))                         # run-to-run identical
        torch.testing.assert_close(c[0], a[0], atol=2e-6, rtol=2e-6)
        diff = (c[1].float() - a[1].float()).abs()
        assert float((diff > 0).float().mean()) < 0.01 and bool((diff <= 2.0 ** -7 * a[1].float().abs() + 4e-6).all())      # one bf16 ulp, or the fp32 difference itself near a cancellation
        for p, q in zip(c[2:], a[2:]):
            torch.testing.assert_close(p, q, atol=1e-4, rtol=1e-5)
    # sparse form: rows r % P == 0 take their incoming gradient from the fp32 buffer, the others start from zero
    P = 5
    sparse0 = torch.zeros(rows, d, device=dev)
    sparse0[::P] = gin.float()[::P]
    e = bwd(x16, dres0=sparse0, accumulate=-P, g16=True, write_f32=False)
    f = bwd(x16.float(), dres0=sparse0, accumulate=True)
    assert torch.equal(e[1], f[1]) and torch.equal(e[0], sparse0)


def test_attention_cls_query_only():
    """q_rows = 1 (the last ViT block feeds only the CLS token on): outputs of the other rows are not written, their dq is
    written as exact zeros, dk / dv get the CLS query's contribution only, and dout of the unused rows does not matter."""
    ops = _ops()
    B, L, H, dh = 3, 50, 2, 64
    d = H * dh
    g = torch.Generator().manual_seed(9)
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    dout = bf(torch.randn(B * L, d, generator=g))
    x = qkv.float().requires_grad_(True)
    o_ref, lse_ref = ref_attn(x, B, L, H, dh, False)
    cls = torch.arange(B) * L
    (o_ref[cls] * dout.float()[cls]).sum().backward()
    out = torch.full((B * L, d), 5.0, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, L, device="cuda")
    ops.attn_fwd(qkv.cuda(), B, L, H, dh, False, out=out, lse=lse, q_rows=1)
    torch.testing.assert_close(out.float().cpu()[cls], o_ref.detach()[cls], atol=2e-2, rtol=2e-2)
    keep = torch.ones(B * L, dtype=torch.bool); keep[cls] = False
    assert bool((out.cpu()[keep] == 5.0).all())
    dout_dev = dout.clone()
    dout_dev[keep] = 1.0e4                               # finite garbage: multiplied by exact zeros only
    out_in = out.clone(); out_in[keep.cuda()] = 0
    dout_dev[keep] = float("nan")                        # the q_rows = 1 kernel does not even read the unused rows
    dqkv = torch.full((B * L, 3 * d), 9.0, dtype=torch.bfloat16, device="cuda")     # no memset needed: every element is written
    ops.attn_bwd(qkv.cuda(), out_in, dout_dev.cuda(), lse, B, L, H, dh, False, dqkv=dqkv, q_rows=1)
    torch.testing.assert_close(dqkv.float().cpu(), x.grad, atol=4e-2, rtol=4e-2)
    assert bool((dqkv.cpu()[keep][:, :d] == 0).all())    # dq of the unconsumed queries: exact zeros


@pytest.mark.parametrize("B,L,H,dh,causal", [(5, 197, 12, 64, False), (4, 77, 8, 64, True), (3, 320, 2, 32, False), (2, 2, 1, 64, False)])
def test_attention_bwd_cls_kernel_shapes(B, L, H, dh, causal):
    """q_rows = 1 backward (rank-one dK / dV, fp32 math) over head dims, the longest sequence, causal (query 0 sees key 0
    only) and the shortest sequence, against autograd of the class-token outputs."""
    ops = _ops()
    d = H * dh
    g = torch.Generator().manual_seed(L * 3 + dh)
    qkv = bf(torch.randn(B * L, 3 * d, generator=g))
    dout = bf(torch.randn(B * L, d, generator=g))
    x = qkv.float().requires_grad_(True)
    o_ref, _ = ref_attn(x, B, L, H, dh, causal)
    cls = torch.arange(B) * L
    (o_ref[cls] * dout.float()[cls]).sum().backward()
    out, lse = ops.attn_fwd(qkv.cuda(), B, L, H, dh, causal)
    dqkv = torch.full((B * L, 3 * d), 9.0, dtype=torch.bfloat16, device="cuda")
    ops.attn_bwd(qkv.cuda(), out, dout.cuda(), lse, B, L, H, dh, causal, dqkv=dqkv, q_rows=1)
    torch.testing.assert_close(dqkv.float().cpu(), x.grad, atol=4e-2, rtol=4e-2)


def test_layernorm_strided_rows():
    """ln_post acts on the CLS rows only: row stride L*d."""
    ops = _ops()
    B, L, d = 5, 7, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * L, d, generator=g)
    gamma = torch.randn(d, generator=g); beta = torch.randn(d, generator=g)
    y = torch.empty(B, d, dtype=torch.bfloat16, device="cuda")
    mean = torch.empty(B, device="cuda"); rstd = torch.empty(B, device="cuda")
    ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), y, mean, rstd, B, d, ldx=L * d)
    ref = O.layer_norm(x.view(B, L, d)[:, 0], gamma, beta)
    torch.testing.assert_close(y.float().cpu(), ref, atol=3e-2, rtol=2e-2)


def test_colsum_l2norm_casts():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = bf(torch.randn(1000, 192, generator=g))
    out = torch.empty(192, device="cuda")
    ops.colsum_bf16(x.cuda(), 1000, 192, out)
    torch.testing.assert_close(out.cpu(), x.float().sum(0), atol=1e-3, rtol=1e-4)
    f = torch.randn(37, 512, generator=g)
    y = torch.empty(37, 512, device="cuda"); ybf = torch.empty(37, 512, dtype=torch.bfloat16, device="cuda")
    inv = torch.empty(37, device="cuda")
    ops.l2norm_fwd(f.cuda(), y, ybf, inv, 37, 512)
    fr = f.clone().requires_grad_(True)
    yr = torch.nn.functional.normalize(fr, dim=-1)
    torch.testing.assert_close(y.cpu(), yr.detach(), atol=1e-6, rtol=1e-5)
    dy = torch.randn(37, 512, generator=g)
    yr.backward(dy)
    dx = torch.empty(37, 512, dtype=torch.bfloat16, device="cuda")
    ops.l2norm_bwd(dy.cuda(), y, inv, dx, 37, 512)
    torch.testing.assert_close(dx.float().cpu(), fr.grad, atol=2e-3, rtol=2e-2)
    w = torch.randn(70, 100, generator=g)
    dst = torch.full((70, 128), 5.0, dtype=torch.bfloat16, device="cuda")
    ops.cast_pad_bf16(w.cuda(), dst, 70, 100, 128)
    assert torch.equal(dst[:, :100].cpu(), bf(w)) and float(dst[:, 100:].abs().max()) == 0.0
    dt = torch.empty(100, 70, dtype=torch.bfloat16, device="cuda")
    ops.cast_transpose_bf16(w.cuda(), dt, 70, 100)
    assert torch.equal(dt.cpu(), bf(w).t())


@pytest.mark.parametrize("P,S,d", [(8, 32, 64), (16, 224, 192), (14, 224, 64), (16, 64, 768), (8, 32, 1024), (16, 48, 1280)])
def test_im2col_embed(P, S, d):
    ops = _ops()
    B = 3
    g = torch.Generator().manual_seed(P)
    img = torch.randn(B, 3, S, S, generator=g)
    G_ = S // P
    L = G_ * G_ + 1
    kp = 3 * P * P
    kpad = (kp + 63) // 64 * 64
    patches = torch.zeros(B * G_ * G_, kpad, dtype=torch.bfloat16, device="cuda")
    ops.im2col(img.cuda(), patches, P)
    assert torch.equal(patches[:, :kp].cpu(), bf(O.patchify(img, P).reshape(-1, kp)))
    # embed + ln_pre fwd/bwd
    patch_out = torch.randn(B * (L - 1), d, generator=g)
    cls = torch.randn(d, generator=g); pos = torch.randn(L, d, generator=g)
    gamma = torch.randn(d, generator=g); beta = torch.randn(d, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (patch_out, cls, pos, gamma, beta)]
    po, c_, p_, g_, b_ = leaves
    tok = torch.cat([c_.view(1, 1, d).expand(B, 1, d), po.view(B, L - 1, d)], 1) + p_
    xref = O.layer_norm(tok, g_, b_)
    dx = torch.randn(B * L, d, generator=g)
    xref.backward(dx.view(B, L, d))
    x = torch.empty(B * L, d, device="cuda"); mean = torch.empty(B * L, device="cuda"); rstd = torch.empty(B * L, device="cuda")
    ops.embed_ln_fwd(patch_out.cuda(), cls.cuda(), pos.cuda(), gamma.cuda(), beta.cuda(), x, mean, rstd, B, L, d)
    torch.testing.assert_close(x.cpu(), xref.detach().view(B * L, d), atol=1e-5, rtol=1e-5)
    dres = dx.clone().cuda()
    dpatch = torch.empty(B * (L - 1), d, dtype=torch.bfloat16, device="cuda")
    dg = torch.empty(d, device="cuda"); db = torch.empty(d, device="cuda")
    dpos = torch.empty(L, d, device="cuda"); dcls = torch.empty(d, device="cuda")
    ops.embed_ln_bwd(dres, patch_out.cuda(), cls.cuda(), pos.cuda(), mean, rstd, gamma.cuda(), dpatch, dg, db, dpos, dcls, B, L, d)
    torch.testing.assert_close(dpatch.float().cpu(), po.grad, atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(dg.cpu(), g_.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(db.cpu(), b_.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dpos.cpu(), p_.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dcls.cpu(), c_.grad, atol=1e-4, rtol=1e-4)


def test_sgemm_all_layouts():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 70, 130, 45
    a = torch.randn(M, K, generator=g); b = torch.randn(N, K, generator=g)
    c = torch.zeros(M, N, device="cuda")
    ops.sgemm(a.cuda(), K, 1, b.cuda(), K, 1, c, N, M, N, K)
    torch.testing.assert_close(c.cpu(), a @ b.t(), atol=1e-4, rtol=1e-4)
    at = a.t().contiguous(); btt = b.t().contiguous()
    ops.sgemm(at.cuda(), 1, M, btt.cuda(), 1, N, c, N, M, N, K, accumulate=True)
    torch.testing.assert_close(c.cpu(), 2 * (a @ b.t()), atol=2e-4, rtol=1e-4)


def test_adamw_and_gradnorm():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    n = 4 * 1000 + 4
    p = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) * 3 for _ in range(3)]
    pr = p.clone(); m = torch.zeros(n); v = torch.zeros(n)
    pd, md, vd = p.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    nc = torch.empty(2, device="cuda")
    W = 2
    for step, gr in enumerate(grads, 1):
        gavg = gr / W
        norm = O.clip_grad_norm([gavg], 1.0)
        O.adamw_step(pr, gavg, m, v, step, 1e-3)
        ops.grad_norm(gr.cuda(), n, 1.0 / W, 1.0, nc)
        assert abs(float(nc[0]) - float(norm)) < 1e-3 * float(norm)
        pbf = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        ops.adamw_step(pd, gr.cuda(), md, vd, n, 1e-3, 0.9, 0.98, 1e-6, 0.1, step, 1.0 / W, nc, pbf)
        torch.testing.assert_close(pd.cpu(), pr, atol=2e-6, rtol=1e-5)
        assert torch.equal(pbf.cpu(), pd.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (6, 6, 32), (64, 64, 32), (65, 63, 33), (256, 2048, 512), (128, 1100, 96),
                                   (2048, 512, 256), (300, 260, 1027)])
def test_sgemm_mfma_every_layout_is_exact_fp32(M, N, K):
    """The exact-fp32 MFMA head GEMM against float64 on small-integer-free random data: every operand layout
    (k-contiguous / m- or n-contiguous), ragged edges, accumulate, 64- and 128-tiles (the latter via a large group)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    ref = (a.double() @ b.double().t())
    tol = 2e-6 * K ** 0.5 * 4
    ad, bd = a.cuda(), b.cuda()
    at, bt = a.t().contiguous().cuda(), b.t().contiguous().cuda()
    for (A, sam, sak) in ((ad, K, 1), (at, 1, M)):
        for (B_, sbn, sbk) in ((bd, K, 1), (bt, 1, N)):
            c = torch.full((M, N), float("nan"), device="cuda")
            ops.sgemm(A, sam, sak, B_, sbn, sbk, c, N, M, N, K)
            err = (c.double().cpu() - ref).abs().max().item()
            assert err <= tol, (sam, sak, sbn, sbk, err)
    # strided output + accumulate, operands with padded row strides (as the gathered features have)
    pad = torch.zeros(M, K + 4, device="cuda"); pad[:, :K] = ad
    c2 = torch.ones(M, N + 8, device="cuda")
    ops.sgemm(pad, K + 4, 1, bd, K, 1, c2, N + 8, M, N, K, accumulate=True)
    assert (c2[:, :N].double().cpu() - (ref + 1)).abs().max().item() <= tol
    assert torch.equal(c2[:, N:], torch.ones(M, 8, device="cuda"))


def test_sgemm_grouped_matches_single_launches_bitwise():
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, G, D = 48, 200, 64
    f = torch.randn(B, D, generator=g).cuda(); a = torch.randn(G, D, generator=g).cuda()
    dz = torch.randn(B, G, generator=g).cuda()
    z1, z2 = torch.empty(B, G, device="cuda"), torch.empty(B, G, device="cuda")
    d1, d2 = torch.empty(B, D, device="cuda"), torch.empty(G, D, device="cuda")
    ops.sgemm_grouped([(f, D, 1, a, D, 1, z1, G, B, G, D), (dz, G, 1, a, 1, D, d1, D, B, D, G),
                       (dz, 1, G, f, 1, D, d2, D, G, D, B), (f, D, 1, a, D, 1, z2, G, B, G, D)])
    s1, s2, s3 = torch.empty_like(z1), torch.empty_like(d1), torch.empty_like(d2)
    ops.sgemm(f, D, 1, a, D, 1, s1, G, B, G, D)
    ops.sgemm(dz, G, 1, a, 1, D, s2, D, B, D, G)
    ops.sgemm(dz, 1, G, f, 1, D, s3, D, G, D, B)
    assert torch.equal(z1, s1) and torch.equal(z2, s1) and torch.equal(d1, s2) and torch.equal(d2, s3)
    torch.testing.assert_close(z1.cpu(), (f @ a.t()).cpu(), atol=1e-4, rtol=1e-4)
    # 128x128 tiles: a group whose problems are all large
    M = N = 4096
    x = torch.randn(M, 96, generator=g).cuda(); y = torch.randn(N, 96, generator=g).cuda()
    c = torch.empty(M, N, device="cuda")
    ops.sgemm(x, 96, 1, y, 96, 1, c, N, M, N, 96)
    assert (c.double() - x.double() @ y.double().t()).abs().max().item() < 1e-4


def test_pack_rows_round_trip():
    ops = _ops()
    B, D = 37, 64
    f = torch.randn(B, D, device="cuda")
    ia = torch.arange(B, device="cuda", dtype=torch.int64) + (1 << 41)
    ib = -ia
    out = torch.zeros(B, D + 4, device="cuda")
    ops.pack_rows(f, ia, ib, out)
    assert torch.equal(out[:, :D], f)
    assert torch.equal(out[:, D:D + 2].contiguous().view(torch.int64).view(-1), ia)
    assert torch.equal(out[:, D + 2:].contiguous().view(torch.int64).view(-1), ib)
    out2 = torch.zeros(B, D, device="cuda")
    ops.pack_rows(f, None, None, out2)
    assert torch.equal(out2, f)


def test_gelu_bf16_matches_the_gemm_epilogue_bitwise():
    """sc_gelu_bf16 rebuilds the GELU output from the saved pre-activation (activation recomputation): it must be the
    GELU-pair GEMM epilogue's second output bit for bit, and the exact-erf GELU within bf16 rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    M, N, K = 300, 512, 128
    a = bf(torch.randn(M, K, generator=g)).cuda()
    w = bf(torch.randn(N, K, generator=g) * 0.3).cuda()
    bias = torch.randn(N, generator=g).cuda()
    u = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); h = torch.empty_like(u)
    ops.gemm(ops.NT, ops.EPI_GELU_PAIR, a, w, u, M=M, N=N, K=K, bias=bias, out2=h)
    h2 = ops.gelu_bf16(u, torch.full_like(u, 3.0))
    assert torch.equal(h, h2)
    ref = torch.nn.functional.gelu(u.float().cpu())
    torch.testing.assert_close(h2.float().cpu(), ref, atol=2e-2, rtol=1e-2)
    with pytest.raises(Exception):
        ops.gelu_bf16(u[:, :7].contiguous(), h[:, :7].contiguous())          # element count not a multiple of 8


def test_cast_transpose_batched_from_master_and_from_mirror():
    """All transposed bf16 weight copies in one launch: from the fp32 master and from the bf16 mirror (16-byte accesses
    on interior 64x64 tiles, element-wise on edge tiles) -- both equal bf16(master)^T exactly."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    shapes = [(128, 192), (100, 70), (512, 200), (64, 64), (72, 136)]
    total = sum(r * c for r, c in shapes)
    master = torch.randn(total, generator=g).cuda()
    mirror = master.to(torch.bfloat16)
    for use_mirror in (False, True):
        outs, desc, prefix, off, tiles = [], [], [0], 0, 0
        for r, c in shapes:
            o = torch.full((c, r + 8), 5.0, dtype=torch.bfloat16, device="cuda")       # ld_dst = r + 8: padded rows
            outs.append(o)
            desc.append([off, o.data_ptr(), r, c, r + 8])
            tiles += ((r + 63) // 64) * ((c + 63) // 64)
            prefix.append(tiles)
            off += r * c
        d = torch.tensor(desc, dtype=torch.int64, device="cuda")
        p = torch.tensor(prefix, dtype=torch.int32, device="cuda")
        ops.cast_transpose_batched(master, d, p, len(shapes), tiles, mirror_bf16=mirror if use_mirror else None)
        off = 0
        for (r, c), o in zip(shapes, outs):
            want = master[off:off + r * c].view(r, c).to(torch.bfloat16).t()
            assert torch.equal(o[:, :r].cpu(), want.cpu()), (use_mirror, r, c)
            assert bool((o[:, r:] == 5.0).all())
            off += r * c


@pytest.mark.parametrize("B,L,d,V", [(64, 77, 512, 49408), (5, 16, 64, 97), (256, 77, 512, 49408), (512, 77, 64, 1000)])      # (the last: two LDS segments)
def test_token_embedding_backward_is_deterministic_and_matches_index_add(B, L, d, V):
    """sc_token_embed_bwd_det (round 6): the scatter-add of the embedding gather without float atomics -- equal to an fp64
    index_add within fp32 rounding, equal to the atomic kernel within rounding, BIT-identical from launch to launch, rows of
    absent tokens exactly zero, positions behind the pooled one skipped (they carry zeros in the causal tower)."""
    ops = _ops()
    from spatial_clip_amd import data
    tokens = data.synthetic_captions(B, L, V, seed=3).cuda()
    eot = torch.empty(B, dtype=torch.int32, device="cuda")
    ops.argmax_rows(tokens, eot, B, L)
    g = torch.Generator(device="cuda").manual_seed(1)
    dres = torch.randn(B * L, d, device="cuda", generator=g)
    live = (torch.arange(L, device="cuda").view(1, L) <= eot.view(B, 1).long()).view(B * L, 1)
    dres = dres * live                                   # what the causal tower hands over: zero rows behind the pooled position
    ref = torch.zeros(V, d, dtype=torch.float64, device="cuda")
    ref.index_add_(0, tokens.view(-1), dres.double())
    outs = []
    for rep in range(3):
        dt = torch.full((V, d), 7.0, device="cuda")
        dp = torch.empty(L, d, device="cuda")
        ops.token_embed_bwd(tokens, dres, dt, dp, B, L, d, V, eot=eot, deterministic=True)
        outs.append(dt)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    scale = float(ref.abs().max())
    assert float((outs[0].double() - ref).abs().max()) <= 2e-6 * max(scale, 1.0) * 16          # <= 256 fp32 additions per element
    used = torch.zeros(V, dtype=torch.bool, device="cuda")
    used[tokens.view(-1)] = True
    if bool((~used).any()):
        assert float(outs[0][~used].abs().max()) == 0.0
    torch.testing.assert_close(dp, dres.view(B, L, d).sum(0), rtol=1e-4, atol=1e-4)
    da = torch.empty(V, d, device="cuda")
    ops.token_embed_bwd(tokens, dres, da, dp, B, L, d, V, deterministic=False)
    # (the atomic kernel's own summation order changes from run to run: up to 256 fp32 additions of O(1) values per element)
    torch.testing.assert_close(da, outs[0], rtol=1e-4, atol=2e-4)
    # without the pooled positions every row is visited: same result (the skipped rows were zeros)
    dn = torch.empty(V, d, device="cuda")
    ops.token_embed_bwd(tokens, dres, dn, dp, B, L, d, V, eot=None, deterministic=True)
    assert torch.equal(dn, outs[0])


@pytest.mark.parametrize("B,L", [(1, 1), (5, 16), (64, 77), (33, 130), (256, 77)])
def test_argmax_rows_first_maximum_like_torch(B, L):
    """EOT pooling position = text.argmax(-1) (src/open_clip/transformer.py:931-934): first maximum, rows with repeated maxima,
    maxima at either end, more tokens than lanes."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + L)
    tokens = torch.randint(0, 50, (B, L), generator=g, dtype=torch.int64)       # small range: many ties
    if L > 1:
        tokens[0, -1] = 10_000
        tokens[min(1, B - 1), 0] = 20_000
    out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    ops.argmax_rows(tokens.cuda(), out, B, L)
    assert torch.equal(out.cpu().long(), tokens.argmax(dim=-1))
