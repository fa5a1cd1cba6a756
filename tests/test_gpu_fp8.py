"""FP8 (e4m3) forward GEMM path of BASELINE configs[4]: exactness of the MX-scaled MFMA kernel on data that e4m3
represents exactly, the per-row power-of-two quantiser, every forward epilogue, and the stated accuracy against the
unquantised product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import ops
    return ops


def deq(q8, sinv):
    return q8.view(torch.float8_e4m3fn).float() * sinv[:, None]


def test_quantiser_rows_power_of_two_and_round_trip():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 256, generator=g) * torch.logspace(-3, 2, 37).view(-1, 1)
    x[5] = 0.0
    for src in (x.cuda(), x.bfloat16().cuda()):
        q, sinv = ops.quantize_rows_fp8(src)
        ref = src.float()
        s = 1.0 / sinv
        assert torch.equal(torch.log2(s).round(), torch.log2(s))                 # exact powers of two
        amax = ref.abs().amax(1)
        nz = amax > 0
        assert bool(((amax * s)[nz] <= 448.0).all()) and bool(((amax * s)[nz] > 224.0).all())   # top binade of e4m3
        assert float(s[5]) == 1.0
        back = deq(q, sinv)
        # e4m3: 3 mantissa bits -> half-ulp relative error 2^-4 of the value (absolute floor from the subnormal step)
        err = (back - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -4 + (sinv * 2.0 ** -10)[:, None]).all())
        want = (ref * s[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(q, want)                                                # same bits as torch's e4m3fn rounding
    q2, s2 = ops.quantize_rows_fp8(x.cuda(), fixed_scale=4.0)
    assert torch.equal(s2, torch.full((37,), 0.25, device="cuda"))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 200, 384), (1024, 768, 768), (197 * 8, 3072, 768)])
def test_gemm_fp8_is_exact_on_representable_data(M, N, K):
    """Small integers x power-of-two row scales: every product and partial sum is exact in fp32, so the kernel must
    reproduce the float64 result bit for bit -- any wrong k-mapping / fragment pairing shows immediately."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-7, 8, (M, K), generator=g).float() * torch.pow(2.0, torch.randint(-3, 4, (M, 1), generator=g).float())
    b = torch.randint(-7, 8, (N, K), generator=g).float() * torch.pow(2.0, torch.randint(-3, 4, (N, 1), generator=g).float())
    a8, sa = ops.quantize_rows_fp8(a.cuda())
    b8, sb = ops.quantize_rows_fp8(b.cuda())
    assert torch.equal(deq(a8, sa).cpu(), a) and torch.equal(deq(b8, sb).cpu(), b)
    ref = (a.double() @ b.double().t())
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_fp8(ops.EPI_F32, a8, sa, b8, sb, out, M=M, N=N, K=K)
    assert torch.equal(out.cpu().double(), ref)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    out2 = torch.empty((M, N), device="cuda")
    ops.gemm_fp8(ops.EPI_F32_BIAS_RES, a8, sa, b8, sb, out2, M=M, N=N, K=K, bias=bias.cuda(), res=res.cuda())
    torch.testing.assert_close(out2.cpu(), (ref + bias + res).float(), atol=1e-3, rtol=1e-6)
    u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h = torch.empty_like(u)
    scale = 1.0 / 64
    ops.gemm_fp8(ops.EPI_GELU_PAIR, a8, sa * scale, b8, sb, u, M=M, N=N, K=K, bias=bias.cuda(), out2=h)
    uref = (ref * scale + bias).float()
    torch.testing.assert_close(u.float().cpu(), uref, atol=2e-2, rtol=1e-2)
    torch.testing.assert_close(h.float().cpu(), torch.nn.functional.gelu(u.float().cpu()), atol=2e-2, rtol=2e-2)
    ob = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_fp8(ops.EPI_BF16_BIAS, a8, sa * scale, b8, sb, ob, M=M, N=N, K=K, bias=bias.cuda())
    torch.testing.assert_close(ob.float().cpu(), uref, atol=2e-2, rtol=1e-2)


def test_gemm_fp8_accuracy_on_activation_like_data():
    """Stated accuracy of the recipe: against the UNQUANTISED product the relative Frobenius error of one
    [tokens, 1024] x [4096, 1024]^T projection stays below 4 % (e4m3 round-off 2^-4 per element, averaged over K), and
    against the product of the dequantised operands the kernel is exact to fp32 accumulation order."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    M, N, K = 2048, 4096, 1024
    a = torch.randn(M, K, generator=g)
    a[:, 7] *= 30.0                                                # an outlier channel, as ViT residual streams have
    w = torch.randn(N, K, generator=g) * K ** -0.5
    a8, sa = ops.quantize_rows_fp8(a.bfloat16().cuda())
    w8, sw = ops.quantize_rows_fp8(w.cuda())
    out = torch.empty((M, N), device="cuda")
    ops.gemm_fp8(ops.EPI_F32, a8, sa, w8, sw, out, M=M, N=N, K=K)
    exact = deq(a8, sa).double() @ deq(w8, sw).double().t()
    assert float((out.double() - exact).abs().max() / exact.abs().max()) < 1e-5
    full = a.bfloat16().float().cuda().double() @ w.cuda().double().t()
    rel = float((out.double() - full).norm() / full.norm())
    print(f"[fp8] relative Frobenius error vs the unquantised product: {rel:.4f}")
    assert rel < 0.04
    # timing of the projection against the bf16 kernel on the same shape
    ab, wb = a.bfloat16().cuda(), w.bfloat16().cuda()
    ob = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    o8 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")

    def t(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20
    tb = t(lambda: ops.gemm(ops.NT, ops.EPI_BF16, ab, wb, ob, M=M, N=N, K=K))
    t8 = t(lambda: ops.gemm_fp8(ops.EPI_BF16, a8, sa, w8, sw, o8, M=M, N=N, K=K))
    fl = 2.0 * M * N * K
    print(f"[fp8] {M}x{N}x{K}: bf16 {tb * 1e3:.0f} us ({fl / tb / 1e9:.0f} TFLOP/s), fp8 {t8 * 1e3:.0f} us ({fl / t8 / 1e9:.0f} TFLOP/s)")
