#!/usr/bin/env python3
"""Same-node yardstick for the kernel floors of DESIGN section 7 (round-5 verdict, item 3).  TOOL ONLY: nothing in the package
imports this file and the package never calls a vendor library; here the vendor kernels (hipBLASLt / rocBLAS behind
``torch.matmul`` / ``F.linear``, the flash-attention kernel behind ``F.scaled_dot_product_attention``) run the training step's
exact shapes on the same box, beside the in-tree kernels, with operands that are COLD in the step's sense: every timed call
takes the next of ``NBUF`` operand sets (> 256 MB Infinity Cache between two uses of the same bytes).

    python tools/vendor_yardstick.py            # ViT-B/16 shapes (M = 256 x 197) + ViT-L/14 attention shape
    REPS=3 N=12 python tools/vendor_yardstick.py

Output: one table (us per launch, TFLOP/s or TB/s algorithmic) -- six GEMM classes (three operand layouts each as PyTorch's
autograd would call them: forward ``x W^T``, data gradient ``dY W``, weight gradient ``dY^T x``) and two attention shapes.
Under ``rocprofv3 --kernel-trace --stats`` the kernel names say which library solution / attention backend ran."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401,E402
from spatial_clip_amd import ops  # noqa: E402
from spatial_clip_amd.towers import _splitk_for  # noqa: E402

dev = "cuda"
N_TIMED = int(os.environ.get("N", 12))
REPS = int(os.environ.get("REPS", 3))
NBUF = int(os.environ.get("NBUF", 6))
BF = torch.bfloat16


def timed(fns, n=N_TIMED):
    """Median over REPS of the mean time of n calls, call i using operand set i % len(fns).  us per call."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    res = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[len(res) // 2]


ROWS = []


def row(cls, what, us_vendor, us_tree, flops=None, bytes_=None, note=""):
    def rate(us):
        if us is None:
            return "      -"
        return f"{flops / us / 1e6:7.1f}" if flops else f"{bytes_ / us / 1e6:7.2f}"
    unit = "TF/s" if flops else "TB/s"
    r = "" if (us_vendor is None or us_tree is None) else f"{us_vendor / us_tree:5.2f}x"
    line = (f"{cls:22s} {what:34s} vendor {us_vendor if us_vendor is not None else float('nan'):8.1f} us {rate(us_vendor)} {unit} | "
            f"in-tree {us_tree if us_tree is not None else float('nan'):8.1f} us {rate(us_tree)} {unit} | vendor/in-tree {r} {note}")
    print(line, flush=True)
    ROWS.append(line)


def gemm_class(name, M, N_out, K_in, fwd_epi="bias"):
    """One Linear of the step: y[M, N_out] = x[M, K_in] W[N_out, K_in]^T.  Forward, data gradient and weight gradient."""
    g = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.randn(M, K_in, device=dev, generator=g).to(BF) for _ in range(NBUF)]
    dys = [torch.randn(M, N_out, device=dev, generator=g).to(BF) for _ in range(NBUF)]
    w = (torch.randn(N_out, K_in, device=dev, generator=g) * 0.03).to(BF)          # forward copy  [N_out, K_in]
    wt = w.t().contiguous()                                                        # data-gradient copy [K_in, N_out]
    bias = torch.randn(N_out, device=dev)
    bias16 = bias.to(BF)
    fl = 2.0 * M * N_out * K_in
    # ---- forward: vendor F.linear (bias fused by hipBLASLt) vs in-tree NT GEMM + bias epilogue
    y = torch.empty(M, N_out, device=dev, dtype=BF)
    v = timed([lambda x=x: F.linear(x, w, bias16) for x in xs])
    t = timed([lambda x=x: ops.gemm(ops.NT, ops.EPI_BF16_BIAS, x, w, y, M=M, N=N_out, K=K_in, bias=bias) for x in xs])
    row(name, f"fwd  x W^T + b   [{M}x{N_out}x{K_in}]", v, t, flops=fl)
    # ---- data gradient: dX = dY W   (vendor: NN on the forward weight; in-tree: NT on the transposed copy)
    dx = torch.empty(M, K_in, device=dev, dtype=BF)
    v = timed([lambda dy=dy: torch.matmul(dy, w) for dy in dys])
    t = timed([lambda dy=dy: ops.gemm(ops.NT, ops.EPI_BF16, dy, wt, dx, M=M, N=K_in, K=N_out) for dy in dys])
    row(name, f"dgrad dY W       [{M}x{K_in}x{N_out}]", v, t, flops=fl)
    # ---- weight gradient: dW = dY^T x  (fp32 result in-tree: split-K slabs + reduce, bias gradient fused; vendor: bf16 out + sum)
    dw = torch.empty(N_out, K_in, device=dev, dtype=torch.float32)
    db = torch.empty(N_out, device=dev, dtype=torch.float32)
    sk = _splitk_for(N_out, K_in, M)
    v = timed([lambda dy=dy, x=x: torch.matmul(dy.t(), x) for dy, x in zip(dys, xs)])
    v2 = timed([lambda dy=dy, x=x: (torch.matmul(dy.t(), x), dy.sum(0)) for dy, x in zip(dys, xs)])
    t = timed([lambda dy=dy, x=x: ops.gemm_wgrad_bias(dy, x, dw, db, M=N_out, N=K_in, K=M, splitk=sk) for dy, x in zip(dys, xs)])
    row(name, f"wgrad dY^T x     [{N_out}x{K_in}x{M}]", v, t, flops=fl, note=f"(vendor + bias-gradient sum: {v2:.1f} us; in-tree has it fused, fp32 out, split-K {sk})")
    del xs, dys
    torch.cuda.empty_cache()


def fused_sequences(M, d, mlp):
    """The two epilogue-heavy forward launches as the reference's op sequence on vendor kernels: c_fc + GELU (in-tree: one launch
    that also stores gelu'(u)), and a residual Linear (x + linear(a); in-tree: residual epilogue)."""
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(M, d, device=dev, generator=g).to(BF) for _ in range(NBUF)]
    w = (torch.randn(mlp, d, device=dev, generator=g) * 0.03).to(BF)
    b = torch.randn(mlp, device=dev)
    b16 = b.to(BF)
    u, h = torch.empty(M, mlp, device=dev, dtype=BF), torch.empty(M, mlp, device=dev, dtype=BF)
    epi_pair, epi_grad_pair, epi_dgelu = ops.act_epilogues(False)
    v = timed([lambda x=x: F.gelu(F.linear(x, w, b16)) for x in xs])
    t = timed([lambda x=x: ops.gemm(ops.NT, epi_grad_pair, x, w, u, M=M, N=mlp, K=d, bias=b, out2=h) for x in xs])
    row("c_fc + GELU", f"fwd gelu(x W^T + b) [{M}x{mlp}x{d}]", v, t, flops=2.0 * M * mlp * d,
        note="(vendor: linear, then F.gelu; in-tree: one launch storing h and gelu'(u))")
    hs = [torch.randn(M, mlp, device=dev, generator=g).to(BF) for _ in range(3)]
    res = [torch.randn(M, d, device=dev, generator=g).to(BF) for _ in range(3)]
    w2 = (torch.randn(d, mlp, device=dev, generator=g) * 0.03).to(BF)
    b2 = torch.randn(d, device=dev)
    b2_16 = b2.to(BF)
    xo = torch.empty(M, d, device=dev, dtype=BF)
    v = timed([lambda a=a, r=r: r + F.linear(a, w2, b2_16) for a, r in zip(hs, res)])
    t = timed([lambda a=a, r=r: ops.gemm(ops.NT, ops.EPI_BF16_BIAS_RES, a, w2, xo, M=M, N=d, K=mlp, bias=b2, res=r) for a, r in zip(hs, res)])
    row("c_proj + residual", f"fwd r + a W^T + b [{M}x{d}x{mlp}]", v, t, flops=2.0 * M * d * mlp,
        note="(bf16 residual stream)")
    # the x act' data gradient: dU = (dY W) * gelu'(u)
    dys = [torch.randn(M, d, device=dev, generator=g).to(BF) for _ in range(NBUF)]
    gf = torch.rand(M, mlp, device=dev, generator=g).to(BF)
    dU = torch.empty(M, mlp, device=dev, dtype=BF)
    w2t = w2.t().contiguous()          # [mlp, d]: in-tree data-gradient copy of c_proj.weight ([d, mlp]) is [mlp... NT operand rows = outputs
    v = timed([lambda dy=dy: torch.matmul(dy, w2) * gf for dy in dys])
    t = timed([lambda dy=dy: ops.gemm(ops.NT, ops.EPI_BF16_MUL_AUX, dy, w2t, dU, M=M, N=mlp, K=d, aux=gf) for dy in dys])
    row("c_proj dgrad x act'", f"dU = (dY W) * g  [{M}x{mlp}x{d}]", v, t, flops=2.0 * M * mlp * d)


def attention(B, H, L, dh, tag):
    g = torch.Generator(device=dev).manual_seed(2)
    d = H * dh
    nb = 3
    qkvs = [torch.randn(B * L, 3 * d, device=dev, generator=g).to(BF) for _ in range(nb)]
    douts = [torch.randn(B * L, d, device=dev, generator=g).to(BF) for _ in range(nb)]
    # vendor operands: [B, H, L, dh] contiguous (the layout the flash kernels are written for), requires_grad for the backward
    trip = []
    for t in qkvs:
        q, k, v = (t.view(B, L, 3, H, dh)[:, :, j].permute(0, 2, 1, 3).contiguous().requires_grad_(True) for j in range(3))
        trip.append((q, k, v))
    dO_v = [t.view(B, L, H, dh).permute(0, 2, 1, 3).contiguous() for t in douts]
    byt_f = (3 + 1) * B * L * d * 2
    byt_b = (3 + 3 + 2) * B * L * d * 2
    try:
        vf = timed([lambda t=t: F.scaled_dot_product_attention(*t) for t in trip])

        def fb(t, go):
            o = F.scaled_dot_product_attention(*t)
            torch.autograd.grad(o, t, go)
        vfb = timed([lambda t=t, go=go: fb(t, go) for t, go in zip(trip, dO_v)])
        vb = vfb - vf
    except Exception as e:              # no flash backend for this architecture in this build
        print(f"[attention {tag}] vendor SDPA failed: {type(e).__name__}: {e}", flush=True)
        vf = vb = None
    out = torch.empty(B * L, d, device=dev, dtype=BF)
    lse = torch.empty(B, H, L, device=dev)
    dqkv = torch.empty(B * L, 3 * d, device=dev, dtype=BF)
    delta = torch.empty(B, H, L, device=dev)
    tf = timed([lambda t=t: ops.attn_fwd(t, B, L, H, dh, False, out=out, lse=lse) for t in qkvs])
    ops.attn_fwd(qkvs[0], B, L, H, dh, False, out=out, lse=lse)
    tb = timed([lambda t=t, go=go: ops.attn_bwd(t, out, go, lse, B, L, H, dh, False, dqkv=dqkv, delta=delta) for t, go in zip(qkvs, douts)])
    row(f"attention {tag}", f"forward  B={B} H={H} L={L} dh={dh}", vf, tf, bytes_=byt_f)
    row(f"attention {tag}", "backward (vendor: fwd+bwd - fwd)", vb, tb, bytes_=byt_b)
    backend = "?"
    try:
        from torch.backends.cuda import flash_sdp_enabled, mem_efficient_sdp_enabled
        backend = f"flash_sdp_enabled={flash_sdp_enabled()} mem_efficient_sdp_enabled={mem_efficient_sdp_enabled()}"
    except Exception:
        pass
    print(f"[attention {tag}] SDPA dispatch flags: {backend}", flush=True)


def main():
    print(f"# torch {torch.__version__}, device {torch.cuda.get_device_name(0)}; N={N_TIMED} calls x REPS={REPS} (median), "
          f"{NBUF} rotating operand sets", flush=True)
    M, d, mlp = int(os.environ.get("M", 256 * 197)), 768, 3072
    which = os.environ.get("WHICH", "gemm,fused,attn")
    if "gemm" in which:
        gemm_class("qkv", M, 3 * d, d)
        gemm_class("out_proj", M, d, d)
        gemm_class("c_fc", M, mlp, d)
        gemm_class("c_proj", M, d, mlp)
        gemm_class("ViT-L/14 c_fc", 256 * 257, 4096, 1024)
        gemm_class("4096^3", 4096, 4096, 4096)
    if "fused" in which:
        fused_sequences(M, d, mlp)
    if "attn" in which:
        attention(256, 12, 197, 64, "ViT-B/16")
        attention(256, 16, 257, 64, "ViT-L/14")
    print("\n# table")
    for r in ROWS:
        print(r)


if __name__ == "__main__":
    main()
