"""Bitwise comparison of the persistent two-pass attention backward against the one-workgroup-per-head kernel (same
arithmetic, so every element must be identical), four launches per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
for (B, L, H, causal) in [(2, 77, 8, True), (2, 197, 3, False), (40, 77, 8, True), (9, 224, 30, True), (26, 197, 12, False)]:
    dh = 64; d = H * dh
    g = torch.Generator().manual_seed(L + dh)
    qkv = torch.randn(B * L, 3 * d, generator=g).bfloat16().cuda()
    dout = torch.randn(B * L, d, generator=g).bfloat16().cuda()
    out, lse = ops.attn_fwd(qkv, B, L, H, dh, causal)
    os.environ["SC_ATTN_BWD2"] = "0"; os.environ["SC_ATTN_BWD1"] = "0"
    ref = ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, causal).clone()
    os.environ["SC_ATTN_BWD2"] = "1"   # BWD1 stays "0": the persistent kernel takes the launch
    for rep in range(4):
        got = torch.full_like(ref, 7.0)
        ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, causal, dqkv=got)
        bad = (got != ref).nonzero()
        print((B, L, H, causal), "rep", rep, "mismatches", bad.shape[0], flush=True)
        if bad.shape[0]:
            rows = bad[:, 0] % L; cols = bad[:, 1]
            for r, c in list(zip(rows.tolist(), cols.tolist()))[:24]:
                i = (bad[:, 0] % L == r).nonzero()[0]
                print("   row", r, "col", c, "tensor", c // d, "head", (c % d) // dh, "d", c % dh)
            print("   sample got", got[bad[0, 0], bad[0, 1]].item(), "ref", ref[bad[0, 0], bad[0, 1]].item())
