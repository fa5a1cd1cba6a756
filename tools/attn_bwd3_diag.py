"""Where the ring attention backward (sc_attention_bwd3.hip) differs from the one-workgroup-per-head kernel: per tensor / head / row block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
for (B, L, H) in [(2, 197, 3), (26, 197, 12), (300, 197, 1)]:
    dh = 64; d = H * dh
    g = torch.Generator().manual_seed(L + dh)
    qkv = torch.randn(B * L, 3 * d, generator=g).bfloat16().cuda()
    dout = torch.randn(B * L, d, generator=g).bfloat16().cuda()
    out, lse = ops.attn_fwd(qkv, B, L, H, dh, False)
    os.environ["SC_ATTN_BWD3"] = "0"; os.environ["SC_ATTN_BWD2"] = "0"; os.environ["SC_ATTN_BWD1"] = "0"
    ref = ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False).clone().float()
    os.environ["SC_ATTN_BWD3"] = "1"
    for rep in range(2):
        got = torch.full((B * L, 3 * d), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=got)
        diff = (got.float() - ref).abs()
        bad = (diff > 0.04 + 0.04 * ref.abs()).nonzero()
        print((B, L, H), "rep", rep, "bad elements", bad.shape[0], flush=True)
        seen = {}
        for r, c in bad.tolist():
            key = (r // L, ("dq", "dk", "dv")[c // d], (c % d) // dh, (r % L) // 16)
            seen[key] = seen.get(key, 0) + 1
        for k in sorted(seen)[:40]:
            print("    batch %d %s head %d rows %d..%d: %d" % (k[0], k[1], k[2], k[3] * 16, k[3] * 16 + 15, seen[k]))
