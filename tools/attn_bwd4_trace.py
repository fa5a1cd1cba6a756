#!/usr/bin/env python3
"""Phase timing inside the eight-wave ring attention backward (sc_attention_bwd4.hip, debug build with -DSC_ATTN_TRACE):

    python tools/build_variant.py trace sc_attention_bwd4.hip -DSC_ATTN_TRACE        (build container)
    python tools/attn_bwd4_trace.py                                                   (GPU box)

s_memrealtime stamps (100 MHz) of workgroup 0, heads 1..3, every wave: end of each sweep step, reduce start / end, end phase.
(The stamps' own global stores make the compiler drain memory operations there: phases that overlap loads or stores read longer
than in the shipped build.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SC_HIP_LIB"] = os.path.join(ROOT, "spatial-clip_amd", "lib", "libspatialclip_hip_trace.so")
import torch
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops, _lib

B, L, H, dh = 256, int(os.environ.get("L", 257)), 16, 64
d = H * dh
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * L, 3 * d, device="cuda", generator=g).bfloat16()
dout = (torch.randn(B * L, d, device="cuda", generator=g) * 0.05).bfloat16()
out, lse = ops.attn_fwd(qkv, B, L, H, dh, False)
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=dqkv)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 8 * 32))()
fn = _lib.lib().__getattr__("sc_debug_attn_trace4")
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
assert fn(buf) == 0
t = list(buf)
nbq = 9 if L > 256 else 8
for i in (1, 2):
    t0 = min(t[(i * 8 + w) * 32 + 0] for w in range(8))
    print(f"head {i} (us from the first wave's start; 100 MHz stamps)")
    for w in range(8):
        r = [(t[(i * 8 + w) * 32 + s] - t0) / 100.0 for s in range(20)]
        steps = " ".join(f"{r[1 + j]:5.1f}" for j in range(nbq))
        print(f"  wave {w}: start {r[0]:5.1f} | steps end {steps} | reduces {r[12]:5.1f}-{r[13]:5.1f} {r[18]:5.1f}-{r[19]:5.1f} | sweep + tail work done {r[14]:5.1f} "
              f"next head's loads consumed {r[15]:5.1f} dK/dV stores issued {r[16]:5.1f} barrier {r[17]:5.1f}")
    t1 = min(t[((i + 1) * 8 + w) * 32 + 0] for w in range(8))
    print(f"  next head starts at {(t1 - t0) / 100.0:5.1f}")
