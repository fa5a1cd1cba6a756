#!/usr/bin/env python3
"""Round-5 verdict item 4a: the L = 197 attention backward takes 13.3 us per head with 128 workgroups and 18.2 us with all 256.
Clock, or memory system?  Diagnostic build of sc_attention_bwd3.hip (-DSC_BWD3_CLOCK, tools/build_variant.py) stamps
s_memtime / s_memrealtime around every workgroup's walk; this tool runs >= 2 s of back-to-back launches per grid size on random
data and prints, per grid: wall time per launch, in-kernel clock (median over workgroups), shader cycles and microseconds per
head.  If cycles per head stay put and the clock drops, it is DVFS; if cycles per head grow, it is the memory system.

    python tools/build_variant.py clk sc_attention_bwd3.hip -DSC_BWD3_CLOCK
    SC_HIP_LIB=spatial-clip_amd/lib/libspatialclip_hip_clk.so python tools/attn_bwd3_clock.py
Without the variant library the stamps are skipped and only wall times are printed (the form the PMC passes run)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa: F401,E402
from spatial_clip_amd import _lib, ops  # noqa: E402

B, L, H, dh = int(os.environ.get("B", 256)), int(os.environ.get("L", 197)), int(os.environ.get("H", 12)), 64
d = H * dh
SECONDS = float(os.environ.get("SECONDS", 2.0))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * L, 3 * d, device="cuda", generator=g).bfloat16()
out = torch.empty(B * L, d, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dout = torch.randn(B * L, d, device="cuda", generator=g).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, L, device="cuda")
ops.attn_fwd(qkv, B, L, H, dh, False, out=out, lse=lse)
lib = _lib.lib()            # SC_HIP_LIB selects the diagnostic build; the same CDLL object carries the extra export if it exists
try:
    stamps = lib.sc_debug_bwd3_stamps
    stamps.restype, stamps.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
except AttributeError:
    stamps = None
print(f"# attention backward ring kernel, B={B} H={H} L={L} dh={dh}: {B * H} heads; stamps {'on' if stamps else 'off (default library)'}", flush=True)


def run(grid):
    os.environ["SC_ATTN_GRID"] = str(grid)
    fn = lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=dqkv, delta=delta)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = int(os.environ.get("N", 0)) or 200
    t0 = time.time()
    total = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(n):
            fn()
        total += n
        torch.cuda.synchronize()
        if time.time() - t0 >= SECONDS or os.environ.get("N"):
            break
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / total * 1e3
    line = f"grid {grid:4d}: {us:8.1f} us per launch, {us * min(grid, 256) / (B * H):6.2f} us per head and workgroup"
    if stamps is not None:
        buf = (ctypes.c_ulonglong * (4 * 1024))()
        rc = stamps(buf, 4 * 1024)
        assert rc == 0, rc
        rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]) for i in range(min(grid, 1024)) if buf[4 * i + 1] > 0]
        clk = sorted(c / r * 0.1 for c, r, h in rows)          # GHz: cycles per 10-ns tick x 0.1
        cyc = sorted(c / h for c, r, h in rows)
        ush = sorted(r / h * 0.01 for c, r, h in rows)
        m = len(rows) // 2
        line += (f"; in-kernel clock median {clk[m]:.3f} GHz (min {clk[0]:.3f}, max {clk[-1]:.3f}); per head: {cyc[m]:8.0f} shader cycles, "
                 f"{ush[m]:6.2f} us (median over {len(rows)} workgroups)")
    print(line, flush=True)


for grid in [int(x) for x in os.environ.get("GRIDS", "64,128,192,256").split(",")]:
    run(grid)
os.environ.pop("SC_ATTN_GRID", None)
