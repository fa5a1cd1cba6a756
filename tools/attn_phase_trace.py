#!/usr/bin/env python3
"""Phase timing inside the persistent two-pass attention backward (debug build with -DSC_ATTN_TRACE):

    python tools/attn_phase_trace.py        # builds lib/libspatialclip_hip_trace.so, runs B=256 L=197 H=12, prints us per phase

Stamps are s_memtime cycles of workgroup 0 (compute wave 0 and the first loader wave) over its first 8 heads."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "spatial-clip_amd", "csrc")
OBJ = os.path.join(ROOT, "spatial-clip_amd", "build")
LIB = os.path.join(ROOT, "spatial-clip_amd", "lib", "libspatialclip_hip_trace.so")


def build():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-value"]
    o = os.path.join(OBJ, "sc_attention_bwd2_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DSC_ATTN_TRACE", "-c", os.path.join(CS, "sc_attention_bwd2.hip"), "-o", o])
    objs = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and f not in ("sc_attention_bwd2.o", "sc_attention_bwd2_trace.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, o, "-o", LIB])


if __name__ == "__main__":
    if "--build-only" in sys.argv or not os.path.exists(LIB):
        build()
        if "--build-only" in sys.argv:
            sys.exit(0)
    os.environ["SC_HIP_LIB"] = LIB
    os.environ["SC_ATTN_BWD1"] = "0"
    import torch
    import spatial_clip_amd  # noqa
    from spatial_clip_amd import ops, _lib
    B, L, H, dh = 256, 197, 12, 64
    d = H * dh
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * L, 3 * d, device="cuda", generator=g).bfloat16()
    dout = torch.randn(B * L, d, device="cuda", generator=g).bfloat16()
    out, lse = ops.attn_fwd(qkv, B, L, H, dh, False)
    dqkv = torch.empty_like(qkv)
    for _ in range(3):
        ops.attn_bwd(qkv, out, dout, lse, B, L, H, dh, False, dqkv=dqkv)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    fn = _lib.lib().sc_debug_attn_trace
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    t = list(buf)
    mhz = 100.0          # s_memtime ticks at the 100 MHz reference clock on this part
    names_c = ["A start", "pass A done", "dq stored", "B start", "pass B done", "ops + dk/dv issued"]
    names_l = ["A start", "Q,dO landed", "B start", "K/V free", "K,V landed"]
    t0 = t[0]
    for i in range(1, 7):
        c = [(t[i * 8 + s] - t0) / mhz for s in range(6)]
        l = [(t[64 + i * 8 + s] - t0) / mhz for s in range(5)]
        print(f"head {i}: compute " + "  ".join(f"{n} {v:7.2f}" for n, v in zip(names_c, c)))
        print(f"        loader  " + "  ".join(f"{n} {v:7.2f}" for n, v in zip(names_l, l)))
