#!/usr/bin/env python3
"""Which NT kernel for SMALL token counts?  ViT-B-32 + CLIP text tower at the reference's batch 32 has M = 1600 (vision) / 2464
(text) token rows: a 256x256 tiling leaves most CUs idle (c_proj: 7 x 3 = 21 tiles).  Times the phase-interleaved 256x256
kernel (default choice) against the 128x128 general kernel (SC_GEMM_FORCE=128 in a child process) on the step's shapes.

    python tools/bench_small_m.py            # parent: runs itself twice and prints the table"""
import os
import subprocess
import sys

SHAPES = [  # name, M, N, K
    ("vision qkv fwd      B=32", 1600, 2304, 768), ("vision out_proj     B=32", 1600, 768, 768), ("vision c_fc fwd     B=32", 1600, 3072, 768),
    ("vision c_proj fwd   B=32", 1600, 768, 3072), ("text qkv fwd        B=32", 2464, 1536, 512), ("text out_proj       B=32", 2464, 512, 512),
    ("text c_fc fwd       B=32", 2464, 2048, 512), ("text c_proj fwd     B=32", 2464, 512, 2048),
    ("vision qkv fwd      B=256", 12800, 2304, 768), ("vision out_proj     B=256", 12800, 768, 768), ("vision c_fc fwd     B=256", 12800, 3072, 768),
    ("vision c_proj fwd   B=256", 12800, 768, 3072), ("text qkv fwd        B=256", 19712, 1536, 512), ("text out_proj       B=256", 19712, 512, 512),
    ("text c_fc fwd       B=256", 19712, 2048, 512), ("text c_proj fwd     B=256", 19712, 512, 2048),
]


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import spatial_clip_amd  # noqa: F401
    from spatial_clip_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm(ops.NT, ops.EPI_BF16, a, b, out, M=M, N=N, K=K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.gemm(ops.NT, ops.EPI_BF16, a, b, out, M=M, N=N, K=K)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}|{e0.elapsed_time(e1) / n * 1e3:.2f}", flush=True)


def main():
    if os.environ.get("SC_SMALL_M_CHILD") == "1":
        return child()
    res = {}
    for tag, force in (("256x256 (8-phase)", None), ("128x128 (general)", "128")):
        env = dict(os.environ, SC_SMALL_M_CHILD="1")
        env.pop("SC_GEMM_FORCE", None)
        if force:
            env["SC_GEMM_FORCE"] = force
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            sys.exit(1)
        for line in r.stdout.splitlines():
            if "|" in line:
                k, v = line.split("|")
                res.setdefault(k, {})[tag] = float(v)
    print(f"{'shape':28s} {'M':>6s} {'N':>5s} {'K':>5s} {'tiles256':>8s} | {'256x256 us':>10s} {'128x128 us':>10s}  TF/s(256) TF/s(128)")
    for name, M, N, K in SHAPES:
        t = res[name]
        a, b = t["256x256 (8-phase)"], t["128x128 (general)"]
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"{name:28s} {M:6d} {N:5d} {K:5d} {tiles:8d} | {a:10.2f} {b:10.2f}  {2.0 * M * N * K / a / 1e6:9.1f} {2.0 * M * N * K / b / 1e6:9.1f}")


if __name__ == "__main__":
    main()
