#!/usr/bin/env python3
"""Where does the GELU-pair GEMM's epilogue time go?  Same launch with (1) the second store dropped, (2) the GELU
arithmetic dropped, (3) both -- against the plain bf16 GEMM of the same shape (SC_EPI_DIAG, read per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
M, d, mlp = 256 * 197, 768, 3072
for rep in range(2):
    for diag, what in ((0, "full"), (1, "no h store"), (2, "no GELU math"), (3, "neither")):
        os.environ["SC_EPI_DIAG"] = str(diag)
        run(f"c_fc fwd gelu pair [{what}]", ops.NT, ops.EPI_GELU_PAIR, M, mlp, d)
    os.environ["SC_EPI_DIAG"] = "0"
    run("same shape, bf16 + bias", ops.NT, ops.EPI_BF16_BIAS, M, mlp, d)
    run("same shape, dgelu", ops.NT, ops.EPI_BF16_DGELU, M, mlp, d)
