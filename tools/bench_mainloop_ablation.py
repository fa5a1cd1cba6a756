#!/usr/bin/env python3
"""What bounds a K tile of the phase-interleaved NT kernel?  The same launch with the operand stream removed (the
fragments are read from whatever the ring holds), with the MFMAs removed, and with both (reads + barriers only).
SC_EPI_DIAG bits 4 / 8, read per call; results are garbage by construction -- timing only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
os.environ["SC_GEMM_PERSIST"] = "0"
for name, M, N, K in (("4096^3", 4096, 4096, 4096), ("c_fc dgrad", 256 * 197, 768, 3072), ("8192x8192x4096", 8192, 8192, 4096)):
    for rep in range(2):
        for diag, what in ((0, "full"),):   # the ablation switches (bits 4 / 8) were compiled out again after the round-3 measurement (profiles/r03_mainloop_ablation.txt)
            os.environ["SC_EPI_DIAG"] = str(diag)
            run(f"{name} [{what}]", ops.NT, ops.EPI_BF16, M, N, K)
os.environ["SC_EPI_DIAG"] = "0"
