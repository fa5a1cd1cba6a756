#!/usr/bin/env python3
"""Short tiles (SC_GEMM_TILE_F) on the N = 768 / 1024 tower GEMMs: 256-row tiling vs every fragment count, same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatial_clip_amd  # noqa
from spatial_clip_amd import ops
from tools.bench_gemm import run
shapes = [("ViT-B out fwd (res f32)", ops.EPI_F32_BIAS_RES, 256 * 197, 768, 768),
          ("ViT-B c_proj fwd (res f32)", ops.EPI_F32_BIAS_RES, 256 * 197, 768, 3072),
          ("ViT-B out dgrad", ops.EPI_BF16, 256 * 197, 768, 768),
          ("ViT-B c_fc dgrad", ops.EPI_BF16, 256 * 197, 768, 3072),
          ("ViT-B qkv dgrad", ops.EPI_BF16, 256 * 197, 768, 2304),
          ("ViT-L out fwd (res f32)", ops.EPI_F32_BIAS_RES, 256 * 257, 1024, 1024),
          ("ViT-L c_proj fwd (res f32)", ops.EPI_F32_BIAS_RES, 256 * 257, 1024, 4096),
          ("ViT-L c_fc dgrad", ops.EPI_BF16, 256 * 257, 1024, 4096)]
for name, epi, M, N, K in shapes:
    for f in ("16", "15", "14", "13", "12", "0"):
        os.environ["SC_GEMM_TILE_F"] = f
        run(f"{name} F={f if f != '0' else 'auto'}", ops.NT, epi, M, N, K)
os.environ.pop("SC_GEMM_TILE_F")
