"""spatial-clip_amd: MI355X-native Spatial-CLIP contrastive training step.

Hand-written gfx950 HIP kernels behind a C ABI (``include/spatial_clip_hip.h``), with a host-side mirror
of the reference's LightningModule / Hydra surface.  There is no CPU or eager fallback: every compute
entry point raises if ``libspatialclip_hip.so`` cannot be loaded.
"""
__version__ = "0.1.0"
