"""Model registry.  The reference resolves ``model_name`` through src/open_clip/model_configs/*.json
(src/open_clip/factory.py:392-502); the same names resolve here to the same architectures (values restated from
those JSON files).  ``<name>-gene`` swaps the reference's CLIP text tower for the gene-expression MLP named by
BASELINE.json (no reference counterpart: SURVEY.md section 8a row G); ``ViT-Ti-16-gene`` is the small config of
BASELINE.json configs[0]."""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Dict, Optional


@dataclass
class VisionCfg:
    image_size: int = 224
    patch_size: int = 16
    width: int = 768
    layers: int = 12
    head_width: int = 64
    mlp_ratio: float = 4.0

    @property
    def heads(self) -> int:
        return self.width // self.head_width          # src/open_clip/model.py:170

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


@dataclass
class TextCfg:
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    heads: int = 8
    layers: int = 12
    mlp_ratio: float = 4.0


@dataclass
class GeneCfg:
    """Gene-expression tower (no reference symbol: SURVEY.md 8a row G).  ``kind="mlp"``: n_genes -> hidden -(GELU)->
    embed_dim (BASELINE configs[0]-[3]).  ``kind="transformer"`` (configs[4]): the expression vector is cut into
    ceil(n_genes / patch) contiguous patches of ``patch`` genes (zero padded), each embedded by one bias-free linear
    map -- a 1-D ViT patch embedding -- then class token + learned positions, ln_pre, ``layers`` pre-LN residual
    attention blocks (the reference's ResidualAttentionBlock), ln_post on the class token, projection to embed_dim."""
    n_genes: int = 20000
    hidden: int = 512
    kind: str = "mlp"
    patch: int = 256
    width: int = 512
    layers: int = 6
    head_width: int = 64
    mlp_ratio: float = 4.0

    @property
    def tokens(self) -> int:
        return (self.n_genes + self.patch - 1) // self.patch + 1

    @property
    def heads(self) -> int:
        return self.width // self.head_width


@dataclass
class ModelCfg:
    embed_dim: int = 512
    vision: VisionCfg = field(default_factory=VisionCfg)
    text: Optional[TextCfg] = None
    gene: Optional[GeneCfg] = None
    init_logit_scale: float = math.log(1 / 0.07)     # src/open_clip/model.py:273
    # `quick_gelu: true` of the *-quickgelu model configs: act_layer = QuickGELU in BOTH reference towers
    # (src/open_clip/model.py:142-145,228; the OpenAI-pretrained weights were trained with it).  The gene towers are this
    # build's own definition and keep the exact-erf GELU.
    quick_gelu: bool = False


def _clip(embed, v_layers, v_width, patch, t_width, t_heads, t_layers=12, image=224) -> ModelCfg:
    return ModelCfg(embed_dim=embed, vision=VisionCfg(image, patch, v_width, v_layers),
                    text=TextCfg(77, 49408, t_width, t_heads, t_layers))


_REGISTRY: Dict[str, ModelCfg] = {
    "ViT-B-16": _clip(512, 12, 768, 16, 512, 8),
    "ViT-B-32": _clip(512, 12, 768, 32, 512, 8),
    "ViT-L-14": _clip(768, 24, 1024, 14, 768, 12),
    "ViT-S-16": _clip(384, 12, 384, 16, 384, 6),
    "ViT-S-32": _clip(384, 12, 384, 32, 384, 6),
    "ViT-Ti-16": _clip(512, 12, 192, 16, 256, 4),
}
# src/open_clip/model_configs/ViT-B-16-quickgelu.json, ViT-B-32-quickgelu.json, ViT-L-14-quickgelu.json: the same
# architectures with `"quick_gelu": true`
for _base in ("ViT-B-16", "ViT-B-32", "ViT-L-14"):
    _REGISTRY[_base + "-quickgelu"] = replace(_REGISTRY[_base], quick_gelu=True)


def get_model_config(model_name: str, n_genes: Optional[int] = None, gene_hidden: Optional[int] = None) -> ModelCfg:
    """``ViT-B-16`` -> reference architecture (vision + CLIP text tower); ``ViT-B-16-gene`` -> gene-MLP tower;
    ``ViT-L-14-genetr`` -> 6-layer gene transformer tower (BASELINE configs[4])."""
    name = model_name
    gene = False
    kind = "mlp"
    if name.endswith("-genetr"):
        name, gene, kind = name[:-7], True, "transformer"
    elif name.endswith("-gene"):
        name, gene = name[:-5], True
    if name not in _REGISTRY:
        # same failure mode as open_clip.factory.create_model (factory.py:399-402)
        raise RuntimeError(f"Model config for {model_name} not found. Available: {sorted(list_models())}")
    cfg = _REGISTRY[name]
    cfg = ModelCfg(cfg.embed_dim, replace(cfg.vision), replace(cfg.text) if cfg.text else None, None,
                   cfg.init_logit_scale, cfg.quick_gelu)
    if gene:
        cfg.text = None
        cfg.gene = GeneCfg(n_genes or 20000, gene_hidden or 512, kind=kind)
    return cfg


def list_models():
    return list(_REGISTRY) + [n + "-gene" for n in _REGISTRY] + [n + "-genetr" for n in _REGISTRY]
