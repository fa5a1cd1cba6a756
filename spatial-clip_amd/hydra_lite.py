"""A small Hydra/OmegaConf-compatible composer (hydra, omegaconf and lightning are not installed on the MI355X box).

Understands what the reference's ``configs/`` tree uses (SURVEY.md section 5 "Config"): defaults lists (``_self_``,
``group: name``, ``optional group: name``, ``group: null``, same-group ``- default``), experiment files with
``# @package _global_`` and ``override /group: name``, command-line overrides (``group=name``, ``a.b=c``, ``+a.b=c``,
``++a.b=c``, ``~a.b``), ``${a.b}`` / ``${oc.env:VAR[,default]}`` / ``${hydra:runtime.cwd|output_dir}``
interpolation, and ``_target_`` / ``_partial_`` instantiation.  Reference ``_target_`` paths
(``src.models.spatial_clip_module.SpatialClipLitModule`` ...) are mapped onto this package's classes, so a reference
config directory can be composed as-is; groups missing from it (``data/spatial.yaml`` is absent from the reference
snapshot) fall back to this package's own ``configs/``."""
from __future__ import annotations

import copy
import functools
import importlib
import os
import re
from typing import Any, Dict, List, Optional, Tuple

import yaml

OWN_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

TARGET_MAP = {
    "src.models.spatial_clip_module.SpatialClipLitModule": "spatial_clip_amd.module.SpatialClipLitModule",
    "src.models.components.spatial_clip_net.SpatialClipNet": "spatial_clip_amd.net.SpatialClipNet",
    "src.models.components.losses.SpatialLoss": "spatial_clip_amd.losses.SpatialLoss",
    "src.models.components.losses.ClipLoss": "spatial_clip_amd.losses.ClipLoss",
    "src.models.components.metrics.ContrastiveMetrics": "spatial_clip_amd.metrics.ContrastiveMetrics",
    "src.data.spatial_datamodule.SpatialClipDataModule": "spatial_clip_amd.data.SpatialClipDataModule",
    "open_clip.AugmentationCfg": "spatial_clip_amd.net.AugmentationCfg",
    "torch.optim.AdamW": "spatial_clip_amd.optim.FusedAdamW",
    "transformers.get_cosine_schedule_with_warmup": "spatial_clip_amd.optim.get_cosine_schedule_with_warmup",
    "lightning.pytorch.Trainer": "spatial_clip_amd.trainer.Trainer",
    "lightning.Trainer": "spatial_clip_amd.trainer.Trainer",
}


class _Loader(yaml.SafeLoader):
    """SafeLoader with OmegaConf's float grammar (PyYAML's YAML-1.1 resolver reads ``1e-4`` as a string)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                   |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                   |\.[0-9_]+(?:[eE][-+][0-9]+)?
                   |[-+]?\.(?:inf|Inf|INF)
                   |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _yaml(text: str):
    return yaml.load(text, Loader=_Loader)


class Cfg(dict):
    """dict with attribute access (the slice of DictConfig the entry points use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return super().get(k, default)


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _load(path: str) -> Tuple[dict, bool]:
    text = open(path).read()
    is_global = bool(re.search(r"^#\s*@package\s+_global_", text, flags=re.M))
    return (_yaml(text) or {}), is_global


def _find(search_path: List[str], rel: str) -> Optional[str]:
    for d in search_path:
        for ext in ("", ".yaml", ".yml"):
            p = os.path.join(d, rel + ext)
            if os.path.isfile(p):
                return p
    return None


def _is_group(search_path: List[str], name: str) -> bool:
    return any(os.path.isdir(os.path.join(d, name)) for d in search_path)


def _set_path(cfg: dict, dotted: str, value, create: bool = True) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create:
                raise KeyError(f"Could not override '{dotted}': key '{k}' is not in the config (use +{dotted}=...)")
            node[k] = {}
        node = node[k]
    if not create and keys[-1] not in node:
        raise KeyError(f"Could not override '{dotted}': no such key (use +{dotted}=... to add it)")
    node[keys[-1]] = value


def _del_path(cfg: dict, dotted: str) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        node = node[k]
    node.pop(keys[-1], None)


def _load_group(search_path: List[str], group: str, name: str, cfg: dict, choices: Dict[str, str],
                optional: bool = False) -> None:
    """Load <group>/<name>.yaml (with its own same-group defaults) and merge it under cfg[group] (or globally)."""
    if group.split("/")[0] in ("hydra", "hydra_logging", "job_logging"):
        return                                        # Hydra's own runtime / logging plugins: nothing to compose
    path = _find(search_path, os.path.join(group, name))
    if path is None:
        if optional:
            return
        raise FileNotFoundError(f"config group '{group}' has no option '{name}' in {search_path}")
    body, is_global = _load(path)
    defaults = body.pop("defaults", [])
    later: List[Tuple[str, str]] = []
    for d in defaults:
        if d == "_self_":
            continue
        if isinstance(d, str):                       # same-group include, e.g. trainer/gpu.yaml: - default
            _load_group(search_path, group, d, cfg, choices)
        elif isinstance(d, dict):
            (k, v), = d.items()
            if k.startswith("override "):
                later.append((k[len("override "):].strip().lstrip("/"), v))
            elif v is not None:
                _load_group(search_path, k.replace("optional ", "").strip().lstrip("/"), v, cfg, choices,
                            optional=k.startswith("optional "))
    for g, v in later:                               # 'override /loss: clip' replaces the earlier choice
        choices[g] = v
        cfg.pop(g, None)
        _load_group(search_path, g, v, cfg, choices)
    if is_global:
        _merge(cfg, body)
    else:
        node = cfg.setdefault(group, {})
        if not isinstance(node, dict):
            cfg[group] = node = {}
        _merge(node, body)


def compose(config_name: str = "train.yaml", overrides: Optional[List[str]] = None,
            config_dir: Optional[str] = None, resolve: bool = True) -> Cfg:
    search_path = ([config_dir] if config_dir else []) + [OWN_CONFIG_DIR]
    overrides = list(overrides or [])
    root_path = _find(search_path, config_name.replace(".yaml", ""))
    if root_path is None:
        raise FileNotFoundError(f"{config_name} not found in {search_path}")
    root, _ = _load(root_path)
    defaults = root.pop("defaults", ["_self_"])
    group_over: Dict[str, str] = {}
    value_over: List[Tuple[str, str, Any]] = []
    for o in overrides:
        if o.startswith("~"):
            value_over.append(("del", o[1:], None))
            continue
        key, _, val = o.partition("=")
        mode = "set"
        if key.startswith("++"):
            key, mode = key[2:], "force"
        elif key.startswith("+"):
            key, mode = key[1:], "add"
        if "." not in key and _is_group(search_path, key) and mode != "add":
            group_over[key] = val
        elif "." not in key and _is_group(search_path, key) and mode == "add":
            group_over[key] = val
            defaults.append({key: None})
        else:
            value_over.append((mode, key, _yaml(val) if val != "" else ""))
    cfg: dict = {}
    choices: Dict[str, str] = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(cfg, root)
            self_done = True
            continue
        (k, v), = d.items() if isinstance(d, dict) else ((d, None),)
        optional = k.startswith("optional ")
        g = k.replace("optional ", "").strip()
        v = group_over.get(g, v)
        if v is None or v == "null":
            continue
        choices[g] = v
        _load_group(search_path, g, v, cfg, choices, optional=optional)
    if not self_done:
        _merge(cfg, root)
    for mode, key, val in value_over:
        if mode == "del":
            _del_path(cfg, key)
        else:
            _set_path(cfg, key, val, create=(mode != "set") or True)
    cfg.setdefault("hydra", {})
    out = _wrap(cfg)
    out["_choices_"] = Cfg(choices)
    if resolve:
        resolve_interpolations(out)
    return out


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(root: dict, dotted: str):
    node = root
    for k in dotted.split("."):
        if isinstance(node, list):
            node = node[int(k)]
        else:
            node = node[k]
    return node


def _resolve_str(root: dict, s: str, depth: int = 0):
    if depth > 20:
        raise RecursionError(f"interpolation loop in {s!r}")

    def one(expr: str):
        expr = expr.strip()
        if expr.startswith("oc.env:"):
            var, _, default = expr[len("oc.env:"):].partition(",")
            if var in os.environ:
                return os.environ[var]
            if default != "":
                return default
            if var == "PROJECT_ROOT":
                return os.getcwd()
            raise KeyError(f"environment variable {var} is not set")
        if expr.startswith("hydra:"):
            what = expr[len("hydra:"):]
            return {"runtime.cwd": os.getcwd(), "runtime.output_dir": os.path.join(os.getcwd(), "outputs")}.get(what, "")
        if expr.startswith("now:"):
            import time
            return time.strftime(expr[4:])
        return _lookup(root, expr)

    m = _INTERP.fullmatch(s)
    if m:                                               # whole-value interpolation keeps the node type
        v = one(m.group(1))
        return _resolve_any(root, copy.deepcopy(v), depth + 1)
    while True:
        m = _INTERP.search(s)
        if not m:
            return s
        s = s[:m.start()] + str(_resolve_any(root, one(m.group(1)), depth + 1)) + s[m.end():]


def _resolve_any(root: dict, node, depth: int = 0):
    if isinstance(node, str):
        return _resolve_str(root, node, depth) if "${" in node else node
    if isinstance(node, dict):
        for k in list(node.keys()):
            if k == "hydra":
                continue
            node[k] = _resolve_any(root, node[k], depth)
        return node
    if isinstance(node, list):
        return [_resolve_any(root, v, depth) for v in node]
    return node


def resolve_interpolations(cfg: dict) -> dict:
    return _resolve_any(cfg, cfg)


def _locate(target: str):
    target = TARGET_MAP.get(target, target)
    mod, _, attr = target.rpartition(".")
    if mod.startswith("spatial_clip_amd"):
        import spatial_clip_amd  # noqa: F401  (import shim)
    return getattr(importlib.import_module(mod), attr)


def instantiate(node, *args, **kwargs):
    """hydra.utils.instantiate: recursive, honours _target_ / _partial_ ; extra kwargs override config keys."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return _wrap({k: instantiate(v) for k, v in node.items()})
    params = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_", "_recursive_", "_convert_")}
    params.update(kwargs)
    fn = _locate(node["_target_"])
    if node.get("_partial_", False):
        return functools.partial(fn, *args, **params)
    return fn(*args, **params)
