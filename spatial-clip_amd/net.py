"""SpatialClipNet: the reference's two-tower wrapper, backed by the HIP towers.

Mirror of ``src/models/components/spatial_clip_net.py:12-53`` (ctor kwargs ``model_name, pretrained, aug_cfg,
cache_dir``; attributes ``model``, ``preprocess_train``, ``preprocess_val``, ``tokenizer``; ``forward(images,
texts) -> {"image_features","text_features","logit_scale","logit_bias"}``).  The ``texts`` slot carries the float
gene matrix [B, n_genes] for ``*-gene`` models.

Autograd bridge: the whole net is ONE autograd node.  Its backward runs the hand-written backward kernel sequence
and writes parameter gradients directly into the flat gradient buffer (``p.grad`` are views of it), so the usual
``loss.backward(); optimizer.step()`` driver works unchanged.  Gradients are overwritten, not accumulated, by each
backward (the reference path never uses gradient accumulation)."""
from __future__ import annotations

import math
import os
from dataclasses import is_dataclass
from typing import Any, Callable, Dict, List, Optional

import torch

from . import ops, streams
from .model_configs import ModelCfg, get_model_config
from .params import ParamStore
from .towers import GeneTower, GeneTransformerTower, TextTower, VisionTower

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # src/open_clip/constants.py:1-2
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


class AugmentationCfg(dict):
    """Stand-in for ``open_clip.AugmentationCfg`` (configs/model/spatial_clip.yaml:12-17): augmentation is CPU-side
    data preparation outside the hot path; the values are kept for the data pipeline."""

    def __init__(self, **kw):
        super().__init__(**kw)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net, images, texts):
        ctx.net = net
        # second tower first: its features (and the tile ids) start travelling on the communication stream while the
        # vision tower -- >99 % of the step's FLOPs -- is still running (comm.FeatureGather)
        fg = net.feature_gather
        side_by_side = net._towers_side_by_side()
        if side_by_side:
            # single process, second tower a transformer: the towers are independent until the loss -- the second one runs on
            # its own stream beside the vision tower (round 6; fork / join by events, also inside a captured graph)
            cur = torch.cuda.current_stream(net.device_)
            ts = streams.tower_stream(net.device_)
            ts.wait_stream(cur)
            with torch.cuda.stream(ts):
                txt = net.second.forward(texts)
            img = net.vision.forward(images)
            cur.wait_stream(ts)
        else:
            txt = net.second.forward(texts)
            if fg is not None:
                fg.put("text", txt)
            img = net.vision.forward(images)
            if fg is not None:
                fg.put("image", img)
        s = torch.empty(1, dtype=torch.float32, device=img.device)
        net.store.wait_names(["logit_scale"])
        ops.exp_scalar(net.store.p("logit_scale").view(1), s)
        net._scale = s
        # Return ALIASES, not the tensors the towers keep (tower.f): autograd stamps this node as grad_fn on the objects
        # returned here, and a tensor that the net holds AND that points back at a node holding the net (ctx.net) is a
        # reference cycle through C++ that Python's collector cannot see -- every net ever built would keep its HBM.
        return img.detach(), txt.detach(), s.view(())

    @staticmethod
    def backward(ctx, d_img, d_txt, d_s):
        ctx.net.store.wait_all()      # an optimiser update running behind the forward reads the gradients this backward overwrites
        ctx.net._backward(d_img, d_txt, d_s)
        return None, None, None, None


def resize_pos_embed(state_dict: Dict[str, torch.Tensor], grid_size, interpolation: str = "bicubic",
                     antialias: bool = True) -> None:
    """Rescale the grid of visual position embeddings of a checkpoint to this model's patch grid when they differ
    (``src/open_clip/model.py:792-823``; load-time host work): class token kept, grid bicubic-resampled with
    antialiasing, ``align_corners=False``.  In place on ``state_dict['visual.positional_embedding']``."""
    old = state_dict.get("visual.positional_embedding")
    if old is None:
        return
    gh, gw = int(grid_size[0]), int(grid_size[1])
    extra = 1
    if gh * gw + extra == old.shape[0]:
        return
    tok, img = old[:extra], old[extra:]
    og = int(math.sqrt(len(img)))
    img = img.float().reshape(1, og, og, -1).permute(0, 3, 1, 2)
    img = torch.nn.functional.interpolate(img, size=(gh, gw), mode=interpolation, antialias=antialias, align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(gh * gw, -1)
    state_dict["visual.positional_embedding"] = torch.cat([tok.float(), img], dim=0).to(old.dtype)


def resize_text_pos_embed(state_dict: Dict[str, torch.Tensor], num_pos: int, interpolation: str = "linear",
                          antialias: bool = False, model_width: Optional[int] = None) -> None:
    """Resample the text tower's positional embedding of a checkpoint to this model's context length when they differ
    (``src/open_clip/model.py:826-860``, called from ``load_checkpoint`` right after ``resize_pos_embed``,
    factory.py:220-221): 1-D linear interpolation along the position axis, ``align_corners=False``, width unchanged.
    In place on ``state_dict['positional_embedding']`` (or ``'text.positional_embedding'``, the custom-text layout)."""
    key = "positional_embedding" if "positional_embedding" in state_dict else "text.positional_embedding"
    old = state_dict.get(key)
    if old is None:
        return
    old_num, width = old.shape
    if old_num == int(num_pos):
        return
    if model_width is not None and int(model_width) != int(width):      # model.py:840: 'text pos_embed width changed!'
        raise AssertionError(f"text pos_embed width changed: checkpoint {width}, model {model_width}")
    x = old.float().reshape(1, old_num, width).permute(0, 2, 1)
    x = torch.nn.functional.interpolate(x, size=int(num_pos), mode=interpolation, antialias=antialias, align_corners=False)
    state_dict[key] = x.permute(0, 2, 1)[0].to(old.dtype)


def read_checkpoint_file(path: str, unsafe_pickle: Optional[bool] = None) -> Dict[str, Any]:
    """What ``open_clip.factory.load_state_dict`` accepts (src/open_clip/factory.py:153-178): a ``.safetensors`` file
    (``safetensors.torch.load_file``), a pickled state_dict / training checkpoint (``{'state_dict': ...}``), or a TorchScript
    archive (OpenAI's released ``.pt`` files: ``state_dict()`` minus the three bookkeeping buffers).  Host tensors.

    Like the reference, pickles are read with ``weights_only=True`` and nothing else by default: a file that pickles more than
    tensors (a Lightning checkpoint with hyper-parameters / callbacks) is refused with the original error unless the caller
    opts in -- ``unsafe_pickle=True`` or ``SC_UNSAFE_PICKLE=1`` -- because unpickling arbitrary objects executes code, and
    ``pretrained`` is typically a downloaded file (advisor, round 5).  ``Trainer.load_checkpoint`` (the user's own resume
    files) opts in itself."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    if unsafe_pickle is None:
        unsafe_pickle = os.environ.get("SC_UNSAFE_PICKLE", "0") == "1"
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as first:
        try:        # TorchScript archives are zip files torch.load(weights_only=True) refuses; torch.jit.load reads them
            ckpt = torch.jit.load(path, map_location="cpu")
        except Exception:
            if not unsafe_pickle:
                raise RuntimeError(
                    f"{path}: not a weights-only pickle, a .safetensors file or a TorchScript archive ({type(first).__name__}: "
                    f"{first}).  If this is a TRUSTED checkpoint that pickles more than tensors (e.g. a Lightning file), pass "
                    "unsafe_pickle=True or set SC_UNSAFE_PICKLE=1") from first
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(ckpt, torch.jit.ScriptModule):
        sd = dict(ckpt.state_dict())
        for k in ("input_resolution", "context_length", "vocab_size"):
            sd.pop(k, None)
        return sd
    return ckpt


_CKPT_PREFIXES = ("net.model.", "model.", "module.", "net.")


def strip_checkpoint_prefix(key: str) -> str:
    """``net.model.visual.conv1.weight`` (Lightning: SpatialClipLitModule.net.model = CLIP), ``module.visual...`` (DDP)
    -> ``visual.conv1.weight``.  Applied repeatedly (``module.net.model.`` occurs)."""
    changed = True
    while changed:
        changed = False
        for p in _CKPT_PREFIXES:
            if key.startswith(p):
                key, changed = key[len(p):], True
    return key


class _ClipFacade:
    """What the reference reaches through ``net.model`` (encode_image / encode_text / logit_scale)."""

    def __init__(self, net: "SpatialClipNet"):
        self._net = net

    @property
    def logit_scale(self) -> torch.nn.Parameter:
        st = self._net.store
        st.wait_names(["logit_scale"])      # an update may still be running behind the forward (communication stream)
        return st.params["logit_scale"]

    logit_bias = None

    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """CLIP.encode_image (src/open_clip/model.py:326-328), inference only: ``normalize=False`` returns the
        projected features before F.normalize."""
        with torch.no_grad():
            self._net._mark_pass(False)
            f = self._net.vision.forward(image)
            return (f if normalize else self._net.vision.raw_features()).clone()

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """CLIP.encode_text (src/open_clip/model.py:330-345) / the gene tower, inference only."""
        with torch.no_grad():
            self._net._mark_pass(False)
            f = self._net.second.forward(text)
            return (f if normalize else self._net.second.raw_features()).clone()

    def set_grad_checkpointing(self, enable: bool = True) -> None:
        """CLIP.set_grad_checkpointing (src/open_clip/model.py:313-315).  Here: activation recomputation inside the
        transformer stacks -- LayerNorm outputs and the GELU output are rebuilt in the backward instead of kept
        (one third less saved activation memory; results bit-identical to the default mode)."""
        self._net.set_grad_checkpointing(enable)


class SpatialClipNet(torch.nn.Module):
    def __init__(self, model_name: str, pretrained: Optional[str] = None, aug_cfg: Optional[Any] = None,
                 cache_dir: Optional[str] = None, n_genes: Optional[int] = None, gene_hidden: Optional[int] = None,
                 device: Optional[str] = None, seed: int = 0, model_cfg: Optional[ModelCfg] = None,
                 tokenizer_vocab: Optional[str] = None, precision: str = "bf16", grad_checkpointing: bool = False,
                 residual_stream: str = "bf16"):
        super().__init__()
        if aug_cfg is not None and not isinstance(aug_cfg, (dict, AugmentationCfg)) and not is_dataclass(aug_cfg) \
                and not hasattr(aug_cfg, "items"):
            raise TypeError(f"Unsupported type for aug_cfg: {type(aug_cfg)}")     # spatial_clip_net.py:33-34
        self.aug_cfg = aug_cfg
        if not torch.cuda.is_available():
            raise RuntimeError("SpatialClipNet needs an MI355X (HIP device): this build has no CPU fallback")
        self.device_ = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.cfg: ModelCfg = model_cfg if model_cfg is not None else get_model_config(model_name, n_genes, gene_hidden)
        if self.cfg.gene is None and self.cfg.text is None:
            raise ValueError(f"{model_name}: the model config has neither a text tower nor a gene tower")
        self.model_name = model_name
        if precision not in ("bf16", "bf16-mixed", "fp8", "fp8-mixed"):
            raise ValueError(f"precision {precision!r}: 'bf16' / 'bf16-mixed' (the reference's bf16-mixed policy) or "
                             "'fp8' / 'fp8-mixed' (e4m3 GEMMs of the transformer blocks, BASELINE configs[4])")
        self.precision = "fp8" if precision.startswith("fp8") else "bf16"
        self.store = ParamStore(self.cfg, self.device_, seed=seed, fp8=self.precision == "fp8")
        for name, p in self.store.params.items():
            p._sc_store = self.store
            self.register_parameter(name.replace(".", "__"), p)
        self.vision = VisionTower(self.cfg, self.store)
        if self.cfg.gene is not None:
            gene_cls = GeneTransformerTower if self.cfg.gene.kind == "transformer" else GeneTower
            self.second = gene_cls(self.cfg, self.store)
        else:
            self.second = TextTower(self.cfg, self.store)
        self.model = _ClipFacade(self)
        self.preprocess_train = self.preprocess_val = self._preprocess
        self.tokenizer = self._tokenizer
        self.tokenizer_vocab = tokenizer_vocab      # BPE merge table for string inputs of the text tower (tokenizer.py)
        self._bpe = None
        self.grad_bucket_hook: Optional[Callable[[int, int], None]] = None
        self.feature_gather = None          # comm.FeatureGather, installed per step by the module when W > 1
        if pretrained:
            self._load_pretrained(pretrained)
        if grad_checkpointing:              # config key model.net.grad_checkpointing (open_clip: --grad-checkpointing)
            self.set_grad_checkpointing(True)
        # config key model.net.residual_stream: "bf16" (default since round 4: what the reference's bf16 autocast keeps for
        # the image tower -- conv1 emits bf16, class / positional embeddings are cast to it, LayerNorm casts back,
        # src/open_clip/transformer.py:26-29,789-791) or "fp32" (the stream of rounds 1-3: +1.7 % step time, 0.4x the
        # feature noise against the fp32 oracle -- DESIGN.md section 7).  Patch towers only: the reference's TEXT tower
        # keeps an fp32 stream under autocast (fp32 embedding + fp32 LayerNorm output + bf16 branch = fp32), and so does
        # this build's.
        if residual_stream not in ("fp32", "bf16"):
            raise ValueError(f"residual_stream {residual_stream!r}: 'fp32' or 'bf16'")
        self.residual_stream = residual_stream
        for tower in (self.vision, self.second):
            stack = getattr(tower, "stack", None)
            if stack is not None:
                stack.res_stream = residual_stream

    # ------------------------------------------------------------------ reference-facing helpers
    def set_grad_checkpointing(self, enable: bool = True) -> None:
        for tower in (self.vision, self.second):
            stack = getattr(tower, "stack", None)
            if stack is not None:
                stack.set_grad_checkpointing(enable)

    def _towers_side_by_side(self) -> bool:
        """Run the second tower on its own stream beside the vision tower?  ``SC_TOWER_OVERLAP`` = auto (default) | 1 | 0, read per
        call.  auto: single process (with a process group the second tower goes FIRST so that its feature all-gather travels under
        the vision tower) and a transformer second tower (the gene-MLP is three launches).  Its stack then keeps its weight
        gradients on its own stream -- a weight-gradient side stream per tower would be the fifth busy stream of the device (§ stream
        roles: the hardware queues are few).  Same kernels on the same values: bit-identical to the sequential order."""
        mode = os.environ.get("SC_TOWER_OVERLAP", "auto")
        if mode == "0" or self.feature_gather is not None or self.grad_bucket_hook is not None:
            return False
        stack = getattr(self.second, "stack", None)
        on = stack is not None and (mode == "1" or mode == "auto")
        if stack is not None:
            stack.no_side_stream = on
        return on

    def _stacks(self):
        return [(name, t.stack) for name, t in (("vision", self.vision), ("second", self.second)) if getattr(t, "stack", None) is not None]

    def side_stream_choice(self) -> Dict[str, Optional[bool]]:
        """Per transformer stack: the weight-gradient schedule ``SC_OVERLAP=auto`` settled on for the batch shape of the last
        backward (True = side stream, False = one stream, None = still measuring, or pinned by SC_OVERLAP=0 / 1)."""
        out = {}
        for name, stack in self._stacks():
            st = getattr(stack, "_ov_auto", {}).get((getattr(stack, "B", None), getattr(stack, "L", None)))
            out[name] = None if st is None else st["choice"]
        return out

    def side_stream_timing(self) -> Dict[str, Optional[Dict[str, float]]]:
        """Per transformer stack: the best backward time (ms) ``SC_OVERLAP=auto`` measured on each schedule, or None."""
        out = {}
        for name, stack in self._stacks():
            st = getattr(stack, "_ov_auto", {}).get((getattr(stack, "B", None), getattr(stack, "L", None)))
            out[name] = None if st is None else st.get("ms_best")
        return out

    def reset_fp8_scaling(self) -> None:
        """fp8 path: drop the per-tensor scales carried from the previous steps (towers.TransformerStack.reset_fp8_scaling)."""
        for _, stack in self._stacks():
            stack.reset_fp8_scaling()

    def fp8_scaling_state(self) -> Optional[Dict[str, Any]]:
        """Delayed-scaling state of both towers for a checkpoint (None off the fp8 path)."""
        if self.precision != "fp8":
            return None
        return {name: stack.fp8_scaling_state() for name, stack in self._stacks()}

    def load_fp8_scaling_state(self, state: Optional[Dict[str, Any]]) -> None:
        """Restore what fp8_scaling_state() saved; anything missing or mismatched resets that tower's history instead."""
        for name, stack in self._stacks():
            stack.load_fp8_scaling_state((state or {}).get(name))

    def _load_pretrained(self, pretrained: str) -> None:
        if os.path.isfile(pretrained):        # local checkpoint path branch of factory.py:418-421
            self.load_checkpoint_state_dict(read_checkpoint_file(pretrained), source=pretrained)
            return
        # tags such as laion2b_s34b_b79k resolve to a hub download in the reference (pretrained.py:843,880-912)
        raise RuntimeError(f"Pretrained weights ({pretrained}) for model {self.model_name} not found: "
                           "no network access; pass a local state_dict path or pretrained=None")

    @staticmethod
    def _preprocess(img: torch.Tensor) -> torch.Tensor:
        """Tensor-only stand-in for the torchvision pipeline: [3,H,W] float in [0,1] -> OPENAI mean/std normalised."""
        mean = torch.tensor(OPENAI_DATASET_MEAN, dtype=img.dtype).view(3, 1, 1)
        std = torch.tensor(OPENAI_DATASET_STD, dtype=img.dtype).view(3, 1, 1)
        return (img - mean) / std

    def _tokenizer(self, x):
        """Gene towers: the 'tokenizer' passes gene-expression vectors through as a float matrix.  Text tower: token ids
        pass through; strings go through the package's own byte-level BPE tokenizer (tokenizer.py; needs CLIP's merge
        table via ``tokenizer_vocab`` / ``$SC_BPE_VOCAB``)."""
        if self.cfg.gene is not None:
            return torch.as_tensor(x, dtype=torch.float32)
        if isinstance(x, (str, list, tuple)) and (isinstance(x, str) or (len(x) and isinstance(x[0], str))):
            if self._bpe is None:
                from .tokenizer import BpeTokenizer      # raises FileNotFoundError with instructions if no merge table
                self._bpe = BpeTokenizer(self.tokenizer_vocab, self.cfg.text.context_length)
            return self._bpe(x)
        return torch.as_tensor(x, dtype=torch.int64)

    def load_checkpoint_state_dict(self, sd: Dict[str, Any], source: str = "checkpoint") -> Dict[str, List[str]]:
        """Load a reference-style checkpoint: plain ``CLIP.state_dict()``, a Lightning file (``state_dict`` with
        ``net.model.`` prefixes), or a DDP / legacy open_clip file (``module.`` prefix; factory.py load_state_dict
        strips it the same way).  The positional embedding is resized to this model's grid.  For ``*-gene`` models
        only the keys of the reference's text tower may go unused, and every vision-tower key must be found --
        anything else raises instead of silently training from random init."""
        sd = dict(sd.get("state_dict", sd))
        sd = {strip_checkpoint_prefix(k): v for k, v in sd.items() if isinstance(v, torch.Tensor)}
        if self.cfg.vision is not None:
            g = self.cfg.vision.image_size // self.cfg.vision.patch_size
            resize_pos_embed(sd, (g, g))
        if self.cfg.text is not None:           # factory.py:221: the text positions follow the model's context length
            resize_text_pos_embed(sd, self.cfg.text.context_length, model_width=self.cfg.text.width)
        ls = sd.get("logit_scale")              # factory.py:203-204: scalar vs 1-element parameter
        if ls is not None and ls.ndim != 0 and ls.numel() == 1:
            sd["logit_scale"] = ls.reshape(())
        matched = {k: v for k, v in sd.items() if k in self.store.by_name}
        missing = [n for n in self.store.by_name if n not in matched]
        unexpected = [k for k in sd if k not in self.store.by_name]
        if not matched:
            raise RuntimeError(f"{source}: none of its {len(sd)} keys matches this model "
                               f"(first keys: {list(sd)[:3]}); nothing was loaded")
        missing_vision = [n for n in missing if n.startswith("visual.")]
        if missing_vision:
            raise RuntimeError(f"{source}: {len(missing_vision)} vision-tower tensors are missing "
                               f"(e.g. {missing_vision[:3]})")
        second_ok = ("gene.",) if self.cfg.gene is not None else ()
        bad_missing = [n for n in missing if not n.startswith(second_ok)] if second_ok else missing
        if bad_missing:
            raise RuntimeError(f"{source}: missing tensors {bad_missing[:5]} ({len(bad_missing)} in all)")
        if missing or unexpected:
            print(f"[spatial_clip_amd] {source}: loaded {len(matched)} tensors; left at init: {len(missing)} "
                  f"(e.g. {missing[:2]}); unused checkpoint keys: {len(unexpected)} (e.g. {unexpected[:2]})")
        self.store.load_state_dict(matched, strict=False)
        self.reset_fp8_scaling()            # new weights: the delayed e4m3 scales of the old ones are not history for them
        return {"missing": missing, "unexpected": unexpected}

    def state_dict(self, *a, **k) -> Dict[str, torch.Tensor]:
        return self.store.state_dict()

    def parameters(self, recurse: bool = True):
        """nn.Module.parameters for host-side readers: the current stream first waits for an optimiser update that may still be
        running behind the forward on the communication stream (optim.FusedAdamW._step_behind_forward)."""
        self.store.wait_all()
        return super().parameters(recurse)

    def named_parameters(self, *a, **k):
        self.store.wait_all()
        return super().named_parameters(*a, **k)

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict(sd, strict=strict)
        self.reset_fp8_scaling()

    # ------------------------------------------------------------------ forward / backward
    def forward(self, images: torch.Tensor, texts: torch.Tensor) -> Dict[str, torch.Tensor]:
        self._mark_pass(torch.is_grad_enabled())    # read HERE: inside autograd.Function.forward grad mode is always off
        img, txt, s = _NetFn.apply(self.store.params["logit_scale"], self, images, texts)
        return {"image_features": img, "text_features": txt, "logit_scale": s, "logit_bias": None}

    def _mark_pass(self, train_pass: bool) -> None:
        """Tell the stacks whether the forward about to run will be differentiated.  Only such a pass records e4m3 maxima,
        consumes the delayed scales or overwrites the per-block e4m3 copies the backward reads; evaluation forwards --
        SpatialClipNet.forward under no_grad AND every encode_image / encode_text call (the zero-shot bank) -- do none of
        that, so they give the numbers a fresh process gives for the same weights (advisor, rounds 3-4)."""
        for _, stack in self._stacks():
            stack.fp8_train_pass = bool(train_pass)

    def _bucket(self, names: List[str]) -> None:
        if self.grad_bucket_hook is not None:
            self.grad_bucket_hook(*self.store.grad_range(names))

    def _backward(self, d_img, d_txt, d_s) -> None:
        s = self.store
        dev = self.device_
        if d_s is None:
            s.g("logit_scale").zero_()
        else:
            ops.exp_scalar_bwd(self._scale, d_s.reshape(1).float().contiguous(), s.g("logit_scale").view(1), 1.0)
        self._bucket(["logit_scale"])
        B, D = self.second.f.shape
        if d_txt is None:
            d_txt = torch.zeros((B, D), dtype=torch.float32, device=dev)
        if d_img is None:
            d_img = torch.zeros((B, D), dtype=torch.float32, device=dev)
        if self._towers_side_by_side():
            cur = torch.cuda.current_stream(dev)
            ts = streams.tower_stream(dev)
            ts.wait_stream(cur)                     # d_txt and the forward's activations are final on the chain
            with torch.cuda.stream(ts):
                self.second.backward(d_txt, on_bucket=self._bucket)
            self.vision.backward(d_img, on_bucket=self._bucket)
            cur.wait_stream(ts)
            return
        self.second.backward(d_txt, on_bucket=self._bucket)
        self.vision.backward(d_img, on_bucket=self._bucket)
