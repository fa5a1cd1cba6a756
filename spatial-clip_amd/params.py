"""Flat parameter store: every parameter of the model lives in ONE fp32 master buffer (plus one flat gradient
buffer and two flat Adam moment buffers), laid out in forward order so that gradient buckets complete back to
front during backward.  Named ``torch.nn.Parameter`` views (reference ``CLIP.state_dict()`` names:
``visual.conv1.weight``, ``visual.transformer.resblocks.N.attn.in_proj_weight`` ...) keep checkpoint interop.

bf16 compute copies of the GEMM weights (``wf`` = [N_out, K_in(pad)] for forward, ``wb`` = [K_in, N_out] for
dgrad) are refreshed from the master buffer after each optimiser step."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import os

import torch

from . import ops
from .model_configs import ModelCfg

ALIGN = 64  # floats


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    init: str                      # "uniform:<bound>", "normal:<std>", "ones", "zeros", "const:<v>"
    offset: int = 0

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclass
class LinearCopy:
    """bf16 copies of one GEMM weight.  ``stored_kn`` marks matrices the reference stores as [K_in, N_out]
    (visual.proj, text_projection) instead of nn.Linear's [N_out, K_in]."""
    name: str
    n_out: int
    k_in: int
    stored_kn: bool = False
    need_wb: bool = True
    k_pad: int = 0
    wf: Optional[torch.Tensor] = None
    wb: Optional[torch.Tensor] = None
    w8: Optional[torch.Tensor] = None       # fp8 path: e4m3 bytes [N_out, K_in] + per-output-channel dequantisation factors
    w8s: Optional[torch.Tensor] = None
    wb8: Optional[torch.Tensor] = None      # fp8 data-gradient path: e4m3 bytes of wb [K_in, N_out] + per-input-channel factors
    wb8s: Optional[torch.Tensor] = None


def _block_specs(prefix: str, d: int, mlp: int) -> List[ParamSpec]:
    """ResidualAttentionBlock parameters with torch's default initialisers
    (src/open_clip/transformer.py:238-265; nn.MultiheadAttention._reset_parameters)."""
    xav = math.sqrt(6.0 / (3 * d + d))
    return [
        ParamSpec(prefix + "ln_1.weight", (d,), "ones"),
        ParamSpec(prefix + "ln_1.bias", (d,), "zeros"),
        ParamSpec(prefix + "attn.in_proj_weight", (3 * d, d), f"uniform:{xav}"),
        ParamSpec(prefix + "attn.in_proj_bias", (3 * d,), "zeros"),
        ParamSpec(prefix + "attn.out_proj.weight", (d, d), f"uniform:{1.0 / math.sqrt(d)}"),
        ParamSpec(prefix + "attn.out_proj.bias", (d,), "zeros"),
        ParamSpec(prefix + "ln_2.weight", (d,), "ones"),
        ParamSpec(prefix + "ln_2.bias", (d,), "zeros"),
        ParamSpec(prefix + "mlp.c_fc.weight", (mlp, d), f"uniform:{1.0 / math.sqrt(d)}"),
        ParamSpec(prefix + "mlp.c_fc.bias", (mlp,), f"uniform:{1.0 / math.sqrt(d)}"),
        ParamSpec(prefix + "mlp.c_proj.weight", (d, mlp), f"uniform:{1.0 / math.sqrt(mlp)}"),
        ParamSpec(prefix + "mlp.c_proj.bias", (d,), f"uniform:{1.0 / math.sqrt(mlp)}"),
    ]


def build_specs(cfg: ModelCfg) -> List[ParamSpec]:
    """Parameter list in forward order.  Vision: src/open_clip/transformer.py:624-637,706 ; text:
    TextTransformer.init_parameters ; logit_scale: src/open_clip/model.py:297-298 ; gene-MLP: nn.Linear defaults."""
    v = cfg.vision
    d = v.width
    specs: List[ParamSpec] = []
    fan_in = 3 * v.patch_size * v.patch_size
    specs.append(ParamSpec("visual.conv1.weight", (d, 3, v.patch_size, v.patch_size), f"uniform:{1.0 / math.sqrt(fan_in)}"))
    specs.append(ParamSpec("visual.class_embedding", (d,), f"normal:{d ** -0.5}"))
    specs.append(ParamSpec("visual.positional_embedding", (v.tokens, d), f"normal:{d ** -0.5}"))
    specs.append(ParamSpec("visual.ln_pre.weight", (d,), "ones"))
    specs.append(ParamSpec("visual.ln_pre.bias", (d,), "zeros"))
    for i in range(v.layers):
        specs += _block_specs(f"visual.transformer.resblocks.{i}.", d, int(d * v.mlp_ratio))
    specs.append(ParamSpec("visual.ln_post.weight", (d,), "ones"))
    specs.append(ParamSpec("visual.ln_post.bias", (d,), "zeros"))
    specs.append(ParamSpec("visual.proj", (d, cfg.embed_dim), f"normal:{d ** -0.5}"))
    if cfg.text is not None:
        t = cfg.text
        specs.append(ParamSpec("token_embedding.weight", (t.vocab_size, t.width), "normal:0.02"))
        specs.append(ParamSpec("positional_embedding", (t.context_length, t.width), "normal:0.01"))
        for i in range(t.layers):
            specs += _block_specs(f"transformer.resblocks.{i}.", t.width, int(t.width * t.mlp_ratio))
        specs.append(ParamSpec("ln_final.weight", (t.width,), "ones"))
        specs.append(ParamSpec("ln_final.bias", (t.width,), "zeros"))
        specs.append(ParamSpec("text_projection", (t.width, cfg.embed_dim), f"normal:{t.width ** -0.5}"))
    if cfg.gene is not None and cfg.gene.kind == "transformer":
        g = cfg.gene
        gd = g.width
        specs.append(ParamSpec("gene.conv1.weight", (gd, g.patch), f"uniform:{1.0 / math.sqrt(g.patch)}"))
        specs.append(ParamSpec("gene.class_embedding", (gd,), f"normal:{gd ** -0.5}"))
        specs.append(ParamSpec("gene.positional_embedding", (g.tokens, gd), f"normal:{gd ** -0.5}"))
        specs.append(ParamSpec("gene.ln_pre.weight", (gd,), "ones"))
        specs.append(ParamSpec("gene.ln_pre.bias", (gd,), "zeros"))
        for i in range(g.layers):
            specs += _block_specs(f"gene.transformer.resblocks.{i}.", gd, int(gd * g.mlp_ratio))
        specs.append(ParamSpec("gene.ln_post.weight", (gd,), "ones"))
        specs.append(ParamSpec("gene.ln_post.bias", (gd,), "zeros"))
        specs.append(ParamSpec("gene.proj", (gd, cfg.embed_dim), f"normal:{gd ** -0.5}"))
    elif cfg.gene is not None:
        g = cfg.gene
        specs.append(ParamSpec("gene.fc1.weight", (g.hidden, g.n_genes), f"uniform:{1.0 / math.sqrt(g.n_genes)}"))
        specs.append(ParamSpec("gene.fc1.bias", (g.hidden,), f"uniform:{1.0 / math.sqrt(g.n_genes)}"))
        specs.append(ParamSpec("gene.fc2.weight", (cfg.embed_dim, g.hidden), f"uniform:{1.0 / math.sqrt(g.hidden)}"))
        specs.append(ParamSpec("gene.fc2.bias", (cfg.embed_dim,), f"uniform:{1.0 / math.sqrt(g.hidden)}"))
    specs.append(ParamSpec("logit_scale", (), f"const:{cfg.init_logit_scale}"))
    off = 0
    for s in specs:
        s.offset = off
        off += _round_up(max(s.numel, 1), ALIGN)
    return specs


def _dist_world() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class ParamStore:
    def __init__(self, cfg: ModelCfg, device: torch.device, seed: int = 0, fp8: bool = False, shard_world: Optional[int] = None):
        self.cfg = cfg
        self.device = device
        self.fp8 = fp8          # forward GEMMs of the transformer blocks in e4m3 (BASELINE configs[4]); see csrc/sc_fp8.hip
        self.specs = build_specs(cfg)
        self.by_name = {s.name: s for s in self.specs}
        last = self.specs[-1]
        # The flat buffers end on a multiple of ALIGN * W floats (W = ranks of the live process group: the launcher's
        # comm.init_from_env runs before the model is built), so that every gradient bucket of the sharded optimiser splits
        # into W equal, ALIGN-aligned pieces (comm.ShardedGradExchange).  The padding holds zeros and stays zero.
        self.shard_world = int(shard_world or _dist_world())
        self.total = _round_up(last.offset + _round_up(max(last.numel, 1), ALIGN), ALIGN * self.shard_world)
        # weight refreshes still in flight on the communication stream: (lo, hi, event) -- the sharded optimiser all-gathers
        # the updated masters bucket by bucket BEHIND the next forward, which waits per bucket (wait_range)
        self.pending: List[Tuple[int, int, object, set]] = []
        self.master = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        # bf16 mirror of the master buffer (same flat layout), written by the AdamW kernel: the forward operand of
        # every Linear whose K needs no padding is a VIEW of it, so no separate cast launch exists for those
        self.master_bf16 = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self.params: Dict[str, torch.nn.Parameter] = {}
        for s in self.specs:
            view = self.master[s.offset:s.offset + s.numel].view(s.shape)
            p = torch.nn.Parameter(view, requires_grad=True)
            p.grad = self.grad[s.offset:s.offset + s.numel].view(s.shape)
            self.params[s.name] = p
        self.copies: Dict[str, LinearCopy] = {}
        self._build_copies()
        self.init_parameters(seed)

    # ------------------------------------------------------------------ views
    def p(self, name: str) -> torch.Tensor:
        """fp32 master view of one parameter.  If an optimiser update / weight gather is still running behind the forward on
        the communication stream, the CURRENT stream first waits for the part that touches this parameter, so whoever reads
        the view on the current stream (a tower, a test, a user) reads finished weights (advisor, round 5)."""
        s = self.by_name[name]
        if self.pending:
            self.wait_range(s.offset, s.offset + max(s.numel, 1))
        return self.master[s.offset:s.offset + s.numel].view(s.shape)

    def parameters(self) -> List[torch.nn.Parameter]:
        """The nn.Parameter views, for host-side readers: the current stream waits for every in-flight update first."""
        self.wait_all()
        return list(self.params.values())

    def g(self, name: str) -> torch.Tensor:
        s = self.by_name[name]
        return self.grad[s.offset:s.offset + s.numel].view(s.shape)

    # ------------------------------------------------------------------ in-flight weight refreshes
    def wait_range(self, lo: int, hi: int) -> None:
        """Make the current stream wait for every in-flight refresh that touches flat[lo:hi] (no-op when nothing is pending:
        the single-process and all-reduce paths never have anything).  Entries stay in ``pending`` until ``wait_all`` -- the
        two towers may read parameters of one bucket from two different streams (net.py: towers side by side), and each of
        those streams has to wait; an entry remembers the streams that already did."""
        if not self.pending:
            return
        hit, cur = [], None
        for ent in self.pending:
            plo, phi, ev, waited = ent
            if plo < hi and lo < phi:
                cur = cur or torch.cuda.current_stream(self.device)
                if cur.cuda_stream not in waited:
                    waited.add(cur.cuda_stream)
                    hit.append(ev)
        if hit:
            from . import comm          # (bench.py's instrumented pass times this wait: comm.CommProbe)
            comm._stalled("weight refresh behind the forward (optimiser / all-gather)", self.device,
                          lambda: [cur.wait_event(ev) for ev in hit])

    def add_pending(self, lo: int, hi: int, event) -> None:
        """Register an update of flat[lo:hi] that is still running on another stream; ``event`` marks its end."""
        self.pending.append((lo, hi, event, set()))

    def wait_names(self, names: List[str]) -> None:
        if self.pending:
            self.wait_range(*self.grad_range(names))

    def wait_all(self) -> None:
        """The current stream waits for every in-flight update; the list is cleared (callers: the start of a backward, of an
        optimiser step, state_dict / load_state_dict -- points behind which every stream of the step is ordered after the
        current one)."""
        if self.pending:
            self.wait_range(0, self.total)
            self.pending = []

    def grad_range(self, names: List[str]) -> Tuple[int, int]:
        lo = min(self.by_name[n].offset for n in names)
        hi = max(self.by_name[n].offset + _round_up(max(self.by_name[n].numel, 1), ALIGN) for n in names)
        return lo, hi

    def num_parameters(self) -> int:
        return sum(s.numel for s in self.specs)

    # ------------------------------------------------------------------ init / state dict
    @torch.no_grad()
    def init_parameters(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        for s in self.specs:
            kind, _, arg = s.init.partition(":")
            shape = s.shape if s.shape else (1,)
            if kind == "uniform":
                t = (torch.rand(shape, generator=g) * 2 - 1) * float(arg)
            elif kind == "normal":
                t = torch.randn(shape, generator=g) * float(arg)
            elif kind == "ones":
                t = torch.ones(shape)
            elif kind == "zeros":
                t = torch.zeros(shape)
            else:
                t = torch.full(shape, float(arg))
            self.p(s.name).copy_(t.view(s.shape).to(self.device))
        self.refresh_compute_copies()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        self.wait_all()
        return {s.name: self.p(s.name).detach().clone() for s in self.specs}

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        self.wait_all()
        missing = [s.name for s in self.specs if s.name not in sd]
        unexpected = [k for k in sd if k not in self.by_name]
        if strict and (missing or unexpected):
            raise KeyError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
        for s in self.specs:
            if s.name in sd:
                t = sd[s.name]
                if tuple(t.shape) != tuple(s.shape):
                    raise ValueError(f"{s.name}: shape {tuple(t.shape)} != {tuple(s.shape)}")
                self.p(s.name).copy_(t.to(self.device, torch.float32))
        self.refresh_compute_copies()

    # ------------------------------------------------------------------ bf16 compute copies
    def _add_copy(self, name: str, n_out: int, k_in: int, stored_kn: bool = False, need_wb: bool = True) -> None:
        c = LinearCopy(name, n_out, k_in, stored_kn, need_wb, k_pad=_round_up(k_in, 64))
        sp = self.by_name[name]
        mirror = self.master_bf16[sp.offset:sp.offset + sp.numel]
        if not stored_kn and c.k_pad == k_in:
            c.wf = mirror.view(n_out, k_in)                  # view of the bf16 mirror: refreshed by AdamW itself
            c.wf_is_view = True
        else:
            c.wf = torch.zeros((n_out, c.k_pad), dtype=torch.bfloat16, device=self.device)
            c.wf_is_view = False
        if need_wb:
            if stored_kn:
                c.wb = mirror.view(k_in, n_out)              # [K_in, N_out] as stored: again a view of the mirror
            else:
                c.wb = torch.zeros((k_in, n_out), dtype=torch.bfloat16, device=self.device)
        self.copies[name] = c

    def _build_copies(self) -> None:
        cfg = self.cfg
        v = cfg.vision
        d = v.width
        self._add_copy("visual.conv1.weight", d, 3 * v.patch_size * v.patch_size, need_wb=False)

        def block(prefix: str, dd: int, mlp: int) -> None:
            self._add_copy(prefix + "attn.in_proj_weight", 3 * dd, dd)
            self._add_copy(prefix + "attn.out_proj.weight", dd, dd)
            self._add_copy(prefix + "mlp.c_fc.weight", mlp, dd)
            self._add_copy(prefix + "mlp.c_proj.weight", dd, mlp)
            if self.fp8:
                if dd % 128 or mlp % 128:
                    raise ValueError(f"fp8 path needs block widths that are multiples of 128 (got {dd} / {mlp})")
                # e4m3 operands exist only where the quantiser of the OTHER operand is fused into the kernel that
                # produces it (towers.TransformerStack): forward qkv / c_fc (A = LayerNorm output), data gradients of
                # c_proj / out_proj (A = the residual gradient LayerNorm backward emits; B = the transposed weight)
                # c_proj forward / c_fc data gradient: A = the e4m3 copy the GELU / GELU' epilogues emit with the previous
                # step's per-tensor scale (delayed scaling, sc_fp8_scale_update)
                for leaf, n_out, k_in in (("attn.in_proj_weight", 3 * dd, dd), ("mlp.c_fc.weight", mlp, dd),
                                          ("mlp.c_proj.weight", dd, mlp)):
                    c = self.copies[prefix + leaf]
                    c.w8 = torch.zeros((n_out, k_in), dtype=torch.uint8, device=self.device)
                    c.w8s = torch.ones(n_out, dtype=torch.float32, device=self.device)
                for leaf, n_out, k_in in (("attn.out_proj.weight", dd, dd), ("mlp.c_proj.weight", dd, mlp),
                                          ("mlp.c_fc.weight", mlp, dd)):
                    c = self.copies[prefix + leaf]
                    c.wb8 = torch.zeros((k_in, n_out), dtype=torch.uint8, device=self.device)
                    c.wb8s = torch.ones(k_in, dtype=torch.float32, device=self.device)

        for i in range(v.layers):
            block(f"visual.transformer.resblocks.{i}.", d, int(d * v.mlp_ratio))
        self._add_copy("visual.proj", cfg.embed_dim, d, stored_kn=True)
        if cfg.text is not None:
            t = cfg.text
            for i in range(t.layers):
                block(f"transformer.resblocks.{i}.", t.width, int(t.width * t.mlp_ratio))
            self._add_copy("text_projection", cfg.embed_dim, t.width, stored_kn=True)
        if cfg.gene is not None and cfg.gene.kind == "transformer":
            g = cfg.gene
            self._add_copy("gene.conv1.weight", g.width, g.patch, need_wb=False)
            for i in range(g.layers):
                block(f"gene.transformer.resblocks.{i}.", g.width, int(g.width * g.mlp_ratio))
            self._add_copy("gene.proj", cfg.embed_dim, g.width, stored_kn=True)
        elif cfg.gene is not None:
            g = cfg.gene
            self._add_copy("gene.fc1.weight", g.hidden, g.n_genes, need_wb=False)
            self._add_copy("gene.fc2.weight", cfg.embed_dim, g.hidden)

    def _transpose_plan(self, copies) -> Optional[Tuple[torch.Tensor, torch.Tensor, int, int]]:
        """Descriptor table of the transposed copies (wb of nn.Linear weights, wf of [K,N]-stored projections) of ``copies``:
        (desc, tile prefix, entries, tiles) for ops.cast_transpose_batched, or None if there is nothing to transpose."""
        desc, prefix, tiles = [], [0], 0
        for c in copies:
            sp = self.by_name[c.name]
            if c.stored_kn:       # wf[n][k] = src[k][n] : src is [k_in, n_out]
                items = [(sp.offset, c.wf, c.k_in, c.n_out, c.k_pad)]
            elif c.wb is not None:  # wb[k][n] = src[n][k] : src is [n_out, k_in]
                items = [(sp.offset, c.wb, c.n_out, c.k_in, c.n_out)]
            else:
                items = []
            for off, dst, rows, cols, ldd in items:
                desc.append([off, dst.data_ptr(), rows, cols, ldd])
                tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
                prefix.append(tiles)
        if not desc:
            return None
        return (torch.tensor(desc, dtype=torch.int64, device=self.device),
                torch.tensor(prefix, dtype=torch.int32, device=self.device), len(desc), tiles)

    def _q8_plan(self, copies):
        """Device tables of ops.quantize_rows_fp8_batched for the e4m3 copies of ``copies`` (cached per set of weights: every
        pointer in it belongs to a persistent buffer)."""
        key = tuple(c.name for c in copies)
        cache = self.__dict__.setdefault("_q8_plans", {})
        if key not in cache:
            desc, prefix, blocks = [], [0], 0
            for c in copies:
                items = []
                if c.w8 is not None:
                    src = self.p(c.name).view(c.n_out, c.k_in)
                    items.append((src, 1, c.w8, c.w8s))
                if c.wb8 is not None:
                    items.append((c.wb, 0, c.wb8, c.wb8s))
                for src, f32, dst, sinv in items:
                    rows, cols = src.shape
                    if cols % 8 or dst.stride(0) % 8 or src.stride(0) % (4 if f32 else 8) or src.data_ptr() % 16 or dst.data_ptr() % 8:
                        raise ValueError(f"e4m3 copy of {c.name}: shape / alignment outside sc_quantize_rows_fp8's rules")
                    desc.append([src.data_ptr(), f32, src.stride(0), rows, cols, dst.data_ptr(), dst.stride(0), sinv.data_ptr()])
                    blocks += (rows + 3) // 4
                    prefix.append(blocks)
            cache[key] = None if not desc else (torch.tensor(desc, dtype=torch.int64, device=self.device),
                                                torch.tensor(prefix, dtype=torch.int32, device=self.device), len(desc), blocks)
        return cache[key]

    def _refresh_copies(self, copies, plan) -> None:
        """Everything derived from the (fresh) bf16 mirror / fp32 masters for the Linear weights in ``copies``."""
        if plan is not None:
            desc, prefix, n, tiles = plan
            ops.cast_transpose_batched(self.master, desc, prefix, n, tiles, mirror_bf16=self.master_bf16)
        if self.fp8:                                      # per-output-channel e4m3 copies straight from the fp32 masters
            q8 = self._q8_plan(copies)                    # ... and of the transposed bf16 copies (rows = input channels): ONE launch
            if q8 is not None and os.environ.get("SC_Q8_BATCH", "1") != "0":
                ops.quantize_rows_fp8_batched(*q8)
            else:                                         # A/B: one launch per matrix (rounds 2-4; same bits)
                for c in copies:
                    if c.w8 is not None:
                        ops.quantize_rows_fp8(self.p(c.name).view(c.n_out, c.k_in), c.w8, c.w8s)
                    if c.wb8 is not None:
                        ops.quantize_rows_fp8(c.wb, c.wb8, c.wb8s)
        for c in copies:
            if not c.stored_kn and not c.wf_is_view:      # K-padded forward operand (gene.fc1, conv1 at patch 14)
                ops.cast_pad_bf16(self.p(c.name).view(c.n_out, c.k_in), c.wf, c.n_out, c.k_in, c.k_pad,
                                  ld_src=c.k_in, ld_dst=c.k_pad)

    def refresh_compute_copies(self, mirror_is_fresh: bool = False) -> None:
        """fp32 master -> bf16 GEMM operands (after init, load_state_dict and every optimiser step).
        ``mirror_is_fresh``: the AdamW kernel has just written the bf16 mirror, only padded / transposed copies remain."""
        self.wait_all()
        if not mirror_is_fresh:
            ops.cast_pad_bf16(self.master.view(1, -1), self.master_bf16.view(1, -1), 1, self.total, self.total)
        if getattr(self, "_tp_plan", False) is False:
            self._tp_plan = self._transpose_plan(list(self.copies.values()))
        self._refresh_copies(list(self.copies.values()), self._tp_plan)

    def copies_ending_in(self, lo: int, hi: int):
        """The Linear weights whose LAST element lies in flat[lo:hi] (each weight belongs to exactly one such range)."""
        out = []
        for c in self.copies.values():
            sp = self.by_name[c.name]
            if lo <= sp.offset + sp.numel - 1 < hi:
                out.append(c)
        return out

    def refresh_range(self, lo: int, hi: int, copies, plan, fresh=None) -> None:
        """The same refresh for ONE range of the flat master buffer whose fp32 values have just arrived (all-gather of the
        sharded optimiser): bf16 mirror of flat[lo:hi] -- except ``fresh`` = (a, b), the piece this rank's AdamW kernel has
        just written itself, mirror included -- then the derived copies of ``copies``: the weights that are complete once
        this range is (the caller passes, per range, the weights that END in the latest-arriving range they touch)."""
        a, b = fresh if fresh is not None else (lo, lo)
        for x, y in ((lo, a), (b, hi)):
            if y > x:         # rows of 64 floats: the flat buffers are 64-float aligned everywhere
                ops.cast_pad_bf16(self.master[x:y].view(-1, 64), self.master_bf16[x:y].view(-1, 64), (y - x) // 64, 64, 64)
        self._refresh_copies(copies, plan)
