"""Encoder towers as explicit forward / backward kernel sequences over pre-allocated HBM buffers.

No autograd graph, no tracing compiler: every step is a C-ABI kernel launch on the current HIP stream, the
activations needed by backward live in buffers sized once per batch shape (288 GB of HBM3E make activation
recomputation unnecessary at these sizes), parameter gradients are written straight into the flat fp32 gradient
buffer of :class:`ParamStore`.

Reference semantics: VisionTransformer.forward (src/open_clip/transformer.py:783-823,907-918),
ResidualAttentionBlock.forward (:289-300), CLIP.encode_image (src/open_clip/model.py:326-328).
Residual stream fp32, GEMM operands bf16 with fp32 accumulation, LayerNorm / softmax / normalisation in fp32 --
the reference's ``precision: bf16-mixed`` autocast policy."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .model_configs import ModelCfg
from .params import ParamStore

BF16 = torch.bfloat16
F32 = torch.float32


def _splitk_for(m_out: int, n_out: int, k: int) -> int:
    """Split-K factor for a weight-gradient GEMM: ~one 256x256 workgroup per CU (256 CUs)."""
    tiles = ((m_out + 255) // 256) * ((n_out + 255) // 256)
    ktiles = (k + 63) // 64
    want = max(1, min((256 + tiles // 2) // max(tiles, 1), ktiles // 4))
    return max(1, min(want, 32))


def _splitk_for_group(shapes, k: int, one_round: bool = False) -> int:
    """Common split-K factor of a grouped weight-gradient launch: the smallest one that fills >= 95 % of whole rounds of
    256 workgroups (256 CUs, one 256x256 tile each); ``one_round``: the largest one that stays within ONE round (the
    weight gradients run beside the data-gradient chain: a launch of several rounds holds the chain's kernels back)."""
    tiles = sum(((m + 255) // 256) * ((n + 255) // 256) for m, n in shapes)
    ktiles = (k + 63) // 64
    best = 1
    for s in range(1, max(1, min(32, ktiles // 4)) + 1):
        wg = tiles * s
        if wg <= 256:
            best = s
        if not one_round and wg / (-(-wg // 256) * 256) >= 0.95:
            return s
    return best


MAX_SIDE_KTILES = 200       # longest K chain (64-token tiles) a grouped weight-gradient workgroup may own


def _wgrad_group_mode() -> str:
    """Read at every backward: SC_WGRAD_GROUP = auto (default), 0 (one launch per Linear, rounds 1-3), 1 (attention branch
    grouped), 2 / 3 (both branches grouped, whole rounds / one round), 4 (all four of a block in one launch) -- A/B switch.
    ``auto`` groups a branch's two weight gradients when the grouped launch fits ONE round of workgroups with K chains of at
    most MAX_SIDE_KTILES tiles: the weight gradients run beside the data-gradient chain, and long-lived or multi-round side
    launches hold the chain's kernels back.  Same-box A/B (profiles/r04_wgrad_group_ab.txt): ViT-B/16 -0.06 ... -0.16 ms per
    step with the attention branch grouped (113-tile chains); the MLP pair at one round (263-tile chains) -0.04 more on one
    box; ViT-L/14 (257- / 514-tile chains) +2.4 ms bf16, +3.9 ms e4m3 with both grouped, +1.3 ms e4m3 with the attention
    branch alone -- hence the cap."""
    return os.environ.get("SC_WGRAD_GROUP", "auto")


def _group_ok(mode: str, branch: str, shapes, k: int) -> bool:
    if mode == "auto":
        s = _splitk_for_group(shapes, k, one_round=True)
        return ((k + 63) // 64 + s - 1) // s <= MAX_SIDE_KTILES
    return mode in ("2", "3") or (mode == "1" and branch == "attn")


class _Bufs:
    """Named device buffers, (re)allocated when the requested shape changes."""

    def __init__(self, device):
        self.device = device
        self._b: Dict[str, torch.Tensor] = {}

    def get(self, name: str, shape, dtype) -> torch.Tensor:
        t = self._b.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._b[name] = t
        return t

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._b.values())


class TransformerStack:
    """N pre-LN residual attention blocks (ResidualAttentionBlock, transformer.py:238-300)."""

    def __init__(self, store: ParamStore, prefix: str, width: int, heads: int, layers: int, mlp: int, causal: bool,
                 cls_only_last: bool = False, res16_ok: bool = False, quick_gelu: bool = False):
        # cls_only_last: only token 0 of the last block's output is consumed downstream (ViT pool 'tok'), so that
        # block computes K/V for all tokens but attention output, out_proj and the MLP for the CLS rows only -- the
        # values the reference would compute for the other 196 rows are dead.
        self.cls_only_last = cls_only_last and os.environ.get("SC_CLS_ONLY", "1") != "0"
        self.s, self.prefix = store, prefix
        self.d, self.H, self.layers, self.mlp, self.causal = width, heads, layers, mlp, causal
        self.dh = width // heads
        self.bufs = _Bufs(store.device)
        # act_layer = QuickGELU for towers built from a `quick_gelu: true` config (OpenAI-pretrained weights,
        # src/open_clip/model.py:142-145,228): the three GELU epilogues with x * sigmoid(1.702 x) and its derivative
        self.quick_gelu = bool(quick_gelu)
        self.epi_pair, self.epi_grad_pair, self.epi_dgelu = ops.act_epilogues(self.quick_gelu)
        self.fp8 = bool(getattr(store, "fp8", False))
        # residual stream in bf16 (what the reference's autocast keeps; SC_RES_STREAM, read at every forward) for the towers
        # whose stem / head kernels take it (the patch towers); off: fp32 stream
        self.res16_ok = res16_ok
        self.res_stream = "bf16"
        self.r16 = False
        # activation recomputation (set_grad_checkpointing): the LayerNorm outputs and the GELU output of a block are not
        # kept for the backward (12 of the 36 d bytes a token saves per block); the backward rebuilds them, bit-identically,
        # from the saved residual stream / pre-activation right before the weight-gradient GEMMs that read them
        self.recompute = False
        # fp8 delayed scaling (h = GELU output -> c_proj forward, dU -> c_fc data gradient): per-tensor scales of the
        # previous step; tensor 2 i = h of block i, 2 i + 1 = dU of block i.  Not "ready" until one backward has run.
        self._dq_ready = False
        self._dq_on = os.environ.get("SC_FP8_DELAYED", "1") != "0"      # A/B switch: 0 keeps h / dU consumers in bf16
        # a forward whose backward will run (grad mode on: set by SpatialClipNet.forward) records maxima and may consume the
        # delayed scales; evaluation forwards (validation, test, zero-shot bank) do neither: they run the h consumer in bf16, so
        # a fresh eval process and an in-fit validation of the same weights give the same numbers (advisor, round 3)
        self.fp8_train_pass = False     # set per call by SpatialClipNet.forward; encode_image / encode_text never train
        self._dq_step = 0
        # e4m3 weight gradients of the MLP pair (round 4, sc_gemm_wgrad_fp8): their operands need ONE scale per tensor -- h and
        # dU have such copies already (entries 2 i, 2 i + 1); the LayerNorm kernels add per-tensor copies of a2 = ln_2(x) and of
        # the residual gradient g0 that enters the block's MLP branch (entries 2 L + 2 i, 2 L + 2 i + 1).  SC_FP8_WGRAD=0: A/B.
        self._w8_on = os.environ.get("SC_FP8_WGRAD", "1") != "0"
        self._fwd_w8 = False
        if self.fp8:
            n = 4 * layers
            self._dq_scale = torch.zeros(n, dtype=F32, device=store.device)
            self._dq_scale_inv = torch.ones(n, dtype=F32, device=store.device)
            self._dq_amax = torch.zeros((n, 64), dtype=F32, device=store.device)
            self._dq_hist = torch.zeros((self.FP8_AMAX_HISTORY, n), dtype=F32, device=store.device)

    FP8_AMAX_HISTORY = 4        # steps of per-tensor maxima the delayed scales are taken over

    def set_grad_checkpointing(self, enable: bool = True) -> None:
        self.recompute = bool(enable)

    def reset_fp8_scaling(self) -> None:
        """Forget the delayed-scaling history (the next step runs the h / dU consumers in bf16 and records fresh maxima):
        the fp8 path is a function of (weights, batch, scales of the previous steps).  Called whenever the weights are
        replaced (load_state_dict, checkpoint load without saved scales)."""
        self._dq_ready = False
        self._dq_step = 0
        if self.fp8:
            self._dq_scale.zero_()
            self._dq_scale_inv.fill_(1.0)
            self._dq_amax.zero_()
            self._dq_hist.zero_()

    def fp8_scaling_state(self) -> Optional[Dict[str, object]]:
        """The delayed-scaling state as host tensors (checkpointed beside the weights: a resumed run continues with the
        scales it stopped with), or None off the fp8 path."""
        if not self.fp8:
            return None
        return {"scale": self._dq_scale.cpu(), "scale_inv": self._dq_scale_inv.cpu(), "hist": self._dq_hist.cpu(),
                "ready": bool(self._dq_ready), "step": int(self._dq_step)}

    def load_fp8_scaling_state(self, st: Optional[Dict[str, object]]) -> None:
        self.reset_fp8_scaling()
        if not self.fp8 or not isinstance(st, dict):
            return
        want = {"scale": self._dq_scale, "scale_inv": self._dq_scale_inv, "hist": self._dq_hist}
        for k, dst in want.items():          # a malformed / foreign checkpoint entry resets the history instead of raising
            t = st.get(k)
            if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(dst.shape):
                return
        for k, dst in want.items():
            dst.copy_(st[k])
        self._dq_ready, self._dq_step = bool(st.get("ready", False)), int(st.get("step", 0))

    def _act(self, kind: str, i: int, shape) -> torch.Tensor:
        """Buffer of a block's recomputable activation (a1 / a2 / h): one per block, or two rotating ones in
        recomputation mode (the CLS-only last block keeps its own: it runs first in the backward, nothing to rebuild)."""
        if self.recompute and not (self.cls_only_last and i == self.layers - 1):
            return self.bufs.get(f"{kind}.rc{i & 1}", shape, BF16)
        return self.bufs.get(f"{kind}.{i}", shape, BF16)

    def _linear_fwd(self, epi: int, x: torch.Tensor, name: str, out: torch.Tensor, *, M: int, N: int, K: int, q8=None,
                    **kw):
        """One forward Linear of a full-width block: the bf16 MFMA GEMM, or -- fp8 path, when the producer of ``x`` has
        also emitted its e4m3 copy ``q8 = (bytes, scale_inv)`` and the weight has an e4m3 copy -- the fp8 MFMA GEMM."""
        cp = self.s.copies[name]
        if q8 is None or cp.w8 is None:
            return ops.gemm(ops.NT, epi, x, cp.wf, out, M=M, N=N, K=K, **kw)
        return ops.gemm_fp8(epi, q8[0], q8[1], cp.w8, cp.w8s, out, M=M, N=N, K=K, **kw)

    def _zero_bias(self, n: int) -> torch.Tensor:
        z = self.bufs.get("zero.bias", (n,), F32)
        if not getattr(self, "_zero_bias_set", False):
            z.zero_()
            self._zero_bias_set = True
        return z

    def _q_dead_ok(self, qa, i: int) -> bool:
        """May the last block skip the q projection of the rows whose attention output nobody reads?  bf16 operands only (the
        e4m3 GEMMs take whole-matrix scales), unpadded K; ``SC_CLS_Q=0`` restores the full projection."""
        cp = self.s.copies[self._n(i, "attn.in_proj_weight")]
        return (qa is None or cp.w8 is None) and cp.wf.shape[1] == self.d and self.d % 64 == 0 and \
            os.environ.get("SC_CLS_Q", "1") != "0"

    def _q8(self, tag: str, M: int, K: int):
        """(e4m3 bytes [M, K], scale_inv [M]) scratch for a fused quantiser output, or None off the fp8 path."""
        if not self.fp8:
            return None
        return self.bufs.get(f"q8.{tag}", (M, K), torch.uint8), self.bufs.get(f"q8s.{tag}", (M,), F32)

    def _n(self, i: int, leaf: str) -> str:
        return f"{self.prefix}{i}.{leaf}"

    def layer_param_names(self, i: int) -> List[str]:
        leaves = ["ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                  "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                  "mlp.c_proj.weight", "mlp.c_proj.bias"]
        return [self._n(i, l) for l in leaves]

    # -------------------------------------------------------------------------------- forward
    def forward(self, x0: torch.Tensor, B: int, L: int) -> torch.Tensor:
        s, d, H, dh, mlp = self.s, self.d, self.H, self.dh, self.mlp
        M = B * L
        self.B, self.L, self.M = B, L, M
        bf = self.bufs
        self.r16 = r16 = self.res16_ok and _res_stream_bf16(self.res_stream)
        XD = BF16 if r16 else F32                     # dtype of the residual stream between the blocks
        epi_res = ops.EPI_BF16_BIAS_RES if r16 else ops.EPI_F32_BIAS_RES
        x = x0
        if r16 and x0.dtype != BF16:                  # a stem that writes fp32 (the text tower's embedding): one cast pass per step
            x = ops.cast_pad_bf16(x0, bf.get("x0.16", (M, d), BF16), M, d, d)
        # e4m3 MLP weight gradients: only for a pass that will be differentiated, on the bf16 stream (the per-tensor LayerNorm
        # copies exist for bf16 rows), token count a multiple of the 128-token K tile.  With activation recomputation the e4m3
        # copies of h and a2 are KEPT per block (1 + 0.25 bytes per MLP element instead of the 2 + 0.5 of the bf16 tensors that
        # recomputation drops), and the backward then has nothing to rebuild for the MLP branch
        use_w8 = self._fwd_w8 = bool(self.fp8 and self._dq_on and self._w8_on and self.fp8_train_pass and r16
                                     and M % 128 == 0 and d % 16 == 0 and mlp % 16 == 0 and d >= 256 and mlp >= 256)
        self.x_in = [None] * self.layers
        for i in range(self.layers):
            s.wait_names(self.layer_param_names(i))      # sharded optimiser: this block's refreshed weights have landed (no-op otherwise)
            self.x_in[i] = x
            a1 = self._act("a1", i, (M, d))
            m1 = bf.get(f"m1.{i}", (M,), F32)
            r1 = bf.get(f"r1.{i}", (M,), F32)
            qa = self._q8("a", M, d)            # fp8: LayerNorm also emits the e4m3 copy + row scales (rotating scratch)
            ops.layernorm_fwd(x, s.p(self._n(i, "ln_1.weight")), s.p(self._n(i, "ln_1.bias")), a1, m1, r1, M, d,
                              q8=qa and qa[0], q8_scale_inv=qa and qa[1])
            qkv = bf.get(f"qkv.{i}", (M, 3 * d), BF16)
            last_cls = self.cls_only_last and i == self.layers - 1
            self._q_cls_only = last_cls and self._q_dead_ok(qa, i)
            if self._q_cls_only:
                # last block, only the class token is consumed: K and V of every token, but Q of the class tokens alone -- the
                # other rows' q (a third of this GEMM, of its data gradient and of its weight gradient) is never read
                wq = s.copies[self._n(i, "attn.in_proj_weight")].wf
                bq = s.p(self._n(i, "attn.in_proj_bias"))
                ops.gemm(ops.NT, ops.EPI_BF16_BIAS, a1, wq[d:], qkv[:, d:], M=M, N=2 * d, K=d, bias=bq[d:])
                ops.gemm(ops.NT, ops.EPI_BF16_BIAS, a1.view(B, L * d)[:, :d], wq[:d], qkv.view(B, L * 3 * d)[:, :d],
                         M=B, N=d, K=d, bias=bq[:d])
            else:
                self._linear_fwd(ops.EPI_BF16_BIAS, a1, self._n(i, "attn.in_proj_weight"), qkv,
                                 M=M, N=3 * d, K=d, bias=s.p(self._n(i, "attn.in_proj_bias")), q8=qa)
            o = bf.get(f"o.{i}", (M, d), BF16)
            lse = bf.get(f"lse.{i}", (B, H, L), F32)
            if last_cls:
                return self._forward_last_cls(i, x, qkv, o, lse)
            ops.attn_fwd(qkv, B, L, H, dh, self.causal, out=o, lse=lse)
            xmid = bf.get(f"xmid.{i}", (M, d), XD)
            self._linear_fwd(epi_res, o, self._n(i, "attn.out_proj.weight"), xmid,
                             M=M, N=d, K=d, bias=s.p(self._n(i, "attn.out_proj.bias")), res=x)
            a2 = self._act("a2", i, (M, d))
            m2 = bf.get(f"m2.{i}", (M,), F32)
            r2 = bf.get(f"r2.{i}", (M,), F32)
            t8a = None
            if use_w8:      # per-tensor e4m3 copy of a2, kept per block: X operand of the e4m3 c_fc weight gradient
                ia = 2 * self.layers + 2 * i
                t8a = (bf.get(f"t8.a2.{i}", (M, d), torch.uint8), self._dq_scale[ia:ia + 1], self._dq_amax[ia])
            ops.layernorm_fwd(xmid, s.p(self._n(i, "ln_2.weight")), s.p(self._n(i, "ln_2.bias")), a2, m2, r2, M, d,
                              q8=qa and qa[0], q8_scale_inv=qa and qa[1], t8=t8a)
            # the MLP's backward needs gelu'(u), not u: the default path stores that factor (the forward epilogue holds the
            # erf pieces anyway) and the c_proj data gradient becomes one multiply; recomputation mode keeps u, from which
            # it rebuilds h = gelu(u)
            u = bf.get(f"u.{i}", (M, mlp), BF16)
            h = self._act("h", i, (M, mlp))
            epi_gelu = self.epi_pair if self.recompute else self.epi_grad_pair
            self._u_holds_grad = not self.recompute        # what THIS forward left in the u buffers (read by backward)
            hq = None
            if self.fp8 and self._dq_on and self.fp8_train_pass:      # the GELU epilogue also emits e4m3(h) with last step's scale + records max|h|
                h8 = bf.get(f"q8.h.{i}" if use_w8 else "q8.h", (M, mlp), torch.uint8)     # kept per block when the weight gradient reads it
                hq = dict(q8_out=h8, q8_scale=self._dq_scale[2 * i:2 * i + 1], q8_amax=self._dq_amax[2 * i])
            self._linear_fwd(epi_gelu, a2, self._n(i, "mlp.c_fc.weight"), u,
                             M=M, N=mlp, K=d, bias=s.p(self._n(i, "mlp.c_fc.bias")), out2=h, q8=qa, **(hq or {}))
            xo = bf.get(f"xout.{i}", (M, d), XD)
            cpj = s.copies[self._n(i, "mlp.c_proj.weight")]
            if hq is not None and self._dq_ready and cpj.w8 is not None:
                ops.gemm_fp8(epi_res, h8, self._dq_scale_inv[2 * i:2 * i + 1], cpj.w8, cpj.w8s, xo, M=M, N=d,
                             K=mlp, bias=s.p(self._n(i, "mlp.c_proj.bias")), res=xmid, a_scale_scalar=True)
            else:
                self._linear_fwd(epi_res, h, self._n(i, "mlp.c_proj.weight"), xo,
                                 M=M, N=d, K=mlp, bias=s.p(self._n(i, "mlp.c_proj.bias")), res=xmid)
            x = xo
        return x

    def _forward_last_cls(self, i: int, x: torch.Tensor, qkv: torch.Tensor, o: torch.Tensor, lse: torch.Tensor):
        """Last block, CLS rows only (compact [B, d] tensors; the CLS rows of full tensors are strided views)."""
        s, d, H, dh, mlp, B, L = self.s, self.d, self.H, self.dh, self.mlp, self.B, self.L
        bf = self.bufs
        ops.attn_fwd(qkv, B, L, H, dh, self.causal, out=o, lse=lse, q_rows=1)
        o_c = o.view(B, L * d)[:, :d]
        x_c = x.view(B, L * d)[:, :d]
        XD = BF16 if self.r16 else F32
        epi_res = ops.EPI_BF16_BIAS_RES if self.r16 else ops.EPI_F32_BIAS_RES
        xmid = bf.get("c.xmid", (B, d), XD)
        ops.gemm(ops.NT, epi_res, o_c, s.copies[self._n(i, "attn.out_proj.weight")].wf, xmid,
                 M=B, N=d, K=d, bias=s.p(self._n(i, "attn.out_proj.bias")), res=x_c)
        a2 = bf.get("c.a2", (B, d), BF16)
        ops.layernorm_fwd(xmid, s.p(self._n(i, "ln_2.weight")), s.p(self._n(i, "ln_2.bias")), a2,
                          bf.get("c.m2", (B,), F32), bf.get("c.r2", (B,), F32), B, d)
        u, h = bf.get("c.u", (B, mlp), BF16), bf.get("c.h", (B, mlp), BF16)
        ops.gemm(ops.NT, self.epi_grad_pair, a2, s.copies[self._n(i, "mlp.c_fc.weight")].wf, u, M=B, N=mlp, K=d,
                 bias=s.p(self._n(i, "mlp.c_fc.bias")), out2=h)           # "u" holds gelu'(u): this block is never recomputed
        xo = bf.get("c.xout", (B, d), XD)
        ops.gemm(ops.NT, epi_res, h, s.copies[self._n(i, "mlp.c_proj.weight")].wf, xo, M=B, N=d, K=mlp,
                 bias=s.p(self._n(i, "mlp.c_proj.bias")), res=xmid)
        return xo

    def _backward_last_cls(self, dres: torch.Tensor, dres_c_bf: torch.Tensor, on_side) -> tuple:
        """Backward of the CLS-only last block.  In: dL/d(x_out[CLS]) in the class-token rows of the FULL fp32 buffer
        ``dres`` [M, d] (the other rows hold nothing yet) + its compact bf16 copy.  Out: the same fp32 buffer and a full
        bf16 buffer holding dL/d(block input) for every row."""
        s, d, H, dh, mlp, B, L, M = self.s, self.d, self.H, self.dh, self.mlp, self.B, self.L, self.M
        bf = self.bufs
        i = self.layers - 1
        g = lambda leaf: s.g(self._n(i, leaf))
        cp = lambda leaf: s.copies[self._n(i, leaf)]
        a1, qkv, o = bf.get(f"a1.{i}", (M, d), BF16), bf.get(f"qkv.{i}", (M, 3 * d), BF16), bf.get(f"o.{i}", (M, d), BF16)
        lse = bf.get(f"lse.{i}", (B, H, L), F32)
        a2, u, h, xmid = bf.get("c.a2", (B, d), BF16), bf.get("c.u", (B, mlp), BF16), bf.get("c.h", (B, mlp), BF16), \
            bf.get("c.xmid", (B, d), BF16 if self.r16 else F32)
        dU = bf.get("c.dU", (B, mlp), BF16)
        dA_c = bf.get("c.dA", (B, d), BF16)
        dres_c_bf_ = dres_c_bf
        ops.gemm(ops.NT, ops.EPI_BF16_MUL_AUX, dres_c_bf_, cp("mlp.c_proj.weight").wb, dU, M=B, N=mlp, K=d, aux=u)

        def w_mlp():
            ops.gemm(ops.TN, ops.EPI_F32, dres_c_bf_, h, g("mlp.c_proj.weight"), M=d, N=mlp, K=B)
            ops.gemm_wgrad_bias(dU, a2, g("mlp.c_fc.weight"), g("mlp.c_fc.bias"), M=mlp, N=d, K=B)
        on_side(w_mlp, ())
        ops.gemm(ops.NT, ops.EPI_BF16, dU, cp("mlp.c_fc.weight").wb, dA_c, M=B, N=d, K=mlp)
        g1_c = bf.get("c.dres_bf1", (B, d), BF16)
        ops.layernorm_bwd(dA_c, xmid, bf.get("c.m2", (B,), F32), bf.get("c.r2", (B,), F32),
                          s.p(self._n(i, "ln_2.weight")), dres, g1_c, g("ln_2.weight"), g("ln_2.bias"),
                          g("attn.out_proj.bias"), B, d, accumulate=True, lddres=L * d)      # in place on the class-token rows
        # attention branch: only the CLS rows of dO exist (written through a strided view); sc_attn_bwd with q_rows = 1
        # reads nothing else of dO / o and writes ALL of dqkv (zeros for the dq rows nobody consumed): no memsets
        dO = bf.get("dO", (M, d), BF16)
        ops.gemm(ops.NT, ops.EPI_BF16, g1_c, cp("attn.out_proj.weight").wb, dO.view(B, L * d)[:, :d], M=B, N=d, K=d)
        o_c = o.view(B, L * d)[:, :d]
        dqkv = bf.get(f"dqkv.{i & 1}", (M, 3 * d), BF16)
        ops.attn_bwd(qkv, o, dO, lse, B, L, H, dh, self.causal, dqkv=dqkv, delta=bf.get("delta", (B, H, L), F32),
                     q_rows=1)

        q_cls = getattr(self, "_q_cls_only", False)      # the forward projected q for the class tokens only
        dq_c, a1_c = dqkv.view(B, L * 3 * d)[:, :d], a1.view(B, L * d)[:, :d]

        def w_attn():
            ops.gemm(ops.TN, ops.EPI_F32, g1_c, o_c, g("attn.out_proj.weight"), M=d, N=d, K=B)
            if q_cls:       # dq is zero off the class-token rows: W_q's gradient from those B rows, W_k / W_v's from all
                gw, gb = g("attn.in_proj_weight"), g("attn.in_proj_bias")
                ops.gemm_wgrad_bias(dqkv[:, d:], a1, gw[d:], gb[d:], M=2 * d, N=d, K=M, splitk=_splitk_for(2 * d, d, M))
                ops.gemm_wgrad_bias(dq_c, a1_c, gw[:d], gb[:d], M=d, N=d, K=B)
            else:
                ops.gemm_wgrad_bias(dqkv, a1, g("attn.in_proj_weight"), g("attn.in_proj_bias"), M=3 * d, N=d, K=M,
                                    splitk=_splitk_for(3 * d, d, M))
        on_side(w_attn, (dqkv,))
        dA = bf.get("dA", (M, d), BF16)
        wb = cp("attn.in_proj_weight").wb
        if q_cls:           # data gradient: the k | v columns for every row, then the q columns added on the class-token rows
            ops.gemm(ops.NT, ops.EPI_BF16, dqkv[:, d:], wb[:, d:], dA, M=M, N=d, K=2 * d)
            dA_c = dA.view(B, L * d)[:, :d]
            ops.gemm(ops.NT, ops.EPI_BF16_BIAS_RES, dq_c, wb[:, :d], dA_c, M=B, N=d, K=d, bias=self._zero_bias(d), res=dA_c)
        else:
            ops.gemm(ops.NT, ops.EPI_BF16, dqkv, wb, dA, M=M, N=d, K=3 * d)
        dres_bf = bf.get("dres_bf.0", (M, d), BF16)
        prev_bias = s.g(self._n(i - 1, "mlp.c_proj.bias")) if i > 0 else None
        qg = self._q8("g", M, d)
        t8g = None
        if self._fwd_w8 and i > 0 and qg is not None and _res_grad_bf16():     # per-tensor e4m3 copy: dY of block i - 1's c_proj weight gradient
            ig = 2 * self.layers + 2 * (i - 1) + 1
            t8g = (bf.get("t8.g.0", (M, d), torch.uint8), self._dq_scale[ig:ig + 1], self._dq_amax[ig])
        ops.layernorm_bwd(dA, self.x_in[i], bf.get(f"m1.{i}", (M,), F32), bf.get(f"r1.{i}", (M,), F32),
                          s.p(self._n(i, "ln_1.weight")), dres, dres_bf, g("ln_1.weight"), g("ln_1.bias"),
                          prev_bias, M, d, accumulate=-L,       # only the class-token rows carry a residual gradient
                          q8=qg and qg[0], q8_scale_inv=qg and qg[1],
                          g16=_res_grad_bf16(), write_f32=(i == 0), t8=t8g)  # bf16 stream: fp32 only in front of the stem
        self._g_t8_ok = t8g is not None
        self._g_q8 = qg
        return dres, dres_bf

    # -------------------------------------------------------------------------------- backward
    def backward(self, dres: torch.Tensor, dres_bf: torch.Tensor, last_bias_colsum_done: bool,
                 on_layer_done: Optional[Callable[[int], None]] = None) -> None:
        """``dres`` (fp32) / ``dres_bf`` (bf16 copy) hold dL/d(output of the last block) on entry; ``dres`` holds
        dL/d(input of block 0) on exit.  ``last_bias_colsum_done``: the caller already wrote the last block's
        c_proj.bias gradient (column sum of dres).

        Two HIP streams: the data-gradient chain (dgrad GEMMs, attention backward, LayerNorm backward) is the critical
        path on the caller's stream; weight-gradient GEMMs and bias column sums only feed the optimiser, so they run on
        a side stream and fill the tail rounds / memory-bound phases of the chain.  Buffers a side kernel reads
        (bf16 residual gradient, dU, dqkv) rotate, and the chain waits on the side stream's event before reusing one."""
        s, d, H, dh, mlp = self.s, self.d, self.H, self.dh, self.mlp
        B, L, M = self.B, self.L, self.M
        bf = self.bufs
        cp_of = lambda name: s.copies[name]
        # Whether the side stream pays depends on the shapes: round 2 measured 37.8 vs 38.4 ms/step with it on ViT-B/16 (B = 256); with
        # round 4's kernels the same model is 0.4-0.5 ms FASTER on one stream, while the configs[4] model gains 2.1-2.4 ms (bf16) / 4 ms
        # (e4m3) from the side stream (profiles/r04_side_stream_auto.txt).  Both schedules give the same bits (tests/test_gpu_model.py),
        # so the default, SC_OVERLAP=auto, times this stack's own backward in both (4 + 4 calls, alternating, after two warm-up
        # calls per batch shape) and keeps the faster one; SC_OVERLAP=1 / 0 pin it.
        ov_env = os.environ.get("SC_OVERLAP", "auto")
        main = torch.cuda.current_stream()
        trial = None
        if getattr(self, "no_side_stream", False):      # this stack runs beside the other tower on a stream of its own (net.py)
            ov_env = "0"
        if ov_env in ("0", "1"):
            overlap = ov_env == "1"
        else:
            overlap, trial = self._overlap_auto((B, L), main)
        if overlap and getattr(self, "_side", None) is None:
            # lowest priority the runtime offers: the side stream's workgroups should only take what the chain leaves
            self._side = torch.cuda.Stream(priority=int(os.environ.get("SC_SIDE_PRIO", "1")))
        side = self._side if overlap else main
        last_read = {}                                   # buffer id -> event recorded on the side stream

        def on_side(fn, reads) -> None:
            """Run fn on the side stream after everything enqueued on main so far; remember what it reads."""
            if not overlap:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                fn()
                done = torch.cuda.Event()
                done.record(side)
            for t in reads:
                last_read[t.data_ptr()] = done

        def before_write(t: torch.Tensor) -> None:
            ev = last_read.pop(t.data_ptr(), None)
            if ev is not None:
                main.wait_event(ev)

        # LayerNorm backward: the token pass stays on the chain; the column reductions (dgamma, dbeta, the bias gradient
        # of the Linear in front) feed only the optimiser, so with the side stream on they leave their partial sums in
        # one of four rotating workspaces and are finished on the side stream
        ln_ws_n = ops.layernorm_bwd_ws_floats(M, d)
        ln_ws = [bf.get(f"ln_ws.{k}", (ln_ws_n,), F32) for k in range(4)] if overlap else None
        ln_pos = [0]

        # fp8 path: LayerNorm backward also emits the e4m3 copy (+ row scales) of the residual gradient it has just
        # formed -- the A operand of the next data-gradient GEMM (c_proj after LN1 of the block above, out_proj after
        # LN2).  Producer and consumer are both on the chain stream, back to back: ONE scratch is enough.
        qg = self._q8("g", M, d)

        # The residual gradient between the blocks travels in bf16 (g_in -> g_bf, three rotating buffers), which is the
        # precision the reference's autocast carries it in; the fp32 buffer is written by the stack's last hop only (the
        # stem reads it).  SC_RES_GRAD=fp32 keeps the fp32 read-modify-write of rounds 1-2 (16 instead of 10 bytes per element).
        g16 = _res_grad_bf16() and self.res16_ok      # text tower: fp32 stream AND fp32 residual gradient, as the reference's

        def ln_bwd(dy, x, mean, rstd, gamma, g_in, g_bf, dgamma, dbeta, colsum, last=False, t8=None) -> None:
            kw = dict(q8=qg[0], q8_scale_inv=qg[1]) if qg is not None else {}
            if g16:
                kw.update(g16=True, g_in=g_in, write_f32=last)
            if t8 is not None:
                kw.update(t8=t8)
            if not overlap:
                ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, g_bf, dgamma, dbeta, colsum, M, d, accumulate=True, **kw)
                return
            ws = ln_ws[ln_pos[0] % 4]
            ln_pos[0] += 1
            before_write(ws)
            ops.layernorm_bwd(dy, x, mean, rstd, gamma, dres, g_bf, dgamma, dbeta, colsum, M, d, accumulate=True,
                              ws=ws, defer_reduce=True, **kw)
            on_side(lambda: ops.layernorm_bwd_reduce(ws, dgamma, dbeta, colsum, M, d), (ws,))

        def dgrad(epi, g_bf, name, out, *, N, K, have_q8, **kw) -> None:
            """Data-gradient GEMM out[M, N] = g[M, K] . W (NT against the transposed weight copy): e4m3 operands when the
            residual gradient's e4m3 copy is live in ``qg`` and the weight has one, bf16 otherwise."""
            c = cp_of(name)
            q8kw = kw.pop("q8kw", None)
            if have_q8 and qg is not None and c.wb8 is not None:
                ops.gemm_fp8(epi, qg[0], qg[1], c.wb8, c.wb8s, out, M=M, N=N, K=K, **kw, **(q8kw or {}))
            else:
                ops.gemm(ops.NT, epi, g_bf, c.wb, out, M=M, N=N, K=K, **kw)

        wg_mode = _wgrad_group_mode()
        dA = bf.get("dA", (M, d), BF16)
        dO = bf.get("dO", (M, d), BF16)
        delta = bf.get("delta", (B, H, L), F32)
        top = self.layers
        g_has_q8 = False                   # is the e4m3 copy of the current residual gradient in qg?
        if self.cls_only_last:
            dres, dres_bf = self._backward_last_cls(dres, dres_bf, on_side)
            top = self.layers - 1
            g_has_q8 = qg is not None
            if on_layer_done is not None:
                on_side(lambda: on_layer_done(self.layers - 1), ())
        ring = [dres_bf, bf.get("dres_bf.1", (M, d), BF16), bf.get("dres_bf.2", (M, d), BF16)]
        rpos = 0
        # e4m3 MLP weight gradients: per-tensor copies of the residual gradient travel beside the bf16 ring (same slots)
        w8 = bool(self._fwd_w8 and g16 and qg is not None)
        t8ring = [bf.get(f"t8.g.{k}", (M, d), torch.uint8) for k in range(3)] if w8 else None
        g0_t8_ok = w8 and self.cls_only_last and getattr(self, "_g_t8_ok", False)      # does t8ring[rpos] hold the copy of ring[rpos]?
        L2 = 2 * self.layers
        for i in reversed(range(top)):
            g = lambda leaf, i=i: s.g(self._n(i, leaf))
            cp = lambda leaf, i=i: s.copies[self._n(i, leaf)]
            a1, qkv, o = self._act("a1", i, (M, d)), bf.get(f"qkv.{i}", (M, 3 * d), BF16), bf.get(f"o.{i}", (M, d), BF16)
            a2, u, h = self._act("a2", i, (M, d)), bf.get(f"u.{i}", (M, mlp), BF16), self._act("h", i, (M, mlp))
            xmid = bf.get(f"xmid.{i}", (M, d), BF16 if self.r16 else F32)
            lse = bf.get(f"lse.{i}", (B, H, L), F32)
            dU = bf.get(f"dU.{i & 1}", (M, mlp), BF16)
            dqkv = bf.get(f"dqkv.{i & 1}", (M, 3 * d), BF16)
            g0 = ring[rpos]
            if i == self.layers - 1 and not last_bias_colsum_done:
                ops.colsum_bf16(g0, M, d, g("mlp.c_proj.bias"))
            # ---- MLP branch: x_out = xmid + c_proj(gelu(c_fc(ln_2(xmid))))
            before_write(dU)
            dq = None
            if self.fp8 and self._dq_on:      # GELU' epilogue: e4m3(dU) with last step's scale + max|dU| for the next one
                dU8 = bf.get(f"q8.dU.{i & 1}" if w8 else "q8.dU", (M, mlp), torch.uint8)     # rotates with dU when the side stream reads it
                before_write(dU8)
                dq = dict(q8_out=dU8, q8_scale=self._dq_scale[2 * i + 1:2 * i + 2], q8_amax=self._dq_amax[2 * i + 1])
            dU_has_q8 = bool(dq) and g_has_q8 and self._dq_ready
            # aux = the stored factor gelu'(u) (default) or u itself (recomputation mode): same bits either way
            dgrad(ops.EPI_BF16_MUL_AUX if self._u_holds_grad else self.epi_dgelu, g0, self._n(i, "mlp.c_proj.weight"), dU,
                  N=mlp, K=d, have_q8=g_has_q8, aux=u, q8kw=dq)

            mlp_probs = [(g0, h, g("mlp.c_proj.weight"), None, d, mlp), (dU, a2, g("mlp.c_fc.weight"), g("mlp.c_fc.bias"), mlp, d)]

            # e4m3 weight gradients of the MLP pair: every operand has a per-tensor copy made with scales that were ready when the
            # forward ran (h8 / a2 by the forward, dU8 / g0 by this backward); the first step after a reset stays in bf16
            w8_now = bool(w8 and self._dq_ready and dq is not None and g0_t8_ok and g_has_q8)
            g0_8 = t8ring[rpos] if w8_now else None

            def w_mlp(g0=g0, h=h, dU=dU, a2=a2, g=g, probs=mlp_probs, i=i, w8_now=w8_now, g0_8=g0_8, dU8=(dU8 if dq else None)):
                if w8_now:
                    sinv = self._dq_scale_inv
                    ops.gemm_wgrad_fp8(g0_8, sinv[L2 + 2 * i + 1:L2 + 2 * i + 2], bf.get(f"q8.h.{i}", (M, mlp), torch.uint8),
                                       sinv[2 * i:2 * i + 1], g("mlp.c_proj.weight"), None, M=d, N=mlp, K=M,
                                       splitk=_splitk_for(d, mlp, M))
                    ops.gemm_wgrad_fp8(dU8, sinv[2 * i + 1:2 * i + 2], bf.get(f"t8.a2.{i}", (M, d), torch.uint8),
                                       sinv[L2 + 2 * i:L2 + 2 * i + 1], g("mlp.c_fc.weight"), g("mlp.c_fc.bias"), M=mlp, N=d, K=M,
                                       splitk=_splitk_for(mlp, d, M))
                    return
                if _group_ok(wg_mode, "mlp", [(d, mlp), (mlp, d)], M):      # both weight gradients of the MLP branch in one launch
                    ops.gemm_wgrad_group(probs, K=M, splitk=_splitk_for_group([(d, mlp), (mlp, d)], M, one_round=wg_mode != "2"))
                    return
                ops.gemm(ops.TN, ops.EPI_F32, g0, h, g("mlp.c_proj.weight"), M=d, N=mlp, K=M, splitk=_splitk_for(d, mlp, M))
                ops.gemm_wgrad_bias(dU, a2, g("mlp.c_fc.weight"), g("mlp.c_fc.bias"), M=mlp, N=d, K=M,
                                    splitk=_splitk_for(mlp, d, M))
            if self.recompute and not w8_now:          # rebuild h = gelu(u) and a2 = ln_2(xmid) for the two (bf16) weight gradients
                before_write(h)
                ops.gelu_bf16(u, h, quick=self.quick_gelu)
                before_write(a2)
                ops.layernorm_fwd(xmid, s.p(self._n(i, "ln_2.weight")), s.p(self._n(i, "ln_2.bias")), a2,
                                  bf.get(f"m2.{i}", (M,), F32), bf.get(f"r2.{i}", (M,), F32), M, d)
            if wg_mode != "4":
                reads_mlp = (g0, dU, h, a2) if (self.recompute and not w8_now) else (g0, dU)
                if w8_now:
                    reads_mlp = reads_mlp + (g0_8, dU8)
                on_side(w_mlp, reads_mlp)
            cfc = cp("mlp.c_fc.weight")
            if dU_has_q8 and cfc.wb8 is not None:
                ops.gemm_fp8(ops.EPI_BF16, dU8, self._dq_scale_inv[2 * i + 1:2 * i + 2], cfc.wb8, cfc.wb8s, dA, M=M, N=d, K=mlp,
                             a_scale_scalar=True)
            else:
                ops.gemm(ops.NT, ops.EPI_BF16, dU, cfc.wb, dA, M=M, N=d, K=mlp)
            # LN2 backward accumulates into the residual gradient; its column sum is out_proj.bias' gradient
            rpos = (rpos + 1) % 3
            g1 = ring[rpos]
            before_write(g1)
            ln_bwd(dA, xmid, bf.get(f"m2.{i}", (M,), F32), bf.get(f"r2.{i}", (M,), F32),
                   s.p(self._n(i, "ln_2.weight")), g0, g1, g("ln_2.weight"), g("ln_2.bias"), g("attn.out_proj.bias"))
            # ---- attention branch: xmid = x_in + out_proj(attn(in_proj(ln_1(x_in))))
            dgrad(ops.EPI_BF16, g1, self._n(i, "attn.out_proj.weight"), dO, N=d, K=d, have_q8=qg is not None)
            before_write(dqkv)
            ops.attn_bwd(qkv, o, dO, lse, B, L, H, dh, self.causal, dqkv=dqkv, delta=delta)

            attn_probs = [(g1, o, g("attn.out_proj.weight"), None, d, d),
                          (dqkv, a1, g("attn.in_proj_weight"), g("attn.in_proj_bias"), 3 * d, d)]

            def w_attn(g1=g1, o=o, dqkv=dqkv, a1=a1, g=g, probs=attn_probs, more=mlp_probs):
                if wg_mode == "4":      # all four weight gradients of the block in one launch
                    ops.gemm_wgrad_group(more + probs, K=M, splitk=_splitk_for_group([(d, mlp), (mlp, d), (d, d), (3 * d, d)], M))
                    return
                if _group_ok(wg_mode, "attn", [(d, d), (3 * d, d)], M):      # out_proj (9 tiles) no longer needs split-K 28 to fill the chip
                    ops.gemm_wgrad_group(probs, K=M, splitk=_splitk_for_group([(d, d), (3 * d, d)], M, one_round=wg_mode != "2"))
                    return
                ops.gemm(ops.TN, ops.EPI_F32, g1, o, g("attn.out_proj.weight"), M=d, N=d, K=M, splitk=_splitk_for(d, d, M))
                ops.gemm_wgrad_bias(dqkv, a1, g("attn.in_proj_weight"), g("attn.in_proj_bias"), M=3 * d, N=d, K=M,
                                    splitk=_splitk_for(3 * d, d, M))
            if self.recompute:          # rebuild a1 = ln_1(x_in) for the in_proj weight gradient
                before_write(a1)
                ops.layernorm_fwd(self.x_in[i], s.p(self._n(i, "ln_1.weight")), s.p(self._n(i, "ln_1.bias")), a1,
                                  bf.get(f"m1.{i}", (M,), F32), bf.get(f"r1.{i}", (M,), F32), M, d)
            reads = (g1, dqkv, a1) if self.recompute else (g1, dqkv)
            if wg_mode == "4":
                reads = reads + ((g0, dU, h, a2) if self.recompute else (g0, dU))
            on_side(w_attn, reads)
            ops.gemm(ops.NT, ops.EPI_BF16, dqkv, cp("attn.in_proj_weight").wb, dA, M=M, N=d, K=3 * d)
            # LN1 backward; its column sum is the previous block's c_proj.bias gradient
            prev_bias = s.g(self._n(i - 1, "mlp.c_proj.bias")) if i > 0 else None
            rpos = (rpos + 1) % 3
            g2 = ring[rpos]
            before_write(g2)
            t8n = None
            if w8 and i > 0:        # g2 is block i - 1's g0: its per-tensor e4m3 copy, in the ring slot beside it
                before_write(t8ring[rpos])
                ig = L2 + 2 * (i - 1) + 1
                t8n = (t8ring[rpos], self._dq_scale[ig:ig + 1], self._dq_amax[ig])
            ln_bwd(dA, self.x_in[i], bf.get(f"m1.{i}", (M,), F32), bf.get(f"r1.{i}", (M,), F32),
                   s.p(self._n(i, "ln_1.weight")), g1, g2, g("ln_1.weight"), g("ln_1.bias"), prev_bias, last=(i == 0), t8=t8n)
            g_has_q8 = qg is not None
            g0_t8_ok = t8n is not None
            if on_layer_done is not None:
                on_side(lambda i=i: on_layer_done(i), ())     # the bucket all-reduce follows the side stream
        # Join the side stream BEFORE the scale update: the side stream's e4m3 weight-gradient GEMMs read ``_dq_scale_inv``
        # through device pointers when they RUN (sc_gemm8p.hip), and the low-priority side stream can lag the chain by
        # several blocks -- an update enqueued on the chain first could replace a scale that operands quantised with the old
        # one are still waiting to be multiplied by (advisor, round 4).
        if overlap:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        if self.fp8 and self._dq_on:          # next step's per-tensor scales from the maxima of the last FP8_AMAX_HISTORY steps
            ops.fp8_scale_update(self._dq_amax, self._dq_scale, self._dq_scale_inv, margin_bits=1, hist=self._dq_hist,
                                 slot=self._dq_step % self.FP8_AMAX_HISTORY)
            self._dq_step += 1
            self._dq_ready = True
        if trial is not None:                     # end of a timed trial call (the side stream's work is joined above)
            trial[2].record(main)
        return dres

    OVERLAP_TRIAL_CALLS = (2, 4)                  # untimed warm-up calls, timed calls per schedule
    OVERLAP_MARGIN = 0.01                         # one stream must beat the side stream by this fraction to be chosen

    def _overlap_auto(self, key, main):
        """SC_OVERLAP=auto: (use the side stream?, trial record or None) for this backward call of batch shape ``key``."""
        st = getattr(self, "_ov_auto", None)
        if st is None:
            st = self._ov_auto = {}
        s = st.setdefault(key, {"calls": 0, "ms": {True: [], False: []}, "pending": [], "choice": None})
        if s["choice"] is not None:
            return s["choice"], None
        warm, per = self.OVERLAP_TRIAL_CALLS
        for rec in list(s["pending"]):            # trials whose end event has passed (normally a whole step ago)
            if rec[2].query():
                s["ms"][rec[0]].append(rec[1].elapsed_time(rec[2]))
                s["pending"].remove(rec)
        c = s["calls"]
        s["calls"] += 1
        if c < warm:
            return True, None
        if c >= warm + 2 * per:
            for rec in s["pending"]:              # decision time: wait for the last trial if it is still in flight
                rec[2].synchronize()
                s["ms"][rec[0]].append(rec[1].elapsed_time(rec[2]))
            s["pending"] = []
            on, off = s["ms"][True], s["ms"][False]
            # The side stream is the schedule of rounds 1-3 and the safe one for the big models; leave it only when one stream is
            # faster by a clear margin (the two are within 1 % on ViT-B/16 and the trial timings carry that much noise: without
            # a margin the choice, and with it the step time, varies from run to run -- advisor, round 4).  All ranks take rank
            # 0's decision: the trials of a rank include its collective waits, and ranks on different schedules would only
            # wait for each other.
            choice = True
            if on and off:
                choice = not (min(off) < min(on) * (1.0 - self.OVERLAP_MARGIN))
                s["ms_best"] = {"side_stream": round(min(on), 3), "one_stream": round(min(off), 3)}
            from . import comm
            s["choice"] = comm.broadcast_flag(choice)
            return s["choice"], None
        use = ((c - warm) % 2) == 0               # on, off, on, off, ...
        rec = (use, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        rec[1].record(main)
        s["pending"].append(rec)
        return use, rec

def _res_stream_bf16(configured: str = "bf16") -> bool:
    """Read at every forward: the residual stream of the patch towers in bf16 (default: the reference's precision under its
    bf16 autocast) or fp32 -- SpatialClipNet(residual_stream=...), overridden by SC_RES_STREAM=bf16|fp32 for A/B."""
    return os.environ.get("SC_RES_STREAM", configured) == "bf16"


def _res_grad_bf16() -> bool:
    """Read at every backward: SC_RES_GRAD=fp32 restores the fp32 residual-gradient buffer (A/B, tests)."""
    return os.environ.get("SC_RES_GRAD", "bf16") != "fp32"


class PatchTransformerTower:
    """Patch embedding (one bias-free linear map per patch) + class token + learned positions + ln_pre + N residual
    attention blocks + ln_post on the class token (pool 'tok') + output projection + L2 normalise.  The image tower
    (VisionTransformer, src/open_clip/transformer.py:583-918) and the gene transformer (configs[4]; 1-D patches of the
    expression vector) are both instances: they differ in how the [B * patches, patch_dim] operand is formed
    (``_patchify``) and in the parameter-name prefix."""

    prefix = "visual."
    quick_gelu = False          # set by the subclass before super().__init__: only the reference's towers take the config flag

    def __init__(self, cfg: ModelCfg, store: ParamStore, width: int, heads: int, layers: int, mlp_ratio: float,
                 tokens: int, patch_dim: int):
        self.cfg, self.s = cfg, store
        self.d, self.D, self.L = width, cfg.embed_dim, tokens
        self.n_layers = layers
        self.kp = patch_dim
        self.kp_pad = store.copies[self.prefix + "conv1.weight"].k_pad
        self.stack = TransformerStack(store, self.prefix + "transformer.resblocks.", width, heads, layers,
                                      int(width * mlp_ratio), causal=False, cls_only_last=True, res16_ok=True,
                                      quick_gelu=self.quick_gelu)
        self.bufs = _Bufs(store.device)

    def _n(self, leaf: str) -> str:
        return self.prefix + leaf

    def param_names_head(self) -> List[str]:
        return [self._n("ln_post.weight"), self._n("ln_post.bias"), self._n("proj")]

    def param_names_stem(self) -> List[str]:
        return [self._n("conv1.weight"), self._n("class_embedding"), self._n("positional_embedding"),
                self._n("ln_pre.weight"), self._n("ln_pre.bias")]

    def _patchify(self, x: torch.Tensor) -> torch.Tensor:
        """-> bf16 [B * (L - 1), kp_pad] patch rows (sets self.B)."""
        raise NotImplementedError

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        s, d, D, L = self.s, self.d, self.D, self.L
        s.wait_names(self.param_names_stem())
        patches = self._patchify(inp)
        B = self.B
        M, Mp = B * L, B * (L - 1)
        bf = self.bufs
        patch_out = bf.get("patch_out", (Mp, d), F32)
        ops.gemm(ops.NT, ops.EPI_F32, patches, s.copies[self._n("conv1.weight")].wf, patch_out, M=Mp, N=d, K=self.kp_pad)
        # the stream's first tensor in the stream's own precision: with the bf16 stream ln_pre writes bf16 (as the reference's does
        # under bf16-mixed) instead of an fp32 copy that a cast pass then halves (310 MB of traffic per step at ViT-B/16)
        r16 = self.stack.res16_ok and _res_stream_bf16(self.stack.res_stream) and os.environ.get("SC_STEM_BF16", "1") != "0"
        x0 = bf.get("x0.16", (M, d), BF16) if r16 else bf.get("x0", (M, d), F32)
        ops.embed_ln_fwd(patch_out, s.p(self._n("class_embedding")), s.p(self._n("positional_embedding")),
                         s.p(self._n("ln_pre.weight")), s.p(self._n("ln_pre.bias")), x0,
                         bf.get("m_pre", (M,), F32), bf.get("r_pre", (M,), F32), B, L, d)
        xf = self.stack.forward(x0, B, L)
        self.xf = xf
        s.wait_names(self.param_names_head())
        pooled = bf.get("pooled", (B, d), BF16)
        self.x_ld = d if self.stack.cls_only_last else L * d          # xf is compact [B, d] in CLS-only mode
        ops.layernorm_fwd(xf, s.p(self._n("ln_post.weight")), s.p(self._n("ln_post.bias")), pooled,
                          bf.get("m_post", (B,), F32), bf.get("r_post", (B,), F32), B, d, ldx=self.x_ld)
        f_raw = bf.get("f_raw", (B, D), F32)
        ops.gemm(ops.NT, ops.EPI_F32, pooled, s.copies[self._n("proj")].wf, f_raw, M=B, N=D, K=d)
        f = torch.empty((B, D), dtype=F32, device=inp.device)
        ops.l2norm_fwd(f_raw, f, None, bf.get("inv", (B,), F32), B, D)
        self.f = f
        return f

    def raw_features(self) -> torch.Tensor:
        """Projected features of the last forward before F.normalize (encode_image(normalize=False))."""
        return self.bufs.get("f_raw", (self.B, self.D), F32)

    def backward(self, d_f: torch.Tensor, on_bucket: Optional[Callable[[List[str]], None]] = None) -> None:
        s, d, D, L, B = self.s, self.d, self.D, self.L, self.B
        M, Mp = B * L, B * (L - 1)
        bf = self.bufs
        d_raw = bf.get("d_raw", (B, D), BF16)
        ops.l2norm_bwd(d_f.contiguous().float(), self.f, bf.get("inv", (B,), F32), d_raw, B, D)
        pooled = bf.get("pooled", (B, d), BF16)
        d_pooled = bf.get("d_pooled", (B, d), BF16)
        cp = s.copies[self._n("proj")]
        ops.gemm(ops.NT, ops.EPI_BF16, d_raw, cp.wb, d_pooled, M=B, N=d, K=D)
        ops.gemm(ops.TN, ops.EPI_F32, pooled, d_raw, s.g(self._n("proj")), M=d, N=D, K=B)
        last = self.n_layers - 1
        if self.stack.cls_only_last:
            # class-token-only last block: its fp32 residual gradient lives in the class-token rows of the FULL buffer
            # (row stride L * d) from the start, so nothing has to be zeroed or copied when the full-width blocks take over
            dres = self.stack.bufs.get("dres_full", (M, d), F32)
            dres_bf = bf.get("dres_c_bf", (B, d), BF16)
            ld = d
        else:
            dres = bf.get("dres", (M, d), F32)
            dres_bf = bf.get("dres_bf", (M, d), BF16)
            dres.zero_()
            dres_bf.zero_()
            ld = L * d
        ops.layernorm_bwd(d_pooled, self.xf, bf.get("m_post", (B,), F32), bf.get("r_post", (B,), F32),
                          s.p(self._n("ln_post.weight")), dres, dres_bf, s.g(self._n("ln_post.weight")),
                          s.g(self._n("ln_post.bias")), s.g(self._n(f"transformer.resblocks.{last}.mlp.c_proj.bias")),
                          B, d, accumulate=False, ldx=self.x_ld, lddres=L * d, lddbf=ld)
        if on_bucket is not None:
            on_bucket(self.param_names_head())
        cb = (lambda i: on_bucket(self.stack.layer_param_names(i))) if on_bucket is not None else None
        dres = self.stack.backward(dres, dres_bf, last_bias_colsum_done=True, on_layer_done=cb)
        # stem: ln_pre / positional / class embedding / patch embedding (no gradient flows to the input)
        dpatch = bf.get("dpatch", (Mp, d), BF16)
        ops.embed_ln_bwd(dres, bf.get("patch_out", (Mp, d), F32), s.p(self._n("class_embedding")),
                         s.p(self._n("positional_embedding")), bf.get("m_pre", (M,), F32), bf.get("r_pre", (M,), F32),
                         s.p(self._n("ln_pre.weight")), dpatch, s.g(self._n("ln_pre.weight")), s.g(self._n("ln_pre.bias")),
                         s.g(self._n("positional_embedding")), s.g(self._n("class_embedding")), B, L, d)
        gw = s.g(self._n("conv1.weight")).view(d, self.kp)
        ops.gemm(ops.TN, ops.EPI_F32, dpatch, bf.get("patches", (Mp, self.kp_pad), BF16), gw, M=d, N=self.kp, K=Mp,
                 splitk=_splitk_for(d, self.kp, Mp))
        if on_bucket is not None:
            on_bucket(self.param_names_stem())


class VisionTower(PatchTransformerTower):
    """VisionTransformer (pool 'tok', learnable pos-embed, ln_pre/ln_post, output projection) + L2 normalise."""

    prefix = "visual."

    def __init__(self, cfg: ModelCfg, store: ParamStore):
        v = cfg.vision
        self.v = v
        self.quick_gelu = bool(getattr(cfg, "quick_gelu", False))
        super().__init__(cfg, store, v.width, v.heads, v.layers, v.mlp_ratio, v.tokens, 3 * v.patch_size * v.patch_size)

    def _patchify(self, images: torch.Tensor) -> torch.Tensor:
        v = self.v
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != v.image_size or images.shape[3] != v.image_size:
            raise ValueError(f"images must be [B,3,{v.image_size},{v.image_size}], got {tuple(images.shape)}")
        images = images.contiguous().float()
        self.B = B = images.shape[0]
        patches = self.bufs.get("patches", (B * (self.L - 1), self.kp_pad), BF16)
        if self.kp_pad != self.kp and getattr(self, "_pad_zeroed", None) is not patches:
            patches.zero_()            # the K padding must be finite zeros; im2col only writes the real columns
            self._pad_zeroed = patches
        ops.im2col(images, patches, v.patch_size)
        return patches


class GeneTransformerTower(PatchTransformerTower):
    """BASELINE configs[4]'s gene transformer (no reference symbol; model_configs.GeneCfg kind="transformer"): fills the
    ``texts`` slot of the batch with a float [B, n_genes] matrix like the gene-MLP tower."""

    prefix = "gene."

    def __init__(self, cfg: ModelCfg, store: ParamStore):
        g = cfg.gene
        self.g = g
        if g.patch % 64:
            raise ValueError(f"gene patch size {g.patch} must be a multiple of 64")
        super().__init__(cfg, store, g.width, g.heads, g.layers, g.mlp_ratio, g.tokens, g.patch)

    def param_names(self) -> List[str]:
        return [n for n in self.s.by_name if n.startswith("gene.")]

    def _patchify(self, x: torch.Tensor) -> torch.Tensor:
        g = self.g
        if x.dim() != 2 or x.shape[1] != g.n_genes:
            raise ValueError(f"gene matrix must be [B,{g.n_genes}], got {tuple(x.shape)}")
        x = x.contiguous().float()
        self.B = B = x.shape[0]
        T = self.L - 1
        patches = self.bufs.get("patches", (B * T, g.patch), BF16)
        # one cast pass: row b of the padded [B, T * patch] matrix IS rows b*T .. b*T+T-1 of the patch operand
        ops.cast_pad_bf16(x, patches.view(B, T * g.patch), B, g.n_genes, T * g.patch)
        return patches


class GeneTower:
    """Row G of SURVEY.md section 8a (no reference symbol): gene-expression MLP  n_genes -> hidden -(GELU)-> embed_dim,
    L2-normalised; fills the ``texts`` slot of the batch with a float [B, n_genes] matrix."""

    def __init__(self, cfg: ModelCfg, store: ParamStore):
        g = cfg.gene
        self.cfg, self.g, self.s = cfg, g, store
        self.D = cfg.embed_dim
        self.kpad = store.copies["gene.fc1.weight"].k_pad
        self.bufs = _Bufs(store.device)

    def param_names(self) -> List[str]:
        return ["gene.fc1.weight", "gene.fc1.bias", "gene.fc2.weight", "gene.fc2.bias"]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s, g, D = self.s, self.g, self.D
        if x.dim() != 2 or x.shape[1] != g.n_genes:
            raise ValueError(f"gene matrix must be [B,{g.n_genes}], got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        self.B = B
        bf = self.bufs
        s.wait_names(self.param_names())
        xg = bf.get("xg", (B, self.kpad), BF16)
        ops.cast_pad_bf16(x, xg, B, g.n_genes, self.kpad)
        u1 = bf.get("u1", (B, g.hidden), BF16)
        h1 = bf.get("h1", (B, g.hidden), BF16)
        # M = batch is small and K = n_genes is huge: split K over the chip (fp32 slabs), then bias + GELU
        ktiles = self.kpad // 64
        tiles = ((B + 127) // 128) * ((g.hidden + 127) // 128)
        splitk = max(1, min(ktiles // 2, 256 // max(tiles, 1), 64))
        if splitk > 1:
            pre = bf.get("pre1", (B, g.hidden), F32)
            ops.gemm(ops.NT, ops.EPI_F32, xg, s.copies["gene.fc1.weight"].wf, pre, M=B, N=g.hidden, K=self.kpad,
                     splitk=splitk)
            ops.bias_gelu_pair(pre, s.p("gene.fc1.bias"), u1, h1, B, g.hidden)
        else:
            ops.gemm(ops.NT, ops.EPI_GELU_PAIR, xg, s.copies["gene.fc1.weight"].wf, u1, M=B, N=g.hidden, K=self.kpad,
                     bias=s.p("gene.fc1.bias"), out2=h1)
        f_raw = bf.get("f_raw", (B, D), F32)
        ops.gemm(ops.NT, ops.EPI_F32_BIAS_RES, h1, s.copies["gene.fc2.weight"].wf, f_raw, M=B, N=D, K=g.hidden,
                 bias=s.p("gene.fc2.bias"))
        f = torch.empty((B, D), dtype=F32, device=x.device)
        ops.l2norm_fwd(f_raw, f, None, bf.get("inv", (B,), F32), B, D)
        self.f = f
        return f

    def raw_features(self) -> torch.Tensor:
        return self.bufs.get("f_raw", (self.B, self.D), F32)

    def backward(self, d_f: torch.Tensor, on_bucket: Optional[Callable[[List[str]], None]] = None) -> None:
        s, g, D, B = self.s, self.g, self.D, self.B
        bf = self.bufs
        d_raw = bf.get("d_raw", (B, D), BF16)
        ops.l2norm_bwd(d_f.contiguous().float(), self.f, bf.get("inv", (B,), F32), d_raw, B, D)
        h1, u1 = bf.get("h1", (B, g.hidden), BF16), bf.get("u1", (B, g.hidden), BF16)
        ops.gemm(ops.TN, ops.EPI_F32, d_raw, h1, s.g("gene.fc2.weight"), M=D, N=g.hidden, K=B)
        ops.colsum_bf16(d_raw, B, D, s.g("gene.fc2.bias"))
        dU = bf.get("dU", (B, g.hidden), BF16)
        ops.gemm(ops.NT, ops.EPI_BF16_DGELU, d_raw, s.copies["gene.fc2.weight"].wb, dU, M=B, N=g.hidden, K=D, aux=u1)
        ops.gemm(ops.TN, ops.EPI_F32, dU, bf.get("xg", (B, self.kpad), BF16), s.g("gene.fc1.weight"),
                 M=g.hidden, N=g.n_genes, K=B)
        ops.colsum_bf16(dU, B, g.hidden, s.g("gene.fc1.bias"))
        if on_bucket is not None:
            on_bucket(self.param_names())


class TextTower:
    """The reference's second tower: CLIP text transformer on BPE token ids (CLIP.encode_text,
    src/open_clip/model.py:330-345): embedding gather + positional embedding, causal pre-LN blocks, ln_final, EOT
    (argmax id) pooling, text_projection, L2 normalise.  ln_final is per token, so only the pooled rows are
    normalised (the reference normalises all 77 and then picks one)."""

    def __init__(self, cfg: ModelCfg, store: ParamStore):
        t = cfg.text
        self.cfg, self.t, self.s = cfg, t, store
        self.d, self.D, self.L, self.V = t.width, cfg.embed_dim, t.context_length, t.vocab_size
        self.stack = TransformerStack(store, "transformer.resblocks.", t.width, t.heads, t.layers,
                                      int(t.width * t.mlp_ratio), causal=True, quick_gelu=bool(getattr(cfg, "quick_gelu", False)))
        self.bufs = _Bufs(store.device)

    def param_names_head(self) -> List[str]:
        return ["ln_final.weight", "ln_final.bias", "text_projection"]

    def param_names_stem(self) -> List[str]:
        return ["token_embedding.weight", "positional_embedding"]

    def forward(self, text: torch.Tensor) -> torch.Tensor:
        s, d, D, L = self.s, self.d, self.D, self.L
        if text.dim() != 2 or text.shape[1] != L or text.dtype != torch.int64:
            raise ValueError(f"texts must be int64 [B,{L}] token ids, got {tuple(text.shape)} {text.dtype}")
        text = text.contiguous()
        B = text.shape[0]
        M = B * L
        self.B, self.text = B, text
        bf = self.bufs
        x0 = bf.get("x0", (M, d), F32)
        s.wait_names(self.param_names_stem())
        ops.token_embed_fwd(text, s.p("token_embedding.weight"), s.p("positional_embedding"), x0, B, L, d, self.V)
        xf = self.stack.forward(x0, B, L)
        s.wait_names(self.param_names_head())
        eot = bf.get("eot", (B,), torch.int32)
        ops.argmax_rows(text, eot, B, L)
        xe = bf.get("x_eot", (B, d), F32)
        ops.gather_rows(xf, eot, L, xe, B, d)
        pooled = bf.get("pooled", (B, d), BF16)
        ops.layernorm_fwd(xe, s.p("ln_final.weight"), s.p("ln_final.bias"), pooled, bf.get("m_post", (B,), F32),
                          bf.get("r_post", (B,), F32), B, d)
        f_raw = bf.get("f_raw", (B, D), F32)
        ops.gemm(ops.NT, ops.EPI_F32, pooled, s.copies["text_projection"].wf, f_raw, M=B, N=D, K=d)
        f = torch.empty((B, D), dtype=F32, device=text.device)
        ops.l2norm_fwd(f_raw, f, None, bf.get("inv", (B,), F32), B, D)
        self.f = f
        return f

    def raw_features(self) -> torch.Tensor:
        return self.bufs.get("f_raw", (self.B, self.D), F32)

    def backward(self, d_f: torch.Tensor, on_bucket: Optional[Callable[[List[str]], None]] = None) -> None:
        s, d, D, L, B = self.s, self.d, self.D, self.L, self.B
        M = B * L
        bf = self.bufs
        d_raw = bf.get("d_raw", (B, D), BF16)
        ops.l2norm_bwd(d_f.contiguous().float(), self.f, bf.get("inv", (B,), F32), d_raw, B, D)
        pooled = bf.get("pooled", (B, d), BF16)
        d_pooled = bf.get("d_pooled", (B, d), BF16)
        ops.gemm(ops.NT, ops.EPI_BF16, d_raw, s.copies["text_projection"].wb, d_pooled, M=B, N=d, K=D)
        ops.gemm(ops.TN, ops.EPI_F32, pooled, d_raw, s.g("text_projection"), M=d, N=D, K=B)
        dxe = bf.get("dx_eot", (B, d), F32)
        last = self.t.layers - 1
        ops.layernorm_bwd(d_pooled, bf.get("x_eot", (B, d), F32), bf.get("m_post", (B,), F32),
                          bf.get("r_post", (B,), F32), s.p("ln_final.weight"), dxe, None, s.g("ln_final.weight"),
                          s.g("ln_final.bias"), s.g(f"transformer.resblocks.{last}.mlp.c_proj.bias"), B, d,
                          accumulate=False)
        dres = bf.get("dres", (M, d), F32)
        dres_bf = bf.get("dres_bf", (M, d), BF16)
        dres.zero_()
        dres_bf.zero_()
        ops.scatter_rows(dxe, bf.get("eot", (B,), torch.int32), L, dres, dres_bf, B, d)
        if on_bucket is not None:
            on_bucket(self.param_names_head())
        cb = (lambda i: on_bucket(self.stack.layer_param_names(i))) if on_bucket is not None else None
        self.stack.backward(dres, dres_bf, last_bias_colsum_done=True, on_layer_done=cb)
        # scatter-add of the embedding gather: bit-reproducible by default (SC_EMBED_BWD_ATOMIC=1: the float-atomic kernel, A/B)
        ops.token_embed_bwd(self.text, dres, s.g("token_embedding.weight"), s.g("positional_embedding"), B, L, d, self.V,
                            eot=bf.get("eot", (B,), torch.int32), deterministic=os.environ.get("SC_EMBED_BWD_ATOMIC", "0") != "1")
        if on_bucket is not None:
            on_bucket(self.param_names_stem())
