"""Kernel sequencing for one RPO step on one MI355X.

This is the host-side counterpart of what sits under ``CustomCLIP.forward``
in the reference (trainers/rpo.py:161-232 calling clip/model.py:188-207): it
owns the packed frozen weights and the activation workspace in HBM and
enqueues librpo_hip.so kernels on HIP streams.  It exploits the read-only
structure (SURVEY.md finding 4): nothing reads a prompt, so

* the image tower's frozen tokens run inference-only and all S = N + K rows of
  every image share each layer's GEMM launches,
* the text tower's frozen tokens are run ONCE (``cache_text_kv``), leaving the
  n_cls*K prompt rows per step,
* the backward touches only the B*K / n_cls*K prompt rows (contiguous at the
  bottom of the row layout) and needs only dX GEMMs (weights are frozen).

Row layout of the image tower: [B*N frozen rows | B*K prompt rows], see
include/rpo_amd.h.  All HBM buffers are allocated once, sized for ``max_batch``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
import os

from ._lib import (EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_LN_BIAS, EPI_LN_BIAS_QGELU, EPI_NONE, EPI_PATCH,
                   EPI_QGELU_BWD)
from .config import RPOConfig
from .engine_coop import CoopEngineMixin

import contextlib

SCALE = 1.0 / math.sqrt(64.0)
_NO_PROBE = contextlib.nullcontext()


# split-K factors of the two fp32-output dX GEMMs of a block's backward (few output tiles, long K):
# d c_fc has K = 4d, d q-proj has K = d.  Slabs are summed in fixed order by rpo_layernorm_bwd.  Tuned at step level:
# 4, 6, 8 slabs for c_fc and 3, 4 for the q-projection were slower (more slabs cost output bandwidth).
SPLIT_FC, SPLIT_Q = 3, 2


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class _Block:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor; ln2_w: torch.Tensor; ln2_b: torch.Tensor
    w_in: torch.Tensor; b_in: torch.Tensor          # [3d, d], [3d]
    w_out: torch.Tensor; b_out: torch.Tensor        # [d, d]
    w_fc: torch.Tensor; b_fc: torch.Tensor          # [4d, d]
    w_proj: torch.Tensor; b_proj: torch.Tensor      # [d, 4d]
    w_q_t: torch.Tensor                             # [d, d]   = w_in[:d].T      (dX of the q projection)
    w_out_t: torch.Tensor                           # [d, d]
    w_fc_t: torch.Tensor                            # [d, 4d]
    w_proj_t: torch.Tensor                          # [4d, d]
    w_oq_t: Optional[torch.Tensor] = None           # [2, d, d] = (w_out_t, w_q_t) in one allocation (prefetch hint)
    # LayerNorm folded into the consuming GEMM (include/rpo_amd.h RPO_EPI_LN_*; image tower, 16-bit modes):
    # W' = gamma o W (act dtype), s = row sums of the ROUNDED W', b' = b + W beta
    w_in_ln: Optional[torch.Tensor] = None; s_in: Optional[torch.Tensor] = None; b_in_ln: Optional[torch.Tensor] = None
    w_fc_ln: Optional[torch.Tensor] = None; s_fc: Optional[torch.Tensor] = None; b_fc_ln: Optional[torch.Tensor] = None


class Engine(CoopEngineMixin):
    """The product engine: the RPO step, its eval branch, plain CLIP, and (engine_coop.CoopEngineMixin) the sibling
    trainers.  The measured-slower experiments of rounds 3 / 4 are NOT here: rpo_amd/experimental.py subclasses this class
    and overrides the hooks marked "experiment hook" below; `make_engine` returns that subclass only under
    RPO_EXPERIMENTAL=1, so a default run imports and executes none of it."""
    def __init__(self, cfg: RPOConfig, state_dict: Dict[str, np.ndarray], tokens: np.ndarray,
                 device: torch.device, act_dtype: torch.dtype = torch.bfloat16, max_batch: int = 32):
        if not torch.cuda.is_available():
            raise RuntimeError("rpo_amd.Engine needs a HIP device; there is no CPU path")
        ops.version()                                   # loads the library or raises
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.cfg, self.dev, self.act = cfg, device, act_dtype
        self.max_batch = max_batch
        self.kmult = 32 if act_dtype == torch.float32 else 64
        # ln_1 / ln_2 of the image blocks ride in the epilogues of the GEMMs around them (16-bit modes; the f32 parity
        # mode keeps the stand-alone LayerNorm kernels, bit for bit what the goldens were validated with)
        self.fold_ln = act_dtype != torch.float32 and os.environ.get("RPO_NO_LN_FOLD") != "1"
        # prefetch hints on the prompt-row chains (text tower, both backward chains): A/B switch RPO_NO_CHAIN_PREFETCH=1
        self._pf_chains = (act_dtype != torch.float32 and os.environ.get("RPO_NO_WPREFETCH") != "1"
                           and os.environ.get("RPO_NO_CHAIN_PREFETCH") != "1")
        # The prompt-row GEMMs (text tower, last image block, both backward chains) on rpo_gemm_ws: the frozen weight packed
        # fragment-major once at load and streamed global -> VGPR (csrc/gemm_ws.hip).  16-bit modes; A/B switch RPO_NO_WS=1.
        self.use_ws = act_dtype != torch.float32 and os.environ.get("RPO_NO_WS") != "1"
        self._wsp: Dict[tuple, "ops.PackedWeight"] = {}   # (data_ptr, shape) of a row-major weight -> its packed twin
        self._ws_okc: Dict[tuple, bool] = {}
        self._ws_cfg = 0
        # The text tower is not on the step's critical path; what it costs is its kernels' footprint beside the image stream.
        # Its GEMMs therefore run on 64x64 tiles (a quarter of the workgroups of the 32x32 tiles the kernel would pick for
        # itself: each launch is slower alone, the step 0.8 % faster -- profiles/r05_ab_ws_text_tiles.txt).  0 = the kernel's choice.
        # Only where the image forward is long enough to hide the slower text launches (>= 6000 token rows: batch 32 of
        # ViT-B/16): at batch 4 / 8 / 16 the text chain is on the critical path and the kernel's own choice wins by 3.2 / 2.9
        # / 0.6 % (same file).
        big = max_batch * (cfg.n_frozen + cfg.K) >= 6000
        self._ws_text_cfg = int(os.environ.get("RPO_WS_TEXT_CFG", "220" if big else "0"))
        # ... and the text BACKWARD, which runs beside the image backward's small launches instead of beside one-round
        # kernels, on 32x64 tiles there: -0.3 .. -0.4 % in 8 of 8 alternating rounds on two boxes; at batch 4 / 8 and for
        # ViT-L/14 the kernel's choice stays (+0.9 / +0.7 / +0.4 % otherwise) -- same file.
        # A/B knob: per-GEMM geometry of the text tower, e.g. RPO_WS_TEXT_TILES="proj=120,dfc=220" (q, out, fc, proj: forward;
        # dproj, dfc, dq: backward) on top of the two defaults above
        self._text_tiles = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("RPO_WS_TEXT_TILES", "").split(",") if kv))
        self._ws_text_cfg_bwd = int(os.environ.get("RPO_WS_TEXT_CFG_BWD", os.environ.get("RPO_WS_TEXT_CFG", "120" if big else "0")))
        # Thousands of text rows (ImageNet's 1000 classes x K = 24: the reference's xd_train.sh): c_fc saves the QuickGELU
        # derivative of EVERY row, which the 256x256 kernel's generic epilogue does in a second pass (131 us per launch
        # against 69 without, profiles/r06_trace_ncls1000.txt); the row-unit kernel forms it in its block loop.  It takes
        # row units that tile the rows in whole rounds of the CUs: u rows each with 225 <= u <= 288 and (rows / u) x
        # (4 d_t / 256) a multiple of 256 -- 250 at 24 000 rows.  No such u: the heuristic's choice stays.
        Rt_, tn = cfg.n_cls * cfg.K, (4 * cfg.d_t) // 256
        u = next((u for u in range(288, 224, -1) if Rt_ % u == 0 and ((Rt_ // u) * tn) % 256 == 0), 0)
        self._text_fc_units = (dict(row_units=(u, 0, Rt_)) if (u and Rt_ >= 2048 and act_dtype != torch.float32
                                                                 and os.environ.get("RPO_NO_TEXT_UNITS") != "1") else {})
        tokens = np.asarray(tokens, dtype=np.int64)
        assert tokens.shape == (cfg.n_cls, cfg.context)
        self.len_np = tokens.argmax(-1) + 1             # trainers/rpo.py:137
        assert int(self.len_np.max()) + cfg.K <= cfg.context, "len_c + K must fit the context (rpo.py:176)"
        self.Lmax = int(self.len_np.max())
        self.len_i32 = torch.tensor(self.len_np, dtype=torch.int32, device=device)
        self.logit_scale_exp = float(np.exp(np.float32(state_dict["logit_scale"])))
        with torch.cuda.device(device):                 # the convert kernels of _pack launch on the current device
            self._pack(state_dict, tokens)
            self._alloc()
        self.text_cache_ready = False
        self._stats_group = {}          # batch size -> columns per partial LayerNorm statistic (64 or 96)
        self.img_cls_f = None
        self.plain_text_f = None
        self.text_x_final = None
        self._eval_graphs = {}          # batch size -> (captured image tower + head, its static input)
        self.probe = None               # optional callable(name) -> context manager bracketing one launch (bench.py)
        self.text_f_version = -1        # prompts version the cached eval text features belong to
        self.params_version = 0         # bumped by whoever changes the prompts (optimiser step, load)

    # ------------------------------------------------------------------ weights
    def _f32(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)

    def _act(self, w: torch.Tensor) -> torch.Tensor:
        """fp32 device matrix -> act dtype via the library's convert kernel."""
        w = w.contiguous()
        if self.act == torch.float32:
            return w
        out = torch.empty(w.shape, dtype=self.act, device=self.dev)
        return ops.convert(w, out)

    def _fold(self, w: np.ndarray, b: np.ndarray, gamma: np.ndarray, beta: np.ndarray):
        """(W', s, b') of the LayerNorm fold, computed on the host: W' is rounded to the act dtype exactly as the
        device would (RNE) and s sums the ROUNDED values, so that mu * s cancels what the MFMA really accumulates."""
        w64, g64 = np.asarray(w, dtype=np.float64), np.asarray(gamma, dtype=np.float64)
        wq = torch.from_numpy((w64 * g64[None, :]).astype(np.float32)).to(self.act)
        s = wq.double().sum(1).float()
        bq = torch.from_numpy((np.asarray(b, dtype=np.float64) + w64 @ np.asarray(beta, dtype=np.float64)).astype(np.float32))
        return wq.contiguous().to(self.dev), s.contiguous().to(self.dev), bq.contiguous().to(self.dev)

    def _block(self, sd, p: str, fold: bool = False) -> _Block:
        g = lambda k: self._f32(sd[p + k])
        w_in, w_out, w_fc, w_proj = g("attn.in_proj_weight"), g("attn.out_proj.weight"), g("mlp.c_fc.weight"), g("mlp.c_proj.weight")
        d = w_out.shape[0]
        extra = {}
        if fold:
            extra["w_in_ln"], extra["s_in"], extra["b_in_ln"] = self._fold(
                sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
            extra["w_fc_ln"], extra["s_fc"], extra["b_fc_ln"] = self._fold(
                sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        return _Block(**extra, **dict(
            ln1_w=g("ln_1.weight"), ln1_b=g("ln_1.bias"), ln2_w=g("ln_2.weight"), ln2_b=g("ln_2.bias"),
            w_in=self._act(w_in), b_in=g("attn.in_proj_bias"), w_out=self._act(w_out), b_out=g("attn.out_proj.bias"),
            w_fc=self._act(w_fc), b_fc=g("mlp.c_fc.bias"), w_proj=self._act(w_proj), b_proj=g("mlp.c_proj.bias"),
            w_fc_t=self._act(w_fc.t()), w_proj_t=self._act(w_proj.t()), **self._oq_t(w_out, w_in[:d])))

    def _oq_t(self, w_out: torch.Tensor, w_q: torch.Tensor) -> dict:
        """The two d x d dX weights of the attention backward in ONE allocation, so that one prefetch hint
        (rpo_gemm_args.prefetch) covers both: the d out-proj operand is read first, the d q-proj operand next.
        With rpo_gemm_ws the allocation has four slots, [w_q_t | w_out_t | packed w_q_t | packed w_out_t]: the image
        tower's chain reads slots 1-2 (the attention backward folds d out-proj in and reads the row-major matrix, the
        d q-proj GEMM the packed one), a chain without the fold slots 2-3 (`_oq_hint`)."""
        d = w_out.shape[0]
        if not self.use_ws:
            both = torch.empty(2, d, d, dtype=self.act, device=self.dev)
            both[0].copy_(self._act(w_out.t()))
            both[1].copy_(self._act(w_q.t()))
            return dict(w_out_t=both[0], w_q_t=both[1], w_oq_t=both)
        both = torch.empty(4, d, d, dtype=self.act, device=self.dev)
        both[0].copy_(self._act(w_q.t()))
        both[1].copy_(self._act(w_out.t()))
        for src, dst in ((0, 2), (1, 3)):
            pw = ops.gemm_ws_pack(both[src])
            both[dst].view(-1).copy_(pw.data)
            self._wsp[(both[src].data_ptr(), (d, d))] = ops.PackedWeight(both[dst].view(-1), d, d)
        return dict(w_out_t=both[1], w_q_t=both[0], w_oq_t=both)

    def _tt(self, name: str) -> dict:
        return {"tile_config": self._text_tiles[name]} if name in self._text_tiles else {}

    def _oq_hint(self, blk: _Block, fold_out: bool, ws: bool = True) -> torch.Tensor:
        if not self.use_ws:
            return blk.w_oq_t
        if not ws:                                  # a chain on rpo_gemm_nt reads the two row-major matrices (slots 0-1)
            return blk.w_oq_t[0:2]
        return blk.w_oq_t[1:3] if fold_out else blk.w_oq_t[2:4]

    def _ws_reg(self, w: Optional[torch.Tensor]) -> None:
        """Packs a frozen row-major [N, K] weight for rpo_gemm_ws (once) and files the copy under the tensor's address."""
        if not self.use_ws or w is None or w.dim() != 2 or w.dtype != self.act or w.stride(1) != 1:
            return
        N, K = w.shape
        key = (w.data_ptr(), (N, K))
        if key in self._wsp or N % 32 != 0 or K % 64 != 0:
            return
        self._wsp[key] = ops.gemm_ws_pack(w)

    def _gemm(self, a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, **kw) -> torch.Tensor:
        """out = a @ w.T (+ epilogue) for the prompt rows: rpo_gemm_ws where the weight has a packed twin and the kernel takes
        the problem, else rpo_gemm_nt.  A prefetch hint that names a weight with a packed twin names the twin instead
        (that is what the next launch reads)."""
        pw = self._wsp.get((w.data_ptr(), tuple(w.shape))) if self.use_ws else None
        if pw is not None:
            M, split = a.shape[0], kw.get("split_k", 1)
            key = (M, pw.N, pw.K, out.dtype, epilogue, split, kw.get("ln_stats") is not None and epilogue == EPI_BIAS_RESID)
            ok = self._ws_okc.get(key)
            if ok is None:
                ok = self._ws_okc[key] = (M < 2048 and ops.gemm_ws_ok(M, pw.N, pw.K, self.act, out.dtype, epilogue, split, key[-1]))
            if ok:
                kws = dict(kw)
                if self._ws_cfg and "tile_config" not in kws:         # (the text tower's geometry: see __init__)
                    kws["tile_config"] = self._ws_cfg
                pf = kws.get("prefetch")
                if pf is not None:
                    kws["prefetch"] = self._wsp.get((pf.data_ptr(), tuple(pf.shape)), pf)
                # the cached verdict covers sizes / dtypes / epilogue / split only; what it does not see (a 96-column
                # statistics group, hi / lo residual halves, row units, a forced geometry that does not fit) makes the
                # kernel answer RPO_E_SHAPE with nothing enqueued -- then the row-major weight goes to rpo_gemm_nt
                if ops.gemm_ws_try(a, pw, out, epilogue, **kws):
                    return out
                kw.pop("tile_config", None)                            # (an rpo_gemm_ws geometry code means nothing to rpo_gemm_nt)
        return ops.gemm_nt(a, w, out, epilogue, **kw)

    def _pack(self, sd, tokens) -> None:
        cfg = self.cfg
        self.vis = [self._block(sd, f"visual.transformer.resblocks.{l}.", fold=self.fold_ln) for l in range(cfg.layers_v)]
        self.txt = [self._block(sd, f"transformer.resblocks.{l}.", fold=self.fold_ln) for l in range(cfg.layers_t)]
        # packed twins (rpo_gemm_ws) of every weight a prompt-row GEMM reads: the dX operands of both chains, the text
        # tower's forward weights (its q rows of the in-projection only) and the last image block's (prompt rows only)
        for blk in self.vis + self.txt:
            self._ws_reg(blk.w_proj_t); self._ws_reg(blk.w_fc_t)
        for blk, d in [(b, cfg.d_t) for b in self.txt] + ([(self.vis[-1], cfg.d_v)] if self.vis else []):
            for w in (blk.w_in[:d], None if blk.w_in_ln is None else blk.w_in_ln[:d], blk.w_out, blk.w_fc, blk.w_fc_ln, blk.w_proj):
                self._ws_reg(w)
        self.kpatch = _round_up(cfg.patch_dim, self.kmult)
        conv = torch.zeros(cfg.d_v, self.kpatch, device=self.dev)
        conv[:, :cfg.patch_dim] = self._f32(sd["visual.conv1.weight"]).reshape(cfg.d_v, -1)
        self.conv_w = self._act(conv)
        self.cls = self._f32(sd["visual.class_embedding"])
        self.pos = self._f32(sd["visual.positional_embedding"])
        self.ln_pre = (self._f32(sd["visual.ln_pre.weight"]), self._f32(sd["visual.ln_pre.bias"]))
        self.ln_post = (self._f32(sd["visual.ln_post.weight"]), self._f32(sd["visual.ln_post.bias"]))
        self.ln_final = (self._f32(sd["ln_final.weight"]), self._f32(sd["ln_final.bias"]))
        vp, tp = self._f32(sd["visual.proj"]), self._f32(sd["text_projection"])      # [d, e]
        self.img_proj_t, self.img_proj = self._act(vp.t()), self._act(vp)           # fwd W=[e,d]; bwd W=[d,e]
        self.text_proj_t, self.text_proj = self._act(tp.t()), self._act(tp)
        for w in (self.img_proj_t, self.img_proj, self.text_proj_t, self.text_proj):
            self._ws_reg(w)
        # small batches (config 1: batch 4): below 1024 image token rows -- B x (N + K) < 1024 -- the WHOLE image forward is a
        # small-M problem with no one-round geometry, so every block's forward weights get a packed twin too
        # (RPO_WS_SMALL=0: not; from 1024 rows on the 64x128 / 128x128 tiles win).
        self.ws_small = (self.use_ws and self.max_batch * (cfg.n_frozen + cfg.K) < 1024 and os.environ.get("RPO_WS_SMALL", "1") != "0")
        if self.ws_small:
            for blk in self.vis:
                for w in (blk.w_in, blk.w_in_ln, blk.w_out, blk.w_fc, blk.w_fc_ln, blk.w_proj):
                    self._ws_reg(w)
        # make_prompts (trainers/rpo.py:135-136): tok_emb[ids] + pos, kept for the frozen tokens only
        tx = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][None]
        self.text_x_frozen = self._f32(tx[:, :self.Lmax].reshape(cfg.n_cls * self.Lmax, cfg.d_t))
        self.text_pos = self._f32(sd["positional_embedding"][:self.Lmax])
        # the token embeddings alone (CoOp's "middle" / "front" class-token positions re-order them before the positional
        # embedding is added, trainers/coop.py:136-183)
        self.text_tok = self._f32(sd["token_embedding.weight"][tokens][:, :self.Lmax].reshape(cfg.n_cls * self.Lmax, cfg.d_t))

    # ------------------------------------------------------------------ workspace
    def _alloc(self) -> None:
        cfg, dev, act = self.cfg, self.dev, self.act
        B, N, K, dv, dt, e = self.max_batch, cfg.n_frozen, cfg.K, cfg.d_v, cfg.d_t, cfg.embed
        R, Rp = B * (N + K), B * K
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        a = lambda *s: torch.empty(*s, dtype=act, device=dev)
        Lv, Lt = cfg.layers_v, cfg.layers_t
        self.im2col = a(B * cfg.n_patches, self.kpatch)
        self.x_pre = f32(R, dv)
        # Residual stream at the layer boundaries, fp32.  INVARIANT of the hi / lo mode (16-bit modes at batch sizes with a
        # one-round geometry, the default there): after a forward pass only the PROMPT rows [B*N, B*(N+K)) of x[l] (l >= 1)
        # and xm[l] hold this pass's values -- the stream of all rows lives in h (hi) + h_lo, updated in place, and the
        # frozen rows of these tensors keep whatever an earlier call left (advisor, round 3).  Nothing in-tree reads them
        # outside forward_plain / the sibling trainers (full_last: fp32 for every row); a tool that inspects frozen rows
        # must run with RPO_NO_HILO=1 (tools/probe_alias_buffers.py does).
        self.x = [f32(R, dv) for _ in range(Lv + 1)]
        self.xm = [f32(R, dv) for _ in range(Lv)]
        self.h = a(R, dv)
        self.h_lo = a(R, dv)                         # lo half of the residual stream (hi / lo mode: _image_forward)
        self.ln_stats = f32(R, dv // 64, 2)          # per-row partial LayerNorm statistics of the tensor self.h copies
        self.qkv = [a(R, 3 * dv) for _ in range(Lv)]
        self.att = a(R, dv)
        self.g = a(R, 4 * dv)
        # saved for the QuickGELU backward, prompt rows only: fp32 pre-activations in the f32 mode, the derivative itself
        # in the act dtype in the 16-bit modes (rpo_gemm_args.aux_dtype)
        au = f32 if os.environ.get("RPO_AUX_F32") == "1" else a      # A/B switch: fp32 pre-activations in every mode
        self.u = [au(Rp, 4 * dv) for _ in range(Lv)]
        self.y_post = a(Rp, dv)
        self.img_f = f32(Rp, e)
        # backward temporaries (prompt rows)
        self.d_img_f = f32(Rp, e)
        self.d_img_f_a = a(Rp, e)
        self.dy_v = f32(max(SPLIT_FC, SPLIT_Q, 4), Rp, dv)       # (rpo_gemm_ws splits d c_fc in up to four)
        self.dxa_v, self.dxb_v = f32(Rp, dv), f32(Rp, dv)
        self.dxc_v = a(Rp, dv)
        self.du_v = a(Rp, 4 * dv)
        self.da_v, self.dq_v = a(Rp, dv), a(Rp, dv)
        # text tower (prompt rows per step)
        Rt = cfg.n_cls * K
        self.Rt = Rt
        self.xt = [f32(Rt, dt) for _ in range(Lt + 1)]
        self.xtm = [f32(Rt, dt) for _ in range(Lt)]
        self.ht = a(Rt, dt)
        self.qt = [a(Rt, dt) for _ in range(Lt)]
        self.att_t = a(Rt, dt)
        self.gt = a(Rt, 4 * dt)
        self.ut = [au(Rt, 4 * dt) for _ in range(Lt)]
        self.y_final = a(Rt, dt)
        self.text_f = f32(Rt, e)
        self.d_text_f = f32(Rt, e)
        self.d_text_f_a = a(Rt, e)
        self.ln_stats_t = f32(Rt, dt // 64, 2)
        self.dy_t = f32(max(SPLIT_FC, SPLIT_Q, 4), Rt, dt)
        self.dxa_t, self.dxb_t = f32(Rt, dt), f32(Rt, dt)
        self.dxc_t = a(Rt, dt)
        self.du_t = a(Rt, 4 * dt)
        self.da_t, self.dq_t = a(Rt, dt), a(Rt, dt)
        # frozen text K/V cache: per layer [n_cls*Lmax, 2*dt] (k | v)
        self.kv_t = [a(cfg.n_cls * self.Lmax, 2 * dt) for _ in range(Lt)]
        # head
        self.logits = f32(B, cfg.n_cls)
        self.loss = f32(1)
        self.head_ws = f32(ops.head_workspace_floats(B, cfg.n_cls, K, e))
        # parameters / gradients / momentum: one flat buffer each [text | img]
        nt, ni = K * dt, K * dv
        self.params = f32(nt + ni)
        self.grads = torch.zeros(nt + ni, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(nt + ni, dtype=torch.float32, device=dev)
        self.text_prompt = self.params[:nt].view(K, dt)
        self.img_prompt = self.params[nt:].view(K, dv)
        self.g_text = self.grads[:nt].view(K, dt)
        self.g_img = self.grads[nt:].view(K, dv)
        self.g_text_flat, self.g_img_flat = self.grads[:nt], self.grads[nt:]    # the two halves as collectives see them
        self.side = torch.cuda.Stream(device=dev)

    def hbm_bytes(self) -> int:
        tot = 0
        for v in vars(self).values():
            ts = v if isinstance(v, (list, tuple)) else [v]
            for t in ts:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    tot += t.numel() * t.element_size()
        return tot

    # ------------------------------------------------------------------ one-off text pass
    def cache_text_kv(self) -> None:
        """Frozen text tokens (positions < len_c) of every class through all blocks with the
        causal AND col<len_c mask (trainers/rpo.py:146-149); keeps each layer's K and V.
        Independent of prompts and images, so it runs once per class set."""
        cfg, act = self.cfg, self.act
        n, L, dt, H = cfg.n_cls, self.Lmax, cfg.d_t, cfg.heads_t
        Rf = n * L
        x = self.text_x_frozen.clone()
        xm = torch.empty_like(x)
        h = torch.empty(Rf, dt, dtype=act, device=self.dev)
        qkv = torch.empty(Rf, 3 * dt, dtype=act, device=self.dev)
        att = torch.empty(Rf, dt, dtype=act, device=self.dev)
        g = torch.empty(Rf, 4 * dt, dtype=act, device=self.dev)
        for l, blk in enumerate(self.txt):
            ops.layernorm_fwd(x, blk.ln1_w, blk.ln1_b, h)
            ops.gemm_nt(h, blk.w_in, qkv, EPI_BIAS, bias=blk.b_in)
            ops.text_attn_fwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], att, self.len_i32, n, L, L, H,
                              causal=True, scale=SCALE)
            self.kv_t[l].copy_(qkv[:, dt:])
            ops.gemm_nt(att, blk.w_out, xm, EPI_BIAS_RESID, bias=blk.b_out, resid=x)
            ops.layernorm_fwd(xm, blk.ln2_w, blk.ln2_b, h)
            ops.gemm_nt(h, blk.w_fc, g, EPI_BIAS_QGELU, bias=blk.b_fc, aux=None, aux_row0=Rf)
            ops.gemm_nt(g, blk.w_proj, x, EPI_BIAS_RESID, bias=blk.b_proj, resid=xm)
        self.text_x_final = x                       # output of the last block for the frozen tokens (forward_plain)
        self.plain_text_f = None
        self.text_cache_ready = True

    def _timed(self, name: str):
        return _NO_PROBE if self.probe is None else self.probe(name)

    # ------------------------------------------------------------------ forward pieces
    def _text_forward(self, train: bool) -> None:
        self._ws_cfg = self._ws_text_cfg
        try:
            self._text_forward_(train)
        finally:
            self._ws_cfg = 0

    def _text_forward_(self, train: bool) -> None:
        cfg = self.cfg
        n, K, dt, H, Rt = cfg.n_cls, cfg.K, cfg.d_t, cfg.heads_t, self.Rt
        ops.broadcast_rows(self.text_prompt, self.xt[0], n)          # trainers/rpo.py:176-177
        fold, st, last = self.fold_ln and os.environ.get("RPO_NO_TEXT_FOLD") != "1", self.ln_stats_t, len(self.txt) - 1
        pf = self._pf_chains
        nxt_q = lambda l: ((self.txt[l + 1].w_in_ln if fold else self.txt[l + 1].w_in)[:dt] if (pf and l < last) else None)
        for l, blk in enumerate(self.txt):
            kv = self.kv_t[l]
            # LayerNorm folded into the GEMMs around it exactly as in the image tower (_image_forward): the residual
            # GEMMs leave the 16-bit copy of their result in ht and its row statistics in st
            if fold and l > 0:
                self._gemm(self.ht, blk.w_in_ln[:dt], self.qt[l], EPI_LN_BIAS, bias=blk.b_in_ln[:dt], ln_stats=st,
                            ln_colsum=blk.s_in[:dt], prefetch=blk.w_out if pf else None, **self._tt("q"))
            else:
                ops.layernorm_fwd(self.xt[l], blk.ln1_w, blk.ln1_b, self.ht)
                self._gemm(self.ht, blk.w_in[:dt], self.qt[l], EPI_BIAS, bias=blk.b_in[:dt],
                            prefetch=blk.w_out if pf else None)
            ops.text_attn_fwd(self.qt[l], kv[:, :dt], kv[:, dt:], self.att_t, self.len_i32, n, K, self.Lmax, H,
                              causal=False, scale=SCALE)
            prod = dict(out2=self.ht, ln_stats=st) if fold else {}
            self._gemm(self.att_t, blk.w_out, self.xtm[l], EPI_BIAS_RESID, bias=blk.b_out, resid=self.xt[l], **prod,
                        prefetch=(blk.w_fc_ln if fold else blk.w_fc) if pf else None, **self._tt("out"))
            if fold:
                self._gemm(self.ht, blk.w_fc_ln, self.gt, EPI_LN_BIAS_QGELU, bias=blk.b_fc_ln,
                            aux=self.ut[l] if train else None, aux_row0=0, ln_stats=st, ln_colsum=blk.s_fc,
                            prefetch=blk.w_proj if pf else None, **self._tt("fc"), **self._text_fc_units)
            else:
                ops.layernorm_fwd(self.xtm[l], blk.ln2_w, blk.ln2_b, self.ht)
                self._gemm(self.ht, blk.w_fc, self.gt, EPI_BIAS_QGELU, bias=blk.b_fc,
                            aux=self.ut[l] if train else None, aux_row0=0, prefetch=blk.w_proj if pf else None)
            prod = dict(out2=self.ht, ln_stats=st) if (fold and l < last) else {}
            self._gemm(self.gt, blk.w_proj, self.xt[l + 1], EPI_BIAS_RESID, bias=blk.b_proj, resid=self.xtm[l], **prod,
                        prefetch=nxt_q(l), **self._tt("proj"))
        ops.layernorm_fwd(self.xt[-1], self.ln_final[0], self.ln_final[1], self.y_final)   # rpo.py:183
        self._gemm(self.y_final, self.text_proj_t, self.text_f, EPI_NONE)                 # rpo.py:191

    def _patch_embed(self, image: torch.Tensor) -> None:
        """conv1 as im2col + GEMM, positional embedding added in the epilogue (trainers/rpo.py:198-202): the PATCH rows of
        x_pre.  Depends on the batch alone -- not on the prompts -- so the trainer may run it for the NEXT batch while
        this step's backward is still running (RPO.step_async(next_image=...)): the only other reader of x_pre after the
        first launch of the image forward is the last LayerNorm backward, which reads its prompt rows."""
        cfg = self.cfg
        B = image.shape[0]
        R = B * (cfg.n_frozen + cfg.K)
        ops.im2col_patches(image, self.im2col[:B * cfg.n_patches], cfg.patch)
        ops.gemm_nt(self.im2col[:B * cfg.n_patches], self.conv_w, self.x_pre[:R], EPI_PATCH, resid=self.pos,
                    group=cfg.n_patches,                                               # rpo.py:198-202
                    prefetch=self.vis[0].w_in if (self._pf_chains and len(self.vis)) else None)

    def _embed_norm(self, B: int, rows: Optional[tuple] = None) -> None:
        """CLS / prompt rows (rpo.py:201-204), ln_pre (:206) and the first block's ln_1 in one launch, for all token rows or
        for `rows` = (row0, row1): the frozen rows [0, B*N) depend on the batch alone (the early patch embed forms them for
        the next batch, RPO.step_async(next_image=...)), the prompt rows [B*N, B*(N+K)) on the prompts."""
        cfg = self.cfg
        R = B * (cfg.n_frozen + cfg.K)
        ops.img_embed_norm(self.x_pre[:R], self.cls, self.pos, self.img_prompt, self.ln_pre[0], self.ln_pre[1], self.x[0][:R],
                           self.vis[0].ln1_w, self.vis[0].ln1_b, self.h[:R], B, cfg.n_frozen, cfg.K, rows=rows)

    def _patch_embed_early(self, image: torch.Tensor) -> None:
        """Everything of the image forward that does not depend on the prompts: the patch rows (_patch_embed) and the
        frozen rows' share of the embedding / ln_pre / first ln_1 launch.  What it writes -- im2col, the patch and CLS rows
        of x_pre, the frozen rows of x[0] and h -- nothing in a step's backward reads."""
        self._patch_embed(image)
        self._embed_norm(image.shape[0], rows=(0, image.shape[0] * self.cfg.n_frozen))

    def _image_forward(self, image: torch.Tensor, train: bool, full_last: bool = False, patch_done: bool = False) -> None:
        cfg = self.cfg
        B, N, K, dv, H = image.shape[0], cfg.n_frozen, cfg.K, cfg.d_v, cfg.heads_v
        Rf, Rp = B * N, B * K
        R = Rf + Rp
        x_pre = self.x_pre[:R]
        if not patch_done:                      # (patch_done: _patch_embed(image) has already been enqueued for this batch)
            self._patch_embed(image)
        h, att, g = self.h[:R], self.att[:R], self.g[:R]
        # CLS / prompt rows (rpo.py:201-204), ln_pre (:206) and the first block's ln_1 in one launch; behind an early
        # patch embed only the prompt rows are left
        self._embed_norm(B, rows=(Rf, R) if (patch_done and K > 0) else None)
        fold = self.fold_ln
        # Row units of the whole-stream GEMMs: one image = N frozen + K prompt rows.  With them c_fc tiles by image
        # (saved pre-activations spread over all workgroups) and out-proj / c_proj run as one round of 224x96 tiles,
        # whose row statistics are over 96 columns instead of 64 (rpo_gemm_args.ln_group): grp says which.
        units = (N, K, Rf)
        which = os.environ.get("RPO_RESID_UNITS", "all")            # A/B switch: all | c_proj | none
        # SPLIT mode (experiment hook, rpo_amd/experimental.py: RPO_SPLIT): the frozen rows on the one-round kernels -- row
        # units of N + 0 rows, rows [0, Rf) -- and the B*K prompt rows on their own launches, as the last block's always are.
        # Rows never mix inside a GEMM, so results are those of the whole-stream launches up to the tile shapes' summation
        # order.  Measured 3 % SLOWER at K = 48 (profiles/r04_ab_split_k48.txt); never taken by this class (_split_rows).
        split = self._split_rows(B)
        # small batches: every "wide" launch is a small-M GEMM and goes to rpo_gemm_ws (no row units, 64-column statistics)
        # (measured, tools/ab_env.py --extra "--batch B": B = 4 step 1.509 vs 1.554 ms, B = 8 1.958 vs 1.723 ms -- from
        #  1024 rows on the 64x128 / 128x128 tiles win; the unmasked towers (full_last: zero-shot / CoOp) keep rpo_gemm_nt)
        small = self.ws_small and R < 1024 and not split and not full_last
        if small:
            units = None
        Mw = Rf if split else R                                       # rows of the whole-batch ("wide") launches
        if split:
            units = (N, 0, Rf)
        u_out, u_proj = (units if which == "all" else None), (units if which in ("all", "c_proj") else None)
        if small:
            u_out = u_proj = None
        grps = self._stats_group.get((B, split, small))
        if grps is None:
            grps = self._stats_group[(B, split, small)] = ((ops.gemm_stats_group(Mw, dv, dv, self.act, u_out),
                                                     ops.gemm_stats_group(Mw, dv, 4 * dv, self.act, u_proj)) if fold else (64, 64))
        g_out, g_proj = grps                 # statistics written by out-proj (read by c_fc) / by c_proj (read by in-proj)
        # Residual stream as 16-bit hi / lo halves (rpo_gemm_args.resid_hi ...): where both residual GEMMs of a block run on
        # the one-round row-unit kernel, the stream of the whole-batch blocks lives in h (hi = the 16-bit copy the next
        # GEMM consumes anyway) + h_lo, updated in place; the fp32 tensors x[l] / xm[l] are then written for the prompt
        # rows only (what the backward and ln_post read).  -10.9 MB of stores per residual GEMM.  Not when the last
        # block's frozen rows are wanted in fp32 (full_last: forward_plain reads the CLS rows).  A/B: RPO_NO_HILO=1.
        hilo = (fold and not full_last and u_out is not None and u_proj is not None and len(self.vis) > 1
                and os.environ.get("RPO_NO_HILO") != "1"
                and ops.gemm_hilo_ok(Mw, dv, dv, self.act, u_out, g_out) and ops.gemm_hilo_ok(Mw, dv, 4 * dv, self.act, u_proj, g_proj))
        h_lo = self.h_lo[:R]
        # RPO_RESID16=1: the stream as the 16-bit hi half ALONE (no lo half read or written: 22 MB less per residual GEMM
        # again; fp16 mode: step -1.1 %).  In the fp16 mode that is what the reference's own PREC: fp16 run keeps
        # (clip/model.py:379-400 converts the model to fp16, LayerNorm casts its fp32 result back, :153-159).  Against the
        # reference's fp32 outputs at B = 32: fp16 logits 7.6e-3 (K = 24) / 1.2e-2 (K = 4) against 7.1e-3 / < 1e-2 with both
        # halves, bf16 7.5e-2 against 6.1e-2 -- outside the fp16 mode's 1e-2 bound at K = 4, so it is an opt-in with its own
        # tolerance row (tests/test_gpu_model.py: RESID16_TOL), not a default.
        if hilo and os.environ.get("RPO_RESID16") == "1":
            h_lo = None
        # per-row partial statistics live in ONE buffer: rows written with g-column groups start at float offset
        # row * (dv / g) * 2, so the frozen rows' 96-column records ([0, Rf * 16)) and the prompt rows' 64-column records
        # ([Rf * 24, R * 24)) of the split mode / the last block never overlap
        stv = lambda grp: self.ln_stats.view(-1)[:R * (dv // grp) * 2].view(R, dv // grp, 2)
        st_out, st_proj, st64 = stv(g_out), stv(g_proj), self.ln_stats[:R]
        last = len(self.vis) - 1
        # Prefetch hint (rpo_gemm_args.prefetch): every big GEMM names the frozen weight matrix the NEXT GEMM of the chain
        # reads -- a whole step after its last use no cache holds it, and with two LDS stages a k-loop cannot cover an HBM
        # round trip per k-tile; touched a kernel ahead, the lines wait in the memory-side cache.  A/B: RPO_NO_WPREFETCH=1.
        pf = os.environ.get("RPO_NO_WPREFETCH") != "1" and self.act != torch.float32
        # Every kernel names its successor's weights.  Other assignments measured same-box (profiles/README.md): naming the
        # matrix two kernels ahead (B), or hosting nothing in c_fc (Bp / C / D), 2.92-2.95 ms against 2.908 ms; off 3.03 ms.
        table = dict(in_="out", out="fc", fc="proj", proj="in+")

        def pf_of(host: str, l: int):
            what = table[host] if pf else None
            if what is None:
                return None
            if what.endswith("+"):
                if l >= last:
                    return None
                b, what = self.vis[l + 1], what[:-1]
            else:
                b = self.vis[l]
            return {"out": b.w_out, "proj": b.w_proj, "fc": b.w_fc_ln if fold else b.w_fc,
                    "in": b.w_in_ln if (fold and b is not self.vis[0]) else b.w_in}[what]
        no_probe = lambda name: _NO_PROBE
        for l, blk in enumerate(self.vis):
            x, xm, xo, qkv = self.x[l][:R], self.xm[l][:R], self.x[l + 1][:R], self.qkv[l][:R]
            # ln_1 + in-proj.  Folded (l > 0): h already holds the 16-bit copy of x[l] and st its row statistics, both
            # left by the previous block's c_proj; the GEMM epilogue applies the normalisation (RPO_EPI_LN_BIAS).
            folded_in = fold and l > 0
            if not folded_in and l > 0:                   # (block 0: h was written with x[0] above)
                ops.layernorm_fwd(x, blk.ln1_w, blk.ln1_b, h)
            w_in, b_in = (blk.w_in_ln, blk.b_in_ln) if folded_in else (blk.w_in, blk.b_in)
            epi_in = EPI_LN_BIAS if folded_in else EPI_BIAS
            # statistics the previous block's c_proj left: over g_proj-column groups where it ran on the wide launch, over 64
            # where the prompt rows had their own (split mode)
            lnk = lambda r0, r1, c0, c1, own=False: (dict(ln_stats=(st64 if own else st_proj)[r0:r1], ln_colsum=blk.s_in[c0:c1],
                                                          ln_group=64 if own else g_proj) if folded_in else {})
            if l < last or full_last:
                if split:
                    with self._timed("in_proj"):
                        ops.gemm_nt(h[:Rf], w_in, qkv[:Rf], epi_in, bias=b_in, **lnk(0, Rf, 0, 3 * dv), prefetch=pf_of("in_", l))
                    # K/V of prompt rows are never read (visual mask, rpo.py:154-156): their q alone
                    ops.gemm_nt(h[Rf:], w_in[:dv], qkv[Rf:, :dv], epi_in, bias=b_in[:dv], **lnk(Rf, R, 0, dv, own=True))
                elif small:
                    # (rpo_gemm_ws has no tile skipping: the K / V of the few prompt rows are computed and never read)
                    with self._timed("in_proj"):
                        self._gemm(h, w_in, qkv, epi_in, bias=b_in, **lnk(0, R, 0, 3 * dv), prefetch=pf_of("in_", l))
                else:
                    # K/V of prompt rows are never read (visual mask, rpo.py:154-156): skip those tiles
                    with self._timed("in_proj"):
                        ops.gemm_nt(h, w_in, qkv, epi_in, bias=b_in, skip_row0=Rf, skip_col0=dv, **lnk(0, R, 0, 3 * dv),
                                    prefetch=pf_of("in_", l))
                with self._timed("attn_fwd"):
                    ops.attn_readonly_fwd(qkv[:, :dv], qkv[:, dv:2 * dv], qkv[:, 2 * dv:], att, B, H, N, K, SCALE)
                segs = [(0, Rf, True), (Rf, R, False)] if split else [(0, R, True)]
            else:
                # Last block: only its K prompt rows are consumed (ln_post reads x[:, -K:], rpo.py:210; the CLS feature
                # i_f of :211 is dead code), and no later block reads the frozen rows.  So the frozen rows contribute
                # their K / V and nothing else: q and everything after attention run on the B*K prompt rows only.
                ops.gemm_nt(h[:Rf], w_in[dv:], qkv[:Rf, dv:], epi_in, bias=b_in[dv:], **lnk(0, Rf, dv, 3 * dv))
                self._gemm(h[Rf:], w_in[:dv], qkv[Rf:, :dv], epi_in, bias=b_in[:dv], **lnk(Rf, R, 0, dv, own=split))
                ops.attn_readonly_fwd(qkv[:, :dv], qkv[:, dv:2 * dv], qkv[:, 2 * dv:], att, B, H, N, K, SCALE,
                                      q_first=N)
                segs = [(Rf, R, False)]
            # the rows after attention: ONE wide launch per GEMM (rows [0, R), or [0, Rf) in the split mode) on the
            # row-unit kernels, and / or the prompt rows [Rf, R) on plain tiles with 64-column statistics
            for lo, hi, wide in segs:
                timed = self._timed if wide else no_probe
                un_o, go, so = (u_out, g_out, st_out) if wide else (None, 64, st64)
                un_p, gp, sp = (u_proj, g_proj, st_proj) if wide else (None, 64, st64)
                un = units if wide else None
                has_prompt = hi > Rf                                  # rows whose pre-activations the backward needs
                aux = self.u[l][:Rp] if (train and has_prompt) else None
                aux_row0 = max(Rf - lo, 0) if has_prompt else hi - lo
                # out-proj + residual; folded: it also leaves the 16-bit copy of xm in h and its row statistics in st
                prod = dict(out2=h[lo:hi], ln_stats=so[lo:hi], ln_group=go) if fold else {}
                # hi / lo stream: block 0's out-proj still reads the fp32 x[0] (img_embed_norm wrote it) and starts the halves
                hl = hilo and wide
                res_o = dict(resid_hi=h[:hi], resid_lo=None if h_lo is None else h_lo[:hi]) if (hl and l > 0) else dict(resid=x[lo:hi])
                if hl:
                    prod.update(out_lo=None if h_lo is None else h_lo[:hi], c_row0=Rf)             # (h_lo None: the hi half alone)
                # (prompt rows alone -- the last block, the split mode -- go to rpo_gemm_ws where the weight has a packed twin)
                gemm = ops.gemm_nt if (wide and not small) else self._gemm
                with timed("out_proj"):
                    gemm(att[lo:hi], blk.w_out, xm[lo:hi], EPI_BIAS_RESID, bias=blk.b_out, row_units=un_o,
                         **res_o, **prod, prefetch=pf_of("out", l) if wide else None)
                # c_proj + residual; folded: copy + statistics of x[l+1] for the next block's in-proj
                prod_p = dict(out2=h[lo:hi], ln_stats=sp[lo:hi], ln_group=gp) if (fold and l < last) else {}
                res_p = dict(resid_hi=h[:hi], resid_lo=None if h_lo is None else h_lo[:hi]) if hl else dict(resid=xm[lo:hi])
                if hl:
                    prod_p.update(out_lo=None if h_lo is None else h_lo[:hi], c_row0=Rf)
                proj_kw = dict(a=g[lo:hi], w=blk.w_proj, out=xo[lo:hi], epilogue=EPI_BIAS_RESID, bias=blk.b_proj, row_units=un_p,
                               **res_p, **prod_p, prefetch=pf_of("proj", l) if wide else None)
                if not wide or small:         # (64-column statistics either way; no row units)
                    proj_kw.pop("row_units"); proj_kw.pop("ln_group", None)
                # experiment hook (rpo_amd/experimental.py: RPO_MLP_FUSED): c_fc -> c_proj of a whole-batch block as ONE
                # launch; None in this class
                if fold and wide and self._fused_mlp is not None:
                    fc_kw = dict(a=h[lo:hi], w=blk.w_fc_ln, out=g[lo:hi], epilogue=EPI_LN_BIAS_QGELU, bias=blk.b_fc_ln,
                                 aux=aux, aux_row0=aux_row0, ln_stats=so[lo:hi], ln_colsum=blk.s_fc, row_units=un,
                                 ln_group=go, prefetch=pf_of("fc", l))
                    with timed("c_fc"):
                        done = self._fused_mlp(fc_kw, proj_kw)
                    if done:
                        continue
                if fold:
                    with timed("c_fc"):
                        gemm(h[lo:hi], blk.w_fc_ln, g[lo:hi], EPI_LN_BIAS_QGELU, bias=blk.b_fc_ln,
                             aux=aux, aux_row0=aux_row0,
                             ln_stats=so[lo:hi], ln_colsum=blk.s_fc, row_units=un, ln_group=go,
                             prefetch=pf_of("fc", l) if wide else None)
                else:
                    with timed("ln_2"):
                        ops.layernorm_fwd(xm[lo:hi], blk.ln2_w, blk.ln2_b, h[lo:hi])
                    with timed("c_fc"):
                        gemm(h[lo:hi], blk.w_fc, g[lo:hi], EPI_BIAS_QGELU, bias=blk.b_fc,
                             aux=aux, aux_row0=aux_row0, row_units=un)
                with timed("c_proj"):
                    if wide and not small:
                        ops.gemm_nt(**proj_kw)
                    else:
                        pk = dict(proj_kw)
                        self._gemm(pk.pop("a"), pk.pop("w"), pk.pop("out"), pk.pop("epilogue"), **pk)
        ops.layernorm_fwd(self.x[-1][Rf:R], self.ln_post[0], self.ln_post[1], self.y_post[:Rp])   # rpo.py:210
        self._gemm(self.y_post[:Rp], self.img_proj_t, self.img_f[:Rp], EPI_NONE)

    # ------------------------------------------------------------------ experiment hooks (overridden in experimental.py)
    _fused_mlp = None                               # callable(fc_kw, proj_kw) -> bool: c_fc -> c_proj as one launch

    def _split_rows(self, B: int) -> bool:
        """Frozen rows and prompt rows of the image forward as separate launches?  Never, in the product engine."""
        return False

    def joint_backward_ok(self, B: Optional[int] = None) -> bool:
        """Both backward chains as one chain of paired launches?  Never, in the product engine."""
        return False

    def _bwd_fold_limits(self) -> tuple:
        """(largest K, widths) for which the attention backward folds the d out-proj GEMM in."""
        return 64, (512, 768, 1024)

    # ------------------------------------------------------------------ backward pieces
    def _rows_backward(self, blocks: List[_Block], x: List[torch.Tensor], xm: List[torch.Tensor],
                       u: List[torch.Tensor], dxa, dxb, dxc, du, da, dq, dy, attn_bwd, fold_out: bool = False,
                       tt: Optional[dict] = None) -> torch.Tensor:
        """Shared by both towers: dx (fp32, in dxa) holds dL/d(block output) on entry; on return the
        tensor holding dL/d(block-0 input).  dxc mirrors dx in the act dtype (GEMM A operand)."""
        pf = self._pf_chains
        tc = lambda n: ({"tile_config": tt[n]} if (tt and n in tt) else {})       # (A/B knob: RPO_WS_TEXT_TILES)
        # split-K factors of the two fp32-output dX GEMMs (slabs summed in fixed order by rpo_layernorm_bwd).  On
        # rpo_gemm_ws the waves of a workgroup already split k four ways: d q-proj runs unsplit (one slab less for the
        # LayerNorm backward to read) and d c_fc in TWO.  Alone, four slabs are the faster launch at 768 rows (96x96 tiles
        # x 4 = one round of the CUs: 9.0 vs 10.7 us, profiles/r05_bench_gemm_ws.txt), but in the step two slabs win
        # (-0.85 % step time, three alternating pairs, profiles/r05_ab_ws_splits.txt): half the workgroups beside the text
        # tower and half the slab bytes for the LayerNorm backward behind it.
        # (whether rpo_gemm_ws really takes this chain: _gemm sends M >= 2048 rows -- the text tower from 86 classes on at
        #  K = 24, the image tower from batch 86 on -- to rpo_gemm_nt's tiles, which keep their own step-tuned split factors
        #  and read the row-major d x d operands)
        ws = self.use_ws and dxa.shape[0] < 2048
        s_fc, s_q = (SPLIT_FC, SPLIT_Q) if not ws else (2, 1)
        if dxa.shape[0] >= 2048:
            # a chain of thousands of rows (the text tower from 86 classes on at K = 24: 24 000 rows at ImageNet's 1000) has
            # hundreds of output tiles by itself: split-K would only write and re-read fp32 slabs (49 MB each at 24 000 rows)
            s_fc = s_q = 1
        if ws and "RPO_WS_SPLITS" in os.environ:             # A/B: "FCxQ" (step-level tuning, tools/ab_env.py)
            s_fc, s_q = (int(v) for v in os.environ["RPO_WS_SPLITS"].split("x"))
        for l in reversed(range(len(blocks))):
            blk = blocks[l]
            a_in = dxa if self.act == torch.float32 else dxc
            # prefetch hints: every dX GEMM names the weights the chain reads next (d c_proj -> w_fc_t; d c_fc -> the two
            # d x d operands of the attention backward; d q-proj -> the next block's w_proj_t)
            # (measured and dropped, same-box: a second hint range for what the backward re-reads from the forward pass --
            #  the frozen rows' K / V of the block, 29 MB, named by this GEMM for the attention backward two kernels on:
            #  step 1.8 % SLOWER, 3.045 vs 2.992 ms; the saved QuickGELU operand u[l-1], named by the d q-proj GEMM: no
            #  effect, 2.875 vs 2.871 ms)
            self._gemm(a_in, blk.w_proj_t, du, EPI_QGELU_BWD, aux=u[l], prefetch=blk.w_fc_t if pf else None, **tc("dproj"))  # d c_proj, d QuickGELU
            dyf = dy[0] if s_fc == 1 else dy[:s_fc]
            self._gemm(du, blk.w_fc_t, dyf, EPI_NONE, split_k=s_fc,
                       prefetch=self._oq_hint(blk, fold_out, ws) if pf else None, **tc("dfc"))    # d c_fc
            ops.layernorm_bwd(dyf, xm[l], blk.ln2_w, dxa, dxb,
                              None if self.act == torch.float32 else dxc)
            a_in = dxb if self.act == torch.float32 else dxc
            if fold_out:
                attn_bwd(l, a_in, dq)                                             # d out_proj inside the attention kernel
            else:
                self._gemm(a_in, blk.w_out_t, da, EPI_NONE)                       # d out_proj
                attn_bwd(l, da, dq)
            dyq = dy[0] if s_q == 1 else dy[:s_q]
            self._gemm(dq, blk.w_q_t, dyq, EPI_NONE, split_k=s_q,
                       prefetch=blocks[l - 1].w_proj_t if (pf and l > 0) else None, **tc("dq"))   # d q-projection
            ops.layernorm_bwd(dyq, x[l], blk.ln1_w, dxb, dxa,
                              None if self.act == torch.float32 else dxc)
        return dxa

    def _image_backward(self, B: int) -> None:
        self._image_backward_rows(B, 0, B)
        self._image_backward_finish(B)

    def _image_backward_rows(self, B: int, b0: int, b1: int) -> None:
        """The image tower's prompt-row chain for images [b0, b1) of a batch of B: every back-propagated row depends only
        on rows of its own image (the K / V it reads belong to frozen tokens), so any split of the batch gives independent
        chains over disjoint row ranges of the same buffers (RPO.step_async runs them on separate streams:
        RPO_BWD_PARTS).  Leaves dL/d(ln_pre input) of those rows in dxb_v; _image_backward_finish sums over the batch."""
        cfg = self.cfg
        N, K, dv, H = cfg.n_frozen, cfg.K, cfg.d_v, cfg.heads_v
        Rf = B * N
        r0, r1 = b0 * K, b1 * K                          # prompt-row range of the part
        f0, f1 = b0 * N, b1 * N                          # its images' frozen rows (keys / values)
        nb = b1 - b0
        dxa, dxb, dxc = self.dxa_v[r0:r1], self.dxb_v[r0:r1], self.dxc_v[r0:r1]
        if self.act == torch.float32:
            d_f = self.d_img_f[r0:r1]
        else:
            d_f = self.d_img_f_a[r0:r1]                # written by the head's backward
        self._gemm(d_f, self.img_proj, self.dy_v[0, r0:r1], EPI_NONE,
                   prefetch=self.vis[-1].w_proj_t if self._pf_chains else None)
        ops.layernorm_bwd(self.dy_v[0, r0:r1], self.x[-1][Rf + r0:Rf + r1], self.ln_post[0], None, dxa,
                          None if self.act == torch.float32 else dxc)

        # 16-bit modes, K <= 64, d = 512 / 768 / 1024: the d out-proj GEMM runs inside the attention backward kernel
        # (round 4: d = 1024 -- ViT-L/14 -- and K in (32, 64] -- one workgroup per 32-query tile)
        kmax, widths = self._bwd_fold_limits()
        fold_out = (self.act != torch.float32 and K <= kmax and dv in widths and os.environ.get("RPO_NO_BWD_FOLD") != "1")

        def attn_bwd(l, da, dq):
            qkv = self.qkv[l]
            if fold_out:
                with self._timed("attn_bwd_proj"):
                    ops.attn_readonly_bwd_proj(qkv[Rf + r0:Rf + r1, :dv], qkv[f0:f1, dv:2 * dv], qkv[f0:f1, 2 * dv:], da,
                                               self.vis[l].w_out_t, dq, nb, H, N, K, SCALE)
            else:
                ops.attn_readonly_bwd(qkv[Rf + r0:Rf + r1, :dv], qkv[f0:f1, dv:2 * dv], qkv[f0:f1, 2 * dv:], da, dq,
                                      nb, H, N, K, SCALE)

        dx = self._image_chain(B, b0, b1, attn_bwd, fold_out)
        # through ln_pre (rpo.py:206) to the appended prompt rows
        ops.layernorm_bwd(dx, self.x_pre[Rf + r0:Rf + r1], self.ln_pre[0], None, dxb)

    def _image_chain(self, B: int, b0: int, b1: int, attn_bwd, fold_out: bool) -> torch.Tensor:
        """The 6 x layers stages of the image tower's prompt-row chain for images [b0, b1), one launch per stage
        (experiment hook: rpo_amd/experimental.py runs them as ONE persistent launch under RPO_CHAIN=1)."""
        K, Rf = self.cfg.K, B * self.cfg.n_frozen
        r0, r1 = b0 * K, b1 * K
        return self._rows_backward(self.vis, [t[Rf + r0:Rf + r1] for t in self.x[:-1]], [t[Rf + r0:Rf + r1] for t in self.xm],
                                   [t[r0:r1] for t in self.u], self.dxa_v[r0:r1], self.dxb_v[r0:r1], self.dxc_v[r0:r1],
                                   self.du_v[r0:r1], self.da_v[r0:r1], self.dq_v[r0:r1], self.dy_v[:, r0:r1], attn_bwd,
                                   fold_out=fold_out)

    def _image_backward_finish(self, B: int) -> None:
        """sum over the batch (.repeat, rpo.py:204)"""
        ops.reduce_groups(self.dxb_v[:B * self.cfg.K], self.g_img, B)

    def _text_backward(self) -> None:
        self._ws_cfg = self._ws_text_cfg_bwd
        try:
            self._text_backward_()
        finally:
            self._ws_cfg = 0

    def _text_backward_(self) -> None:
        cfg = self.cfg
        n, K, dt, H, Rt = cfg.n_cls, cfg.K, cfg.d_t, cfg.heads_t, self.Rt
        dxa, dxb, dxc = self.dxa_t, self.dxb_t, self.dxc_t
        d_f = self.d_text_f if self.act == torch.float32 else self.d_text_f_a      # written by the head's backward
        self._gemm(d_f, self.text_proj, self.dy_t[0], EPI_NONE,
                   prefetch=self.txt[-1].w_proj_t if self._pf_chains else None)
        ops.layernorm_bwd(self.dy_t[0], self.xt[-1], self.ln_final[0], None, dxa,
                          None if self.act == torch.float32 else dxc)

        dx = self._text_chain()
        ops.reduce_groups(dx, self.g_text, n)            # same prompt row written into every class

    def _text_chain(self) -> torch.Tensor:
        """The 6 x layers stages of the text tower's prompt-row chain, one launch per stage (experiment hooks in
        rpo_amd/experimental.py: RPO_CHAIN_TEXT -- one persistent launch; RPO_TEXT_BWD_FOLD -- the MFMA attention backward
        with the d out-proj GEMM folded in)."""
        cfg = self.cfg
        n, K, dt, H = cfg.n_cls, cfg.K, cfg.d_t, cfg.heads_t

        def attn_bwd(l, da, dq):
            kv = self.kv_t[l]
            ops.text_attn_bwd(self.qt[l], kv[:, :dt], kv[:, dt:], da, dq, self.len_i32, n, K, self.Lmax, H, SCALE)

        return self._rows_backward(self.txt, self.xt[:-1], self.xtm, self.ut, self.dxa_t, self.dxb_t, self.dxc_t, self.du_t,
                                   self.da_t, self.dq_t, self.dy_t, attn_bwd, fold_out=False, tt=self._text_tiles)

    # ------------------------------------------------------------------ public
    def forward_eval(self, image: torch.Tensor, use_graph: bool = True) -> torch.Tensor:
        """logits[B, n_cls] (trainers/rpo.py:232).  The image tower + head of a batch size are captured in a HIP graph
        on first use and replayed afterwards: launched eagerly the ~100 kernels of this path are launch-bound
        (round 1: 6.4 ms per 100 images)."""
        B = self._check(image)
        main = torch.cuda.current_stream()
        # text features depend on the prompts only: in evaluation they are computed once, not per batch
        # (the reference recomputes the text tower for every test batch)
        text_stale = self.text_f_version != self.params_version
        if text_stale:
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._text_forward(train=False)
            main.wait_stream(self.side)
            self.text_f_version = self.params_version
        if not use_graph:
            self._eval_body(image, B)
            return self.logits[:B]
        entry = self._eval_graphs.get(B)
        if entry is None:
            static = torch.empty_like(image)
            static.copy_(image)
            self._eval_body(static, B)                  # eager warm-up: sets kernel attributes (not capturable)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._eval_body(static, B)
            entry = self._eval_graphs[B] = (g, static)
        g, static = entry
        if static.data_ptr() != image.data_ptr():
            static.copy_(image, non_blocking=True)
        g.replay()
        return self.logits[:B]

    def _eval_body(self, image: torch.Tensor, B: int) -> None:
        self._image_forward(image, train=False)
        K, e = self.cfg.K, self.cfg.embed
        ops.head_fwd_bwd(self.img_f[:B * K].view(B, K, e), self.text_f.view(self.cfg.n_cls, K, e), None,
                         self.logit_scale_exp, self.logits[:B], None, None, None, self.head_ws)

    def _head_act(self, B: int) -> dict:
        """16-bit modes: the head's backward also leaves the act-dtype copies of the two feature gradients that the dX GEMMs
        of the projections read (no convert launches at the head of the backward chains)."""
        if self.act == torch.float32:
            return {}
        return dict(d_img_f_act=self.d_img_f_a[:B * self.cfg.K], d_text_f_act=self.d_text_f_a)

    def set_context(self, ctx) -> None:
        """CoOp's learned context (trainers/coop.py:117-134, generic context, class token at the end): the input
        embeddings of positions 1 .. n_ctx of every class become ctx + positional embedding (TextEncoder.forward,
        :47-58).  Affects forward_plain (and everything that reads the frozen-token pass) from the next call on."""
        c = torch.as_tensor(np.asarray(ctx, dtype=np.float32), device=self.dev)
        n_ctx = c.shape[0]
        assert c.ndim == 2 and c.shape[1] == self.cfg.d_t and 1 + n_ctx < self.Lmax, "ctx: [n_ctx, d_t], 1 + n_ctx < prompt length"
        x = self.text_x_frozen.view(self.cfg.n_cls, self.Lmax, self.cfg.d_t)
        x[:, 1:1 + n_ctx] = (c + self.text_pos[1:1 + n_ctx])[None]
        self.text_cache_ready = False
        self.plain_text_f = None
        self.text_f_version = -1

    def forward_plain(self, image: torch.Tensor) -> torch.Tensor:
        """Plain CLIP inference, logits[B, n_cls] = CLIP.forward(image, tokens) (clip/model.py:344-372): what
        trainers/zsclip.py:58-63 computes, and the towers the sibling trainers call (trainers/coop.py:196-208).
        No token reads a prompt (section 2 of DESIGN.md), so the frozen rows of this engine ARE the unmasked towers:
        the image feature is ln_post + proj of the CLS row of a complete last block, the text feature the EOT row of the
        frozen-token pass that fills the K / V cache.  The prompts play no part."""
        cfg = self.cfg
        B = self._check(image)
        N, dv, e, n = cfg.n_frozen, cfg.d_v, cfg.embed, cfg.n_cls
        if not self.text_cache_ready:
            self.cache_text_kv()
        if self.plain_text_f is None:
            rows = torch.arange(n, device=self.dev) * self.Lmax + (self.len_i32.to(torch.int64) - 1)   # EOT positions
            eot = self.text_x_final.index_select(0, rows).contiguous()
            y = torch.empty(n, cfg.d_t, dtype=self.act, device=self.dev)
            ops.layernorm_fwd(eot, self.ln_final[0], self.ln_final[1], y)
            self.plain_text_f = torch.empty(n, e, dtype=torch.float32, device=self.dev)
            ops.gemm_nt(y, self.text_proj_t, self.plain_text_f, EPI_NONE)
        self._image_forward(image, train=False, full_last=True)
        cls_rows = self.x[-1][:B * N].view(B, N, dv)[:, 0, :]                  # [B, dv], row stride N * dv
        ops.layernorm_fwd(cls_rows, self.ln_post[0], self.ln_post[1], self.y_post[:B])
        if self.img_cls_f is None:
            self.img_cls_f = torch.empty(self.max_batch, e, dtype=torch.float32, device=self.dev)
        ops.gemm_nt(self.y_post[:B], self.img_proj_t, self.img_cls_f[:B], EPI_NONE)
        ops.head_fwd_bwd(self.img_cls_f[:B].view(B, 1, e), self.plain_text_f.view(n, 1, e), None, self.logit_scale_exp,
                         self.logits[:B], None, None, None, self.head_ws)
        return self.logits[:B]

    def forward_backward(self, image: torch.Tensor, label: torch.Tensor) -> None:
        """Enqueue loss + both prompt gradients (trainers/rpo.py:229-230, :308).  Results land in
        self.loss, self.logits, self.grads (= [g_text | g_img]).  Capturable in a HIP graph."""
        B = self._check(image)
        self.text_f_version = -1
        assert label.dtype == torch.int64 and label.shape == (B,)
        K, e, n = self.cfg.K, self.cfg.embed, self.cfg.n_cls
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self._text_forward(train=True)
        self._image_forward(image, train=True)
        main.wait_stream(self.side)
        ops.head_fwd_bwd(self.img_f[:B * K].view(B, K, e), self.text_f.view(n, K, e), label, self.logit_scale_exp,
                         self.logits[:B], self.loss, self.d_img_f[:B * K].view(B, K, e),
                         self.d_text_f.view(n, K, e), self.head_ws, **self._head_act(B))
        if self.joint_backward_ok(B):
            self._joint_backward(B)
            return
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self._text_backward()
        self._image_backward(B)
        main.wait_stream(self.side)

    def head(self, B: int, label: torch.Tensor) -> None:
        K, e, n = self.cfg.K, self.cfg.embed, self.cfg.n_cls
        ops.head_fwd_bwd(self.img_f[:B * K].view(B, K, e), self.text_f.view(n, K, e), label, self.logit_scale_exp,
                         self.logits[:B], self.loss, self.d_img_f[:B * K].view(B, K, e),
                         self.d_text_f.view(n, K, e), self.head_ws, **self._head_act(B))

    def _check(self, image: torch.Tensor) -> int:
        cfg = self.cfg
        assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous()
        # kernels are enqueued on the CURRENT device's streams (ops._stream): refuse a foreign device instead of
        # launching GPU-0 kernels on GPU-1 pointers
        assert image.device == self.dev and torch.cuda.current_device() == self.dev.index, \
            f"engine lives on {self.dev}: make it the current device and pass tensors that live there"
        B = image.shape[0]
        assert 1 <= B <= self.max_batch and tuple(image.shape[1:]) == (3, cfg.image_size, cfg.image_size)
        if not self.text_cache_ready:
            self.cache_text_kv()
        return B


def make_engine(*args, **kw) -> Engine:
    """The engine a model / trainer is built on: `Engine`, or -- only under RPO_EXPERIMENTAL=1, which also selects the
    -DRPO_EXPERIMENTAL build of the library -- its subclass with the measured-slower experiments (rpo_amd/experimental.py)."""
    from ._lib import EXPERIMENTAL
    if EXPERIMENTAL:
        from .experimental import ExperimentalEngine
        return ExperimentalEngine(*args, **kw)
    return Engine(*args, **kw)
