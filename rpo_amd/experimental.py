"""The measured-slower EXPERIMENTS of rounds 3 / 4 on the engine (DESIGN.md sections 11c / 11d / 15): kept for reproduction,
compiled only into the -DRPO_EXPERIMENTAL build of the library (include/rpo_amd_experimental.h) and reached only under
RPO_EXPERIMENTAL=1 -- `rpo_amd.engine.make_engine` returns this subclass then and the plain `Engine` otherwise, so the
default train step, the eval path, the sibling trainers and bench.py import and execute none of this file.

  RPO_CHAIN=1 (+ RPO_CHAIN_TEXT=1, RPO_CHAIN_IMAGE=0)   a backward chain as ONE persistent launch (rpo_chain_bwd)
  RPO_MLP_FUSED=1|safe                                  c_fc -> c_proj as one launch (rpo_mlp_fused)
  RPO_JOINT_BWD=1                                       both backward chains as paired launches (rpo_*_pair)
  RPO_TEXT_BWD_FOLD=1                                   text attention backward on the MFMA kernel, d out-proj folded in
  RPO_SPLIT=1|force                                     frozen rows on one-round kernels, prompt rows on their own launches
  RPO_BWD_FOLD_R3=1                                     round-3 coverage of the folded d out-proj
(RPO_BWD_PARTS lives in rpo_amd/trainer.py: it re-orders graph replays, not kernels.)"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops
from ._lib import EPI_NONE, EPI_QGELU_BWD, xenv as _xenv
from .engine import SCALE, SPLIT_FC, SPLIT_Q, Engine


class ExperimentalEngine(Engine):
    def _alloc(self) -> None:
        super()._alloc()
        cfg, dev = self.cfg, self.dev
        B, K = self.max_batch, cfg.K
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.mlp_counters = torch.zeros(B + 1, dtype=torch.int32, device=dev)   # rpo_mlp_fused: one per row unit + give-ups, never reset
        # the persistent backward chain (rpo_chain_bwd): 4 k-slice slabs, its scratch (one per tower: the two chains run
        # concurrently), optional stage timeline (tools/prof_chain.py sets it)
        self.dy4_v = f32(4, B * K, cfg.d_v)
        self.dy4_t = f32(4, self.Rt, cfg.d_t)
        self.chain_state_v = torch.zeros(ops.chain_state() // 4, dtype=torch.int32, device=dev)
        self.chain_state_t = torch.zeros(ops.chain_state() // 4, dtype=torch.int32, device=dev)
        self.chain_timeline = None
        mode = _xenv("RPO_MLP_FUSED", "0")
        if mode in ("1", "safe") and self.act != torch.float32:
            self._fused_mlp = lambda fc_kw, proj_kw: ops.mlp_fused(fc_kw, proj_kw, self.mlp_counters, safe=(mode == "safe"))

    def _bwd_fold_limits(self) -> tuple:
        # RPO_BWD_FOLD_R3=1: the round-3 coverage (not for d = 1024 / K in (32, 64])
        return (32, (512, 768)) if _xenv("RPO_BWD_FOLD_R3") == "1" else super()._bwd_fold_limits()

    def _split_rows(self, B: int) -> bool:
        """Whether the image forward runs its frozen rows and its prompt rows as separate launches (see _image_forward):
        16-bit modes with the LayerNorm fold, when the one-round row-unit kernels take an image's N frozen rows but not
        its N + K rows (RPO_SPLIT=1), or wherever they take the frozen rows (RPO_SPLIT=force: the test's switch)."""
        cfg = self.cfg
        mode = _xenv("RPO_SPLIT", "0")          # "1": where the whole rows do not fit; "force": wherever the frozen rows do
        if self.act == torch.float32 or not self.fold_ln or mode not in ("1", "force") or cfg.K == 0:
            return False
        key = ("split", B)
        if key not in self._stats_group:
            N, K, dv = cfg.n_frozen, cfg.K, cfg.d_v
            Rf, R = B * N, B * (N + K)
            whole = ops.gemm_hilo_ok(R, dv, dv, self.act, (N, K, Rf), ops.gemm_stats_group(R, dv, dv, self.act, (N, K, Rf)))
            frozen = ops.gemm_hilo_ok(Rf, dv, dv, self.act, (N, 0, Rf), ops.gemm_stats_group(Rf, dv, dv, self.act, (N, 0, Rf)))
            # (round 4: with the 256x96 geometry K = 48 -- 245 rows -- fits whole; "1" now engages from 257 rows on)
            self._stats_group[key] = bool(frozen and (mode == "force" or not whole))
        return self._stats_group[key]

    def chain_ok(self, tower: str, units: int) -> bool:
        """Whether the prompt-row backward chain of a tower ("v" / "t") runs as one persistent launch (rpo_chain_bwd):
        16-bit modes with the QuickGELU derivative saved in the act dtype, widths / key counts / rows per group the
        kernel covers.  OPT-IN (RPO_CHAIN=1; the text tower also needs RPO_CHAIN_TEXT=1): measured SLOWER than the
        launch-per-stage chain -- image tower at B = 32: 1.10-1.20 ms against 0.77 ms (profiles/r04_chain_*.txt).  A
        stage costs ~5-6 us of drain + counter + poll + first dependent load whether or not a kernel boundary sits in
        it, and one workgroup per CU cannot keep enough LDS-DMA bytes in flight (72 KB ring: ~40 GB/s per CU)."""
        if self.act == torch.float32 or _xenv("RPO_CHAIN") != "1" or os.environ.get("RPO_AUX_F32") == "1":
            return False
        cfg = self.cfg
        if tower == "v":
            if _xenv("RPO_CHAIN_IMAGE", "1") == "0":     # (text tower alone: RPO_CHAIN=1 RPO_CHAIN_TEXT=1 RPO_CHAIN_IMAGE=0)
                return False
            return ops.chain_bwd_ok(cfg.layers_v, units, cfg.K, cfg.d_v, cfg.heads_v, cfg.n_frozen, self.act)
        if _xenv("RPO_CHAIN_TEXT", "0") != "1" or self.Lmax > 96:
            return False
        return ops.chain_bwd_ok(cfg.layers_t, units, cfg.K, cfg.d_t, cfg.heads_t, self.Lmax, self.act)

    def _image_chain(self, B: int, b0: int, b1: int, attn_bwd, fold_out: bool) -> torch.Tensor:
        cfg = self.cfg
        N, K, dv, H = cfg.n_frozen, cfg.K, cfg.d_v, cfg.heads_v
        nb, Rf = b1 - b0, B * N
        if not (nb == B and self.chain_ok("v", nb)):   # (never for a part of the batch: one scratch buffer, one resident chain per tower)
            return super()._image_chain(B, b0, b1, attn_bwd, fold_out)
        r0, r1, f0, f1 = b0 * K, b1 * K, b0 * N, b1 * N
        # the 6 x layers stages as ONE persistent launch (rpo_chain_bwd, csrc/chain.hip)
        layers = [dict(w_proj_t=b.w_proj_t, w_fc_t=b.w_fc_t, w_out_t=b.w_out_t, w_q_t=b.w_q_t, aux=self.u[l][r0:r1],
                       x_ln2=self.xm[l][Rf + r0:Rf + r1], x_ln1=self.x[l][Rf + r0:Rf + r1], ln2_w=b.ln2_w, ln1_w=b.ln1_w,
                       q_rows=self.qkv[l][Rf + r0:Rf + r1, :dv], k=self.qkv[l][f0:f1, dv:2 * dv],
                       v=self.qkv[l][f0:f1, 2 * dv:]) for l, b in enumerate(self.vis)]
        dxa, dxb, dxc = self.dxa_v[r0:r1], self.dxb_v[r0:r1], self.dxc_v[r0:r1]
        ops.chain_bwd(layers, units=nb, Kp=K, d=dv, H=H, keys=N, dtype=self.act, ldx=dv, ldq=3 * dv, ldkv=3 * dv,
                      dxa=dxa, dxb=dxb, dxc=dxc, du=self.du_v[r0:r1], dq=self.dq_v[r0:r1], dy=self.dy4_v[:, r0:r1],
                      scale=SCALE, state=self.chain_state_v, timeline=self.chain_timeline)
        return dxa

    def _text_chain(self) -> torch.Tensor:
        cfg = self.cfg
        n, K, dt, H = cfg.n_cls, cfg.K, cfg.d_t, cfg.heads_t
        dxa, dxb, dxc = self.dxa_t, self.dxb_t, self.dxc_t
        if self.chain_ok("t", n):
            # RPO_CHAIN=1 RPO_CHAIN_TEXT=1: the text tower's chain as one persistent launch, classes as units
            layers = [dict(w_proj_t=b.w_proj_t, w_fc_t=b.w_fc_t, w_out_t=b.w_out_t, w_q_t=b.w_q_t, aux=self.ut[l],
                           x_ln2=self.xtm[l], x_ln1=self.xt[l], ln2_w=b.ln2_w, ln1_w=b.ln1_w, q_rows=self.qt[l],
                           k=self.kv_t[l][:, :dt], v=self.kv_t[l][:, dt:]) for l, b in enumerate(self.txt)]
            ops.chain_bwd(layers, units=n, Kp=K, d=dt, H=H, keys=self.Lmax, dtype=self.act, key_len=self.len_i32,
                          key_stride=self.Lmax, ldx=dt, ldq=dt, ldkv=2 * dt, dxa=dxa, dxb=dxb, dxc=dxc, du=self.du_t,
                          dq=self.dq_t, dy=self.dy4_t, scale=SCALE, state=self.chain_state_t)
            return dxa
        # RPO_TEXT_BWD_FOLD=1 (16-bit modes, K <= 32, <= 96 keys): the MFMA attention backward of the image tower with
        # per-class key counts and the d out-proj GEMM folded in (rpo_attn_bwd_proj_pair) instead of a GEMM + the VALU
        # kernel.  Measured and NOT the default: the text chain alone gets 16 % shorter (583 -> 490 us) and the step 0.4 %
        # LONGER (3.016 vs 3.003 ms at B = 32, three alternating pairs; 0 at B = 4) -- the text chain is not what the step
        # waits for, its kernels' footprint on the CUs is, and the MFMA kernel (250 VGPRs, 66 KB of LDS per workgroup)
        # stands in the image chain's way more than the VALU kernel + a 64x64 GEMM do.
        fold_out = (self.act != torch.float32 and K <= 32 and dt in (512, 768) and self.Lmax <= 96
                    and os.environ.get("RPO_NO_BWD_FOLD") != "1" and _xenv("RPO_TEXT_BWD_FOLD") == "1")
        if not fold_out:
            return super()._text_chain()

        def attn_bwd(l, da, dq):
            kv = self.kv_t[l]
            ops.attn_bwd_proj_pair(dict(q_rows=self.qt[l], k=kv[:, :dt], v=kv[:, dt:], dx=da, w_out_t=self.txt[l].w_out_t,
                                        dq=dq, groups=n, H=H, keys=self.Lmax, Kp=K, scale=SCALE, key_len=self.len_i32,
                                        key_stride=self.Lmax))

        return self._rows_backward(self.txt, self.xt[:-1], self.xtm, self.ut, dxa, dxb, dxc, self.du_t, self.da_t, self.dq_t,
                                   self.dy_t, attn_bwd, fold_out=True, tt=self._text_tiles)

    # ------------------------------------------------------------------ both backward chains as ONE chain of launches
    def joint_backward_ok(self, B: Optional[int] = None) -> bool:
        """The two prompt-row chains can be issued pairwise (one launch per stage for both towers) when both attention
        backwards run with the d out-proj GEMM folded in: 16-bit modes, K <= 32, widths 512 / 768, <= 96 text keys.
        Measured and NOT the default: the 64x64-tile GEMMs of the chains are bound by the CUs' LDS-DMA rate, not by
        latency (a 64-deep k-tile of a 64x64 tile is 16 KB at the ~34 B/clk a CU's DMA path delivers = the ~480 cycles
        per k-tile of the timeline), so a paired launch takes the SUM of its two problems' times (d c_proj pair 23.3 us
        against 15 + 9 alone, attention pair 21.9 against 17.3 + 8.9), and what is left to gain is the second queue.
        Same box, backward phase alone at B = 4 / 8 / 16 / 32: 0.70 / 0.71 / 0.78 / 1.02 ms paired against 0.73 / 0.73 /
        0.78 / 0.91 ms as two chains on two streams; whole step 1.70 / 2.04 / - / 3.20 ms against 1.65 / 1.97 / - / 3.02.
        RPO_JOINT_BWD=1 turns it on (results are bit-identical either way: tests/test_gpu_model.py)."""
        cfg = self.cfg
        can = (self.act != torch.float32 and cfg.K <= 32 and cfg.d_v == 768 and cfg.d_t in (512, 768)
               and self.Lmax <= 96 and cfg.n_frozen > 96 and cfg.n_frozen <= 224
               and os.environ.get("RPO_NO_BWD_FOLD") != "1")
        return can and _xenv("RPO_JOINT_BWD") == "1"

    def _joint_backward(self, B: int) -> None:
        """_image_backward + _text_backward with every stage of the two chains in ONE launch (rpo_gemm_nt_pair,
        rpo_layernorm_bwd_pair, rpo_attn_bwd_proj_pair): the chains have the same six stages per block, and as two
        chains on two queues their ~150 small kernels delayed each other (backward pair 0.93 ms against 0.78 ms for the
        image chain alone, profiles/README.md).  The arithmetic of every problem is that of the separate launches."""
        cfg = self.cfg
        N, K, dv, dt = cfg.n_frozen, cfg.K, cfg.d_v, cfg.d_t
        Rf, Rp, Rt, n = B * N, B * K, self.Rt, cfg.n_cls
        R = Rf + Rp
        pf = self._pf_chains
        V = dict(blocks=self.vis, x=[t[Rf:R] for t in self.x], xm=[t[Rf:R] for t in self.xm], u=[t[:Rp] for t in self.u],
                 dxa=self.dxa_v[:Rp], dxb=self.dxb_v[:Rp], dxc=self.dxc_v[:Rp], du=self.du_v[:Rp], dq=self.dq_v[:Rp],
                 dy=self.dy_v[:, :Rp])
        T = dict(blocks=self.txt, x=self.xt, xm=self.xtm, u=self.ut, dxa=self.dxa_t, dxb=self.dxb_t, dxc=self.dxc_t,
                 du=self.du_t, dq=self.dq_t, dy=self.dy_t)

        def attn_args(c, l):
            if c is V:
                qkv = self.qkv[l]
                return dict(q_rows=qkv[Rf:R, :dv], k=qkv[:Rf, dv:2 * dv], v=qkv[:Rf, 2 * dv:], dx=c["dxc"],
                            w_out_t=self.vis[l].w_out_t, dq=c["dq"], groups=B, H=cfg.heads_v, keys=N, Kp=K, scale=SCALE)
            kv = self.kv_t[l]
            return dict(q_rows=self.qt[l], k=kv[:, :dt], v=kv[:, dt:], dx=c["dxc"], w_out_t=self.txt[l].w_out_t,
                        dq=c["dq"], groups=n, H=cfg.heads_t, keys=self.Lmax, Kp=K, scale=SCALE, key_len=self.len_i32,
                        key_stride=self.Lmax)

        def gemm(calls):
            if len(calls) == 2:
                ops.gemm_nt_pair(*calls)
            else:
                ops.gemm_nt(**calls[0])

        def ln(calls):
            if len(calls) == 2:
                ops.layernorm_bwd_pair(*calls)
            else:
                ops.layernorm_bwd(**calls[0])

        # heads of the chains: d projection, then ln_post / ln_final (rpo.py:210 / :183)
        gemm([dict(a=self.d_img_f_a[:Rp], w=self.img_proj, out=self.dy_v[0, :Rp], epilogue=EPI_NONE,
                   prefetch=self.vis[-1].w_proj_t if pf else None),
              dict(a=self.d_text_f_a, w=self.text_proj, out=self.dy_t[0], epilogue=EPI_NONE,
                   prefetch=self.txt[-1].w_proj_t if pf else None)])
        ln([dict(dy=self.dy_v[0, :Rp], x=V["x"][-1], gamma=self.ln_post[0], dres=None, dx=V["dxa"], dx_cast=V["dxc"]),
            dict(dy=self.dy_t[0], x=self.xt[-1], gamma=self.ln_final[0], dres=None, dx=T["dxa"], dx_cast=T["dxc"])])
        Lv, Lt = len(self.vis), len(self.txt)
        for s_ in range(max(Lv, Lt)):
            live = [(c, len(c["blocks"]) - 1 - s_) for c in (V, T) if len(c["blocks"]) - 1 - s_ >= 0]
            blk = lambda c, l: c["blocks"][l]
            # the six stages of _rows_backward, for every tower that still has a block at this depth
            gemm([dict(a=c["dxc"], w=blk(c, l).w_proj_t, out=c["du"], epilogue=EPI_QGELU_BWD, aux=c["u"][l],
                       prefetch=blk(c, l).w_fc_t if pf else None) for c, l in live])                       # d c_proj, d QuickGELU
            gemm([dict(a=c["du"], w=blk(c, l).w_fc_t, out=c["dy"][:SPLIT_FC], epilogue=EPI_NONE, split_k=SPLIT_FC,
                       prefetch=blk(c, l).w_oq_t if pf else None) for c, l in live])                       # d c_fc
            ln([dict(dy=c["dy"][:SPLIT_FC], x=c["xm"][l], gamma=blk(c, l).ln2_w, dres=c["dxa"], dx=c["dxb"],
                     dx_cast=c["dxc"]) for c, l in live])
            ops.attn_bwd_proj_pair(*[attn_args(c, l) for c, l in live])                                    # d out-proj + attention
            gemm([dict(a=c["dq"], w=blk(c, l).w_q_t, out=c["dy"][:SPLIT_Q], epilogue=EPI_NONE, split_k=SPLIT_Q,
                       prefetch=c["blocks"][l - 1].w_proj_t if (pf and l > 0) else None) for c, l in live])  # d q-projection
            ln([dict(dy=c["dy"][:SPLIT_Q], x=c["x"][l], gamma=blk(c, l).ln1_w, dres=c["dxb"], dx=c["dxa"],
                     dx_cast=c["dxc"]) for c, l in live])
        # image: through ln_pre (rpo.py:206) to the appended prompt rows; both: sum over the batch / the classes (.repeat)
        ops.layernorm_bwd(V["dxa"], self.x_pre[Rf:R], self.ln_pre[0], None, V["dxb"])
        ops.reduce_groups(V["dxb"], self.g_img, B)
        ops.reduce_groups(T["dxa"], self.g_text, n)
