"""The sibling trainers' share of the engine: CoOp (trainers/coop.py) and CoCoOp
(trainers/cocoop.py) train context vectors that EVERY later token reads, so they need the dense text-tower forward +
backward over all tokens; the image tower is plain frozen CLIP.  Mixed into rpo_amd.engine.Engine, whose packed weights,
K / V cache, head and streams these methods use."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import ops
from ._lib import EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_NONE, EPI_QGELU_BWD

SCALE = 0.125                       # 1 / sqrt(head_dim = 64)
SPLIT_FC, SPLIT_Q = 3, 2            # split-K factors of the two fp32-output dX GEMMs on rpo_gemm_nt (engine.py)


class CoopEngineMixin:
    # ------------------------------------------------------------------ CoOp / CoCoOp: training context vectors
    def coop_layout(self, n_ctx: int, class_token_position: str = "end"):
        """Where CoOp's PromptLearner.forward (trainers/coop.py:117-183) puts things: for every class c and sequence
        position p, `src[c, p]` = the position of the "X X .. name." prompt whose TOKEN embedding sits there (-1 where a
        context vector sits) and `ctx_pos[c, j]` = the position of context vector j.  "end": [SOS | ctx | name . EOT];
        "middle": [SOS | ctx[:n/2] | name | ctx[n/2:] | . EOT]; "front": [SOS | name | ctx | . EOT].  name_len of a class
        = its prompt length - n_ctx - 3 (SOS, '.', EOT), i.e. `len(_tokenizer.encode(name))` (:99)."""
        assert class_token_position in ("end", "middle", "front"), class_token_position    # `else: raise ValueError`, :185
        n, L = self.cfg.n_cls, self.Lmax
        src = np.tile(np.arange(L, dtype=np.int64), (n, 1))
        ctx_pos = np.zeros((n, n_ctx), dtype=np.int64)
        half = n_ctx // 2
        for c in range(n):
            nl = int(self.len_np[c]) - n_ctx - 3
            assert nl >= 1, "tokens must be the ids of the 'X X .. name.' prompts with n_ctx placeholders"
            name = np.arange(1 + n_ctx, 1 + n_ctx + nl)
            if class_token_position == "end":
                cp = np.arange(1, 1 + n_ctx)
            elif class_token_position == "middle":
                cp = np.concatenate([np.arange(1, 1 + half), np.arange(1 + half + nl, 1 + n_ctx + nl)])
                src[c, 1 + half:1 + half + nl] = name
            else:
                cp = np.arange(1 + nl, 1 + nl + n_ctx)
                src[c, 1:1 + nl] = name
            src[c, cp] = -1
            ctx_pos[c] = cp
        return src, ctx_pos

    def coop_setup(self, n_ctx: int, replicas: int = 1, meta_hidden: int = 0, csc: bool = False,
                   class_token_position: str = "end") -> None:
        """Buffers of the sibling trainers CoOp (trainers/coop.py) and CoCoOp (trainers/cocoop.py): the learned context
        `coop_ctx` [n_ctx, d_t] and, per text block, everything the DENSE text-tower backward re-reads -- the gradient of a
        context vector flows through every token of every class (plain causal mask), unlike RPO's prompts.
        `replicas` > 1 (CoCoOp): the class set is run once per IMAGE with that image's shifted context, i.e. as
        replicas * n_cls virtual classes; `meta_hidden` > 0 adds the meta-net (linear1 [h, e], linear2 [d_t, h]).  All
        trained tensors live in one flat fp32 buffer (`coop_params` = [ctx | w1 | b1 | w2 | b2]) with matching gradient
        and momentum buffers: one SGD launch.  n_cls * Lmax rows per replica (a few hundred), so this is small."""
        cfg, dev, act = self.cfg, self.dev, self.act
        n, L, dt, e = cfg.n_cls, self.Lmax, cfg.d_t, cfg.embed
        assert 1 + n_ctx < self.Lmax and self.Lmax <= 80, "tokens must be the ids of the 'X X .. name.' prompts"
        assert 1 <= replicas <= self.max_batch
        nv = replicas * n
        Rf = nv * L
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        a = lambda *s: torch.empty(*s, dtype=act, device=dev)
        au = f32 if act == torch.float32 else a
        Lt, h = cfg.layers_t, meta_hidden
        self.coop_n_ctx, self.coop_replicas, self.coop_hidden = n_ctx, replicas, h
        # class-specific contexts (TRAINER.COOP.CSC, trainers/coop.py:84-86): ctx [n_cls, n_ctx, d_t] instead of one
        # [n_ctx, d_t] expanded over the classes (:119-121) -- no sum over the classes in the backward
        assert not (csc and (replicas > 1 or h)), "class-specific contexts are CoOp's (CoCoOp's context is generic)"
        self.coop_csc, self.coop_position = bool(csc), class_token_position
        nctx_rows = (n if csc else 1) * n_ctx
        src, ctx_pos = self.coop_layout(n_ctx, class_token_position)
        rows = np.arange(n)[:, None] * L
        # gather map of the token embeddings (context slots read row 0 and are overwritten) and the rows / positions of
        # the context vectors, class-major
        self.c_src_rows = torch.as_tensor((rows + np.maximum(src, 0)).reshape(-1), device=dev)
        self.c_ctx_rows_idx = torch.as_tensor((rows + ctx_pos).reshape(-1), device=dev)
        self.c_ctx_pos = torch.as_tensor(ctx_pos.reshape(-1), device=dev)
        sizes = [nctx_rows * dt] + ([h * e, h, dt * h, dt] if h else [])
        tot = sum(sizes)
        self.coop_params = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.coop_grads = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.coop_moms = torch.zeros(tot, dtype=torch.float32, device=dev)
        offs = np.cumsum([0] + sizes)
        view = lambda buf, i, *shape: buf[offs[i]:offs[i + 1]].view(*shape)
        cshape = (n, n_ctx, dt) if csc else (n_ctx, dt)
        self.coop_ctx, self.coop_grad = view(self.coop_params, 0, *cshape), view(self.coop_grads, 0, *cshape)
        if h:
            shapes = [(h, e), (h,), (dt, h), (dt,)]
            self.meta = [view(self.coop_params, i + 1, *sh) for i, sh in enumerate(shapes)]          # w1, b1, w2, b2
            self.meta_grad = [view(self.coop_grads, i + 1, *sh) for i, sh in enumerate(shapes)]
            self.c_fn, self.c_hid, self.c_bias, self.c_dbias = f32(replicas, e), f32(replicas, h), f32(replicas, dt), f32(replicas, dt)
        self.c_shift = f32(replicas, nctx_rows, dt)                   # the context each replica's classes carry
        self.c_dshift = f32(replicas, nctx_rows, dt)
        self.c_len = self.len_i32.repeat(replicas).contiguous()
        self.cx = [f32(Rf, dt) for _ in range(Lt + 1)]
        self.cxm = [f32(Rf, dt) for _ in range(Lt)]
        self.cqkv = [a(Rf, 3 * dt) for _ in range(Lt)]
        self.cu = [au(Rf, 4 * dt) for _ in range(Lt)]
        self.ch, self.catt, self.cg = a(Rf, dt), a(Rf, dt), a(Rf, 4 * dt)
        self.c_dxa, self.c_dxb = f32(Rf, dt), f32(Rf, dt)
        self.c_dxc, self.c_da = a(Rf, dt), a(Rf, dt)
        self.c_du, self.c_dqkv = a(Rf, 4 * dt), a(Rf, 3 * dt)
        self.c_dy = f32(max(SPLIT_FC, SPLIT_Q), Rf, dt)
        self.c_eot = torch.arange(nv, device=dev) * L + (self.c_len.to(torch.int64) - 1)          # EOT row of every class
        self.c_x_eot, self.c_dx_eot, self.c_dy_eot = f32(nv, dt), f32(nv, dt), f32(nv, dt)
        self.c_y_eot = a(nv, dt)
        self.c_text_f, self.c_d_text_f = f32(nv, e), f32(nv, e)
        self.c_d_text_f_a = a(nv, e)
        self.c_d_img_f = f32(self.max_batch, e)
        self.c_loss_b = f32(self.max_batch)
        self.c_ctx_rows = f32(nv * n_ctx, dt)
        if self.img_cls_f is None:
            self.img_cls_f = torch.empty(self.max_batch, e, dtype=torch.float32, device=dev)
        # dX of the packed in-projection needs the whole W_in transposed ([d, 3d]); RPO's backward only its q third
        self.c_w_in_t = [blk.w_in.t().contiguous() for blk in self.txt]

    def _coop_text_forward(self, train: bool, R: int = 1) -> None:
        """TextEncoder.forward of trainers/coop.py:47-58 on prompts = [SOS | ctx | class name . EOT] (:117-134): all
        tokens up to the longest EOT, plain causal mask, every block's inputs kept for the backward.  R replicas of the
        class set, replica r carrying the context c_shift[r] (CoOp: one replica, the context itself)."""
        cfg = self.cfg
        n, L, dt, H, nc = cfg.n_cls, self.Lmax, cfg.d_t, cfg.heads_t, self.coop_n_ctx
        nv = R * n
        Rf = nv * L
        x0 = self.cx[0][:Rf]
        if self.coop_position == "end" and not self.coop_csc:
            x0.view(R, n * L, dt).copy_(self.text_x_frozen.view(1, n * L, dt).expand(R, -1, -1))
            x0.view(R, n, L, dt)[:, :, 1:1 + nc] = (self.c_shift[:R] + self.text_pos[1:1 + nc]).unsqueeze(1)
        else:
            # prompts = cat([prefix, ctx / class name in the configured order, suffix]) (trainers/coop.py:117-183), then
            # + positional_embedding by position (TextEncoder.forward, :48)
            tokpos = torch.index_select(self.text_tok, 0, self.c_src_rows).view(n, L, dt) + self.text_pos
            x0.view(R, n * L, dt).copy_(tokpos.view(1, n * L, dt).expand(R, -1, -1))
            cpos = self.text_pos.index_select(0, self.c_ctx_pos)                                  # [n * nc, dt]
            cs = self.c_shift[:R]
            vals = (cs if self.coop_csc else cs.unsqueeze(1).expand(R, n, nc, dt).reshape(R, n * nc, dt)) + cpos
            for r in range(R):
                x0[r * n * L:(r + 1) * n * L].index_copy_(0, self.c_ctx_rows_idx, vals[r])
        ch, catt, cg, ln = self.ch[:Rf], self.catt[:Rf], self.cg[:Rf], self.c_len[:nv]
        for l, blk in enumerate(self.txt):
            x, xm, qkv = self.cx[l][:Rf], self.cxm[l][:Rf], self.cqkv[l][:Rf]
            ops.layernorm_fwd(x, blk.ln1_w, blk.ln1_b, ch)
            ops.gemm_nt(ch, blk.w_in, qkv, EPI_BIAS, bias=blk.b_in)
            ops.text_attn_fwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], catt, ln, nv, L, L, H, causal=True, scale=SCALE)
            ops.gemm_nt(catt, blk.w_out, xm, EPI_BIAS_RESID, bias=blk.b_out, resid=x)
            ops.layernorm_fwd(xm, blk.ln2_w, blk.ln2_b, ch)
            ops.gemm_nt(ch, blk.w_fc, cg, EPI_BIAS_QGELU, bias=blk.b_fc, aux=self.cu[l][:Rf] if train else None,
                        aux_row0=0 if train else Rf)
            ops.gemm_nt(cg, blk.w_proj, self.cx[l + 1][:Rf], EPI_BIAS_RESID, bias=blk.b_proj, resid=xm)
        torch.index_select(self.cx[-1], 0, self.c_eot[:nv], out=self.c_x_eot[:nv])        # feature at the EOT token
        ops.layernorm_fwd(self.c_x_eot[:nv], self.ln_final[0], self.ln_final[1], self.c_y_eot[:nv])
        ops.gemm_nt(self.c_y_eot[:nv], self.text_proj_t, self.c_text_f[:nv], EPI_NONE)

    def _coop_text_backward(self, R: int = 1) -> None:
        """c_dshift[r] = d loss / d (context of replica r): autograd of the whole text tower for all tokens (dX GEMMs only
        -- the weights are frozen), the causal attention backward with dK / dV (rpo_text_attn_bwd_dense), then the rows of
        the context positions summed over the classes (ctx.unsqueeze(0).expand, trainers/coop.py:119-121)."""
        cfg = self.cfg
        n, L, dt, H, nc = cfg.n_cls, self.Lmax, cfg.d_t, cfg.heads_t, self.coop_n_ctx
        nv = R * n
        Rf = nv * L
        f32m = self.act == torch.float32
        dxa, dxb, dxc, dy = self.c_dxa[:Rf], self.c_dxb[:Rf], self.c_dxc[:Rf], self.c_dy[:, :Rf]
        du, da, dq, ln = self.c_du[:Rf], self.c_da[:Rf], self.c_dqkv[:Rf], self.c_len[:nv]
        ops.gemm_nt(self.c_d_text_f[:nv] if f32m else self.c_d_text_f_a[:nv], self.text_proj, self.c_dy_eot[:nv], EPI_NONE)
        ops.layernorm_bwd(self.c_dy_eot[:nv], self.c_x_eot[:nv], self.ln_final[0], None, self.c_dx_eot[:nv])
        dxa.zero_()
        dxa.index_copy_(0, self.c_eot[:nv], self.c_dx_eot[:nv])
        if not f32m:
            ops.convert(dxa, dxc)
        for l in reversed(range(len(self.txt))):
            blk, qkv = self.txt[l], self.cqkv[l][:Rf]
            ops.gemm_nt(dxa if f32m else dxc, blk.w_proj_t, du, EPI_QGELU_BWD, aux=self.cu[l][:Rf])
            ops.gemm_nt(du, blk.w_fc_t, dy[:SPLIT_FC], EPI_NONE, split_k=SPLIT_FC)
            ops.layernorm_bwd(dy[:SPLIT_FC], self.cxm[l][:Rf], blk.ln2_w, dxa, dxb, None if f32m else dxc)
            ops.gemm_nt(dxb if f32m else dxc, blk.w_out_t, da, EPI_NONE)
            ops.text_attn_bwd_dense(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], da, dq[:, :dt], dq[:, dt:2 * dt],
                                    dq[:, 2 * dt:], ln, nv, L, H, SCALE)
            ops.gemm_nt(dq, self.c_w_in_t[l], dy[:SPLIT_Q], EPI_NONE, split_k=SPLIT_Q)
            ops.layernorm_bwd(dy[:SPLIT_Q], self.cx[l][:Rf], blk.ln1_w, dxb, dxa, None if f32m else dxc)
        rows = self.c_ctx_rows[:nv * nc]
        if self.coop_position == "end" and not self.coop_csc:
            rows.view(nv, nc, dt).copy_(dxa.view(nv, L, dt)[:, 1:1 + nc])
        else:
            for r in range(R):          # the rows the context vectors sat in, class-major
                torch.index_select(dxa[r * n * L:(r + 1) * n * L], 0, self.c_ctx_rows_idx, out=rows[r * n * nc:(r + 1) * n * nc])
        if self.coop_csc:               # every class has its own vectors: nothing to sum
            self.c_dshift[0].copy_(rows[:n * nc])
            return
        for r in range(R):              # reduce_groups sums `groups` consecutive blocks of `rows` rows: the classes
            ops.reduce_groups(rows[r * n * nc:(r + 1) * n * nc], self.c_dshift[r], n)

    def _plain_image_features(self, image: torch.Tensor) -> int:
        cfg = self.cfg
        B = image.shape[0]
        assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous() and B <= self.max_batch
        assert image.device == self.dev and torch.cuda.current_device() == self.dev.index
        N, dv = cfg.n_frozen, cfg.d_v
        self._image_forward(image, train=False, full_last=True)
        cls_rows = self.x[-1][:B * N].view(B, N, dv)[:, 0, :]
        ops.layernorm_fwd(cls_rows, self.ln_post[0], self.ln_post[1], self.y_post[:B])
        ops.gemm_nt(self.y_post[:B], self.img_proj_t, self.img_cls_f[:B], EPI_NONE)
        return B

    def coop_forward_backward(self, image: torch.Tensor, label: Optional[torch.Tensor]) -> torch.Tensor:
        """trainers/coop.py:196-208 + :266-270: logits = exp(logit_scale) * normalise(image features of the plain image
        tower) @ normalise(text features of [SOS | ctx | name . EOT])^T; with `label`, also the mean cross-entropy
        (self.loss) and d loss / d ctx (self.coop_grad).  Returns self.logits[:B]."""
        cfg = self.cfg
        e, n = cfg.embed, cfg.n_cls
        train = label is not None
        self.c_shift[0].copy_(self.coop_ctx.view(-1, cfg.d_t))
        # the two towers are independent until the head: the (small, latency-bound) dense text forward runs on the side
        # stream under the image tower, as RPO's text chain does (RPO_COOP_SERIAL=1: one stream)
        if os.environ.get("RPO_COOP_SERIAL") == "1":
            self._coop_text_forward(train)
            B = self._plain_image_features(image)
        else:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._coop_text_forward(train)
            B = self._plain_image_features(image)
            main.wait_stream(self.side)
        extra = {} if (not train or self.act == torch.float32) else dict(d_text_f_act=self.c_d_text_f_a[:n])
        ops.head_fwd_bwd(self.img_cls_f[:B].view(B, 1, e), self.c_text_f[:n].view(n, 1, e), label, self.logit_scale_exp,
                         self.logits[:B], self.loss if train else None,
                         self.c_d_img_f[:B].view(B, 1, e) if train else None,
                         self.c_d_text_f[:n].view(n, 1, e) if train else None, self.head_ws, **extra)
        if train:
            self._coop_text_backward()
            self.coop_grad.view(-1, cfg.d_t).copy_(self.c_dshift[0])
        return self.logits[:B]

    def cocoop_forward_backward(self, image: torch.Tensor, label: Optional[torch.Tensor]) -> torch.Tensor:
        """trainers/cocoop.py:166-192: image features -> meta-net -> one shifted context per image -> that image's own
        text features for every class -> logits[b] = exp(logit_scale) * imf_n[b] @ normalise(text_f[b])^T; with `label`
        the mean cross-entropy (self.loss) and the gradients of ctx and of the four meta-net tensors (self.coop_grads).
        The batch size is bounded by coop_setup's `replicas` (the reference trains CoCoOp at batch 1:
        configs/trainers/CoCoOp/vit_b16_c4_ep10_batch1.yaml)."""
        cfg = self.cfg
        e, n, nc = cfg.embed, cfg.n_cls, self.coop_n_ctx
        train = label is not None
        B = self._plain_image_features(image)
        assert B <= self.coop_replicas and self.coop_hidden > 0
        w1, b1, w2, b2 = self.meta
        ops.metanet_fwd(self.img_cls_f[:B], w1, b1, w2, b2, self.c_fn[:B], self.c_hid[:B], self.c_bias[:B])
        torch.add(self.coop_ctx.unsqueeze(0), self.c_bias[:B].unsqueeze(1), out=self.c_shift[:B])   # ctx + bias (:141-143)
        self._coop_text_forward(train, B)
        f32m = self.act == torch.float32
        for b in range(B):                  # every image has its own text features: the head runs per image (:183-188)
            tf = slice(b * n, (b + 1) * n)
            extra = {} if (not train or f32m) else dict(d_text_f_act=self.c_d_text_f_a[tf])
            ops.head_fwd_bwd(self.img_cls_f[b:b + 1].view(1, 1, e), self.c_text_f[tf].view(n, 1, e),
                             label[b:b + 1] if train else None, self.logit_scale_exp, self.logits[b:b + 1],
                             self.c_loss_b[b:b + 1] if train else None,
                             self.c_d_img_f[b:b + 1].view(1, 1, e) if train else None,
                             self.c_d_text_f[tf].view(n, 1, e) if train else None, self.head_ws, **extra)
        if train:
            torch.mean(self.c_loss_b[:B], dim=0, keepdim=True, out=self.loss)         # F.cross_entropy: mean over the batch
            self._coop_text_backward(B)
            ds = self.c_dshift[:B]
            ds.mul_(1.0 / B)                                                          # ... and so are its gradients
            torch.sum(ds, dim=0, out=self.coop_grad)                                  # ctx is shared by all images
            torch.sum(ds, dim=1, out=self.c_dbias[:B])                                # bias[b] is added to every context row
            g1, gb1, g2, gb2 = self.meta_grad
            ops.metanet_bwd(self.c_dbias[:B], self.c_fn[:B], self.c_hid[:B], w2, g1, gb1, g2, gb2)
        return self.logits[:B]

