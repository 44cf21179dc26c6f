"""ORACLE (test infrastructure, not product): the *minimal* RPO step on the CPU.

Same results as ``oracle/rpo_oracle.py`` (the dense restatement of
trainers/rpo.py:161-232) but organised the way the HIP path is: the frozen
tokens of both towers run inference-only, the K prompt rows read the frozen
tokens' keys/values and are the only rows that are back-propagated
(SURVEY.md finding 4: the visual mask blocks prompt *columns* for every row,
trainers/rpo.py:154-156, and the text mask is causal AND ``col < len_c``,
:144-151, so no token ever reads a prompt).  Backward is written out by hand --
no autograd -- so every function here is the op-level oracle of one HIP kernel:

  ln_fwd / ln_bwd            <-> rpo_layernorm_fwd / rpo_layernorm_bwd
  qgelu / qgelu_grad         <-> GEMM epilogues RPO_EPI_BIAS_QGELU / RPO_EPI_QGELU_BWD
  attn_rows_fwd / _bwd       <-> rpo_attn_readonly_fwd / rpo_attn_readonly_bwd
                                 rpo_text_attn_fwd / rpo_text_attn_bwd
  head_fwd_bwd               <-> rpo_head_fwd_bwd
  sgd                        <-> rpo_sgd_step

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  tests/test_oracle_rows.py checks it against the dense oracle's
autograd (bit-for-bit structure claim of SURVEY.md appendix A.5, to fp32
round-off).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .rpo_oracle import HEAD_DIM, LN_EPS, OracleRPO, _t

QG = 1.702


# ---------------------------------------------------------------------------
# op-level oracles
# ---------------------------------------------------------------------------

def ln_fwd(x, w, b, eps: float = LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def ln_bwd(dy, x, w, eps: float = LN_EPS):
    """dL/dx of y = LN(x)*w + b given dL/dy (weights frozen: no dw/db)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    xh = (x - mu) * rstd
    g = dy * w
    return rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))


def qgelu(u):
    return u * torch.sigmoid(QG * u)


def qgelu_grad(u):
    s = torch.sigmoid(QG * u)
    return s * (1.0 + QG * u * (1.0 - s))


def _heads(x, H):            # [R, H*64] -> [H, R, 64]
    R = x.shape[0]
    return x.reshape(R, H, HEAD_DIM).transpose(0, 1)


def _unheads(x):             # [H, R, 64] -> [R, H*64]
    H, R, _ = x.shape
    return x.transpose(0, 1).reshape(R, H * HEAD_DIM)


def attn_rows_fwd(q, k, v, H: int):
    """q[R, D] rows attend to keys k[N, D] / values v[N, D] of ONE sequence, all N
    keys visible; returns [R, D].  softmax(q k^T / 8) v per head."""
    qh, kh, vh = _heads(q, H), _heads(k, H), _heads(v, H)
    p = torch.softmax(qh @ kh.transpose(1, 2) * (1.0 / math.sqrt(HEAD_DIM)), dim=-1)
    return _unheads(p @ vh)


def attn_rows_bwd(q, k, v, da, H: int):
    """dL/dq for attn_rows_fwd (k, v come from frozen tokens: no dk/dv needed)."""
    sc = 1.0 / math.sqrt(HEAD_DIM)
    qh, kh, vh, dah = _heads(q, H), _heads(k, H), _heads(v, H), _heads(da, H)
    p = torch.softmax(qh @ kh.transpose(1, 2) * sc, dim=-1)
    dp = dah @ vh.transpose(1, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdim=True))
    return _unheads(ds @ kh) * sc


def attn_causal_fwd(q, k, v, H: int, klimit):
    """Row r sees keys [0, klimit[r]) -- the one-off frozen-row text pass
    (causal AND col < len_c, trainers/rpo.py:146-149)."""
    qh, kh, vh = _heads(q, H), _heads(k, H), _heads(v, H)
    s = qh @ kh.transpose(1, 2) * (1.0 / math.sqrt(HEAD_DIM))
    cols = torch.arange(k.shape[0])[None, :]
    s = s.masked_fill(cols >= _t(klimit).long()[:, None], float("-inf"))
    return _unheads(torch.softmax(s, dim=-1) @ vh)


def head_fwd_bwd(img_f, text_f, label, logit_scale_exp: float):
    """trainers/rpo.py:215-230 plus its backward.
    img_f[B,K,e], text_f[C,K,e] -> logits[B,C], loss, d_img_f, d_text_f."""
    B, K, _ = img_f.shape
    ni = img_f.norm(dim=-1, keepdim=True)
    nt = text_f.norm(dim=-1, keepdim=True)
    ih, th = img_f / ni, text_f / nt
    logits = torch.einsum("bke,cke->bc", ih, th) * (logit_scale_exp / K)
    if label is None:
        return logits, None, None, None
    label = _t(label).long()
    lse = torch.logsumexp(logits, dim=-1)
    loss = (lse - logits[torch.arange(B), label]).mean()
    dl = torch.softmax(logits, dim=-1)
    dl[torch.arange(B), label] -= 1.0
    dl = dl * (logit_scale_exp / (K * B))
    dih = torch.einsum("bc,cke->bke", dl, th)
    dth = torch.einsum("bc,bke->cke", dl, ih)
    d_img = (dih - ih * (ih * dih).sum(-1, keepdim=True)) / ni
    d_text = (dth - th * (th * dth).sum(-1, keepdim=True)) / nt
    return logits, loss, d_img, d_text


def sgd(p, g, buf, lr, momentum, wd, first: bool, grad_scale: float = 1.0):
    """torch.optim.SGD (dampening 0, no nesterov) on one tensor; returns (p, buf)."""
    g = g * grad_scale + wd * p
    buf = g.clone() if first else momentum * buf + g
    return p - lr * buf, buf


# ---------------------------------------------------------------------------
# the minimal step
# ---------------------------------------------------------------------------

class RowsRPO:
    """Minimal-work RPO built on an ``OracleRPO``'s weights."""

    def __init__(self, dense: OracleRPO):
        self.m = dense
        self.K = dense.K
        self.text_kv: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None

    # -- one residual block on a set of rows that read external keys/values ----
    @staticmethod
    def _block_rows_fwd(x, kf, vf, blk, H, save):
        """x[G, R, D] prompt rows of G sequences; kf/vf: list of G [N_g, D]."""
        D = x.shape[-1]
        wq, bq = blk["attn.in_proj_weight"][:D], blk["attn.in_proj_bias"][:D]
        h1 = ln_fwd(x, blk["ln_1.weight"], blk["ln_1.bias"])
        q = h1 @ wq.t() + bq
        a = torch.stack([attn_rows_fwd(q[g], kf[g], vf[g], H) for g in range(x.shape[0])])
        x1 = x + a @ blk["attn.out_proj.weight"].t() + blk["attn.out_proj.bias"]
        h2 = ln_fwd(x1, blk["ln_2.weight"], blk["ln_2.bias"])
        u = h2 @ blk["mlp.c_fc.weight"].t() + blk["mlp.c_fc.bias"]
        x2 = x1 + qgelu(u) @ blk["mlp.c_proj.weight"].t() + blk["mlp.c_proj.bias"]
        save.append(dict(x=x, q=q, x1=x1, u=u))
        return x2

    @staticmethod
    def _block_rows_bwd(dx2, sv, kf, vf, blk, H):
        D = dx2.shape[-1]
        wq = blk["attn.in_proj_weight"][:D]
        dg = dx2 @ blk["mlp.c_proj.weight"]
        du = dg * qgelu_grad(sv["u"])
        dh2 = du @ blk["mlp.c_fc.weight"]
        dx1 = dx2 + ln_bwd(dh2, sv["x1"], blk["ln_2.weight"])
        da = dx1 @ blk["attn.out_proj.weight"]
        dq = torch.stack([attn_rows_bwd(sv["q"][g], kf[g], vf[g], da[g], H) for g in range(dx2.shape[0])])
        dh1 = dq @ wq
        return dx1 + ln_bwd(dh1, sv["x"], blk["ln_1.weight"])

    # -- frozen rows: plain inference, returning per-layer K/V ------------------
    @staticmethod
    def _frozen_block(x, blk, H, klimit=None):
        """x[N, D] one sequence of frozen tokens; returns (x_out, k, v)."""
        D = x.shape[-1]
        h1 = ln_fwd(x, blk["ln_1.weight"], blk["ln_1.bias"])
        qkv = h1 @ blk["attn.in_proj_weight"].t() + blk["attn.in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        a = attn_rows_fwd(q, k, v, H) if klimit is None else attn_causal_fwd(q, k, v, H, klimit)
        x1 = x + a @ blk["attn.out_proj.weight"].t() + blk["attn.out_proj.bias"]
        h2 = ln_fwd(x1, blk["ln_2.weight"], blk["ln_2.bias"])
        u = h2 @ blk["mlp.c_fc.weight"].t() + blk["mlp.c_fc.bias"]
        return x1 + qgelu(u) @ blk["mlp.c_proj.weight"].t() + blk["mlp.c_proj.bias"], k, v

    def cache_text_kv(self):
        """One-off pass over the frozen text tokens (positions < len_c) of every
        class: per layer, per class, K and V.  Independent of prompts and images."""
        m = self.m
        per_layer: List[Tuple[list, list]] = [([], []) for _ in m.text_blocks]
        for c in range(m.text_x.shape[0]):
            n = int(m.len_prompts[c])
            x = m.text_x[c, :n].clone()
            klim = torch.arange(1, n + 1)
            for l, blk in enumerate(m.text_blocks):
                x, k, v = self._frozen_block(x, blk, m.heads_t, klim)
                per_layer[l][0].append(k)
                per_layer[l][1].append(v)
        self.text_kv = per_layer
        return per_layer

    # -- full step --------------------------------------------------------------
    def step(self, image, label, text_prompt, img_prompt):
        """Returns dict(logits, loss, g_text, g_img, text_f, img_f)."""
        m = self.m
        image = _t(image).float()
        text_prompt = _t(text_prompt).float()
        img_prompt = _t(img_prompt).float()
        B, K = image.shape[0], self.K
        sd = m.sd
        if self.text_kv is None:
            self.cache_text_kv()
        n_cls = m.text_x.shape[0]

        # text tower, prompt rows only
        tsave: list = []
        xt = text_prompt[None].repeat(n_cls, 1, 1)
        for l, blk in enumerate(m.text_blocks):
            xt = self._block_rows_fwd(xt, self.text_kv[l][0], self.text_kv[l][1], blk, m.heads_t, tsave)
        text_f = ln_fwd(xt, sd["ln_final.weight"], sd["ln_final.bias"]) @ sd["text_projection"]

        # image tower: frozen rows (inference) + prompt rows
        w = sd["visual.conv1.weight"].reshape(m.d_v, -1)
        g = image.shape[-1] // m.patch
        patches = image.reshape(B, 3, g, m.patch, g, m.patch).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
        xf = torch.cat([sd["visual.class_embedding"].repeat(B, 1, 1), patches @ w.t()], dim=1)
        xf = xf + sd["visual.positional_embedding"]
        x_pre_prompt = img_prompt[None].repeat(B, 1, 1)
        xf = ln_fwd(xf, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
        xp = ln_fwd(x_pre_prompt, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
        isave: list = []
        ikv: list = []
        for blk in m.img_blocks:
            ks, vs, nxt = [], [], []
            for b in range(B):
                xo, k, v = self._frozen_block(xf[b], blk, m.heads_v)
                ks.append(k); vs.append(v); nxt.append(xo)
            ikv.append((ks, vs))
            xp = self._block_rows_fwd(xp, ks, vs, blk, m.heads_v, isave)
            xf = torch.stack(nxt)
        y_post = ln_fwd(xp, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
        img_f = y_post @ sd["visual.proj"]

        logits, loss, d_img_f, d_text_f = head_fwd_bwd(img_f, text_f, label, float(sd["logit_scale"].exp()))
        out = dict(logits=logits, loss=loss, text_f=text_f, img_f=img_f, g_text=None, g_img=None)
        if label is None:
            return out

        # backward, image prompt rows
        dx = ln_bwd(d_img_f @ sd["visual.proj"].t(), xp, sd["visual.ln_post.weight"])
        for l in reversed(range(len(m.img_blocks))):
            dx = self._block_rows_bwd(dx, isave[l], ikv[l][0], ikv[l][1], m.img_blocks[l], m.heads_v)
        dx = ln_bwd(dx, x_pre_prompt, sd["visual.ln_pre.weight"])
        out["g_img"] = dx.sum(0)          # img_prompt.repeat(B,1,1): grads add over the batch

        # backward, text prompt rows
        dx = ln_bwd(d_text_f @ sd["text_projection"].t(), xt, sd["ln_final.weight"])
        for l in reversed(range(len(m.text_blocks))):
            dx = self._block_rows_bwd(dx, tsave[l], self.text_kv[l][0], self.text_kv[l][1],
                                      m.text_blocks[l], m.heads_t)
        out["g_text"] = dx.sum(0)         # same prompt row written into every class
        return out
