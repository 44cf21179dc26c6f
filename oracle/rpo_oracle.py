"""ORACLE (test infrastructure, not product): dense CPU restatement of RPO's train step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``rpo_amd``) never does and fails
loudly when its HIP library is missing.

What it restates (all citations into /root/reference):

* ``CustomCLIP.make_prompts``  trainers/rpo.py:132-138
* ``CustomCLIP.define_mask``   trainers/rpo.py:140-159
* ``CustomCLIP.forward``       trainers/rpo.py:161-232
* ``ResidualAttentionBlock``   clip/model.py:167-191  (+ LayerNorm :153-159,
  QuickGELU :162-164, Transformer :194-207)
* one SGD train step           trainers/rpo.py:306-309

The arithmetic of the reference lives in PyTorch (``nn.MultiheadAttention``,
``nn.LayerNorm``, ``nn.Linear``, ``nn.Conv2d``, ``F.cross_entropy``,
``torch.optim.SGD``; unpinned -- requirements.txt names no torch).  This
restatement spells those ops out with plain fp32 tensor algebra on the CPU
(matmul / softmax / mean / var), dimension-generic so ViT-L/14 is covered, and
keeps the reference's *dense* structure: every row of both towers goes through
every block and autograd back-props through all of them.  It is therefore also
the CPU baseline whose cost matches the reference's.

Pinning: ``tools/make_golden.py`` imports the real reference in the build
container, feeds it the same generated weights/inputs, and commits its outputs
under tests/golden/; ``tests/test_oracle_golden.py`` checks this file against
them (<= 2e-5 abs on logits at logit-scale 100, ~1e-6 relative on grads).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HEAD_DIM = 64
LN_EPS = 1e-5


def _t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------------------
# primitive ops, spelled out
# ---------------------------------------------------------------------------

def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """clip/model.py:156-159: always computed in fp32, biased variance."""
    x32 = x.float()
    mu = x32.mean(dim=-1, keepdim=True)
    var = ((x32 - mu) ** 2).mean(dim=-1, keepdim=True)
    return ((x32 - mu) * torch.rsqrt(var + eps) * w + b).to(x.dtype)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """clip/model.py:162-164."""
    return x * torch.sigmoid(1.702 * x)


def mha(x: torch.Tensor, in_w, in_b, out_w, out_b, n_head: int, mask: torch.Tensor) -> torch.Tensor:
    """nn.MultiheadAttention(x, x, x, attn_mask=mask) for x[L, N, D]
    (clip/model.py:186): packed in-proj ordered q,k,v; heads of 64; scale 1/8;
    additive float mask [N*H, L, L] or [L, L]; softmax; out-proj."""
    L, N, D = x.shape
    qkv = x @ in_w.t() + in_b                                   # [L,N,3D]
    q, k, v = qkv.split(D, dim=-1)
    hd = D // n_head
    # [L, N*H, hd] -> [N*H, L, hd]  (torch's head-major flattening: index n*H + h)
    q = q.reshape(L, N * n_head, hd).transpose(0, 1)
    k = k.reshape(L, N * n_head, hd).transpose(0, 1)
    v = v.reshape(L, N * n_head, hd).transpose(0, 1)
    s = (q @ k.transpose(1, 2)) * (1.0 / math.sqrt(hd)) + mask  # [N*H, L, L]
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(0, 1).reshape(L, N, D)
    return o @ out_w.t() + out_b


def res_block(x: torch.Tensor, blk: Dict[str, torch.Tensor], n_head: int, mask: torch.Tensor) -> torch.Tensor:
    """clip/model.py:188-191."""
    h = layer_norm(x, blk["ln_1.weight"], blk["ln_1.bias"])
    x = x + mha(h, blk["attn.in_proj_weight"], blk["attn.in_proj_bias"],
                blk["attn.out_proj.weight"], blk["attn.out_proj.bias"], n_head, mask)
    h = layer_norm(x, blk["ln_2.weight"], blk["ln_2.bias"])
    u = h @ blk["mlp.c_fc.weight"].t() + blk["mlp.c_fc.bias"]
    x = x + (quick_gelu(u) @ blk["mlp.c_proj.weight"].t() + blk["mlp.c_proj.bias"])
    return x


# ---------------------------------------------------------------------------
# model container
# ---------------------------------------------------------------------------

def _blocks(sd: Dict[str, torch.Tensor], prefix: str) -> List[Dict[str, torch.Tensor]]:
    n = 0
    while f"{prefix}{n}.ln_1.weight" in sd:
        n += 1
    out = []
    for l in range(n):
        p = f"{prefix}{l}."
        out.append({k[len(p):]: v for k, v in sd.items() if k.startswith(p)})
    return out


@dataclass
class OracleOutput:
    logits: torch.Tensor
    loss: Optional[torch.Tensor]
    text_f: torch.Tensor
    img_f: torch.Tensor


class OracleRPO:
    """CPU counterpart of ``CustomCLIP`` (trainers/rpo.py:93-232) built from a
    state dict with the reference's key names."""

    def __init__(self, state_dict, tokens, K: int, patch: int):
        sd = {k: _t(v).float() for k, v in state_dict.items()}
        self.sd = sd
        self.K = int(K)
        self.patch = int(patch)
        self.tokens = _t(tokens).long()
        self.d_t = sd["ln_final.weight"].shape[0]
        self.d_v = sd["visual.class_embedding"].shape[0]
        self.embed = sd["text_projection"].shape[1]
        self.heads_t = self.d_t // HEAD_DIM
        self.heads_v = self.d_v // HEAD_DIM
        self.n_frozen = sd["visual.positional_embedding"].shape[0]
        self.context = sd["positional_embedding"].shape[0]
        self.text_blocks = _blocks(sd, "transformer.resblocks.")
        self.img_blocks = _blocks(sd, "visual.transformer.resblocks.")
        # make_prompts (trainers/rpo.py:135-137)
        self.text_x = sd["token_embedding.weight"][self.tokens] + sd["positional_embedding"]
        self.len_prompts = self.tokens.argmax(dim=-1) + 1
        assert int(self.len_prompts.max()) + self.K <= self.context
        self.text_mask, self.visual_mask = self._masks()
        self.text_prompt: torch.Tensor = torch.zeros(self.K, self.d_t)
        self.img_prompt: torch.Tensor = torch.zeros(self.K, self.d_v)

    # trainers/rpo.py:140-159
    def _masks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        T = self.context
        ninf = float("-inf")
        per_cls = []
        for idx in self.len_prompts.tolist():
            m = torch.full((T, T), ninf).triu_(1)
            m[:, idx:] = ninf
            per_cls.append(m.unsqueeze(0).expand(self.heads_t, T, T))
        text_mask = torch.cat(per_cls, dim=0).contiguous()
        S = self.n_frozen + self.K
        vis = torch.zeros(S, S)
        vis[:, self.n_frozen:] = ninf
        return text_mask, vis

    def set_prompts(self, text_prompt, img_prompt) -> None:
        self.text_prompt = _t(text_prompt).float().clone().requires_grad_(True)
        self.img_prompt = _t(img_prompt).float().clone().requires_grad_(True)

    # -- towers ------------------------------------------------------------
    def text_tower(self, text_prompt: torch.Tensor, return_rows: bool = False):
        """trainers/rpo.py:172-191 -> text_f[n_cls, K, e]."""
        n_cls = self.text_x.shape[0]
        ar = torch.arange(n_cls)
        x = self.text_x.clone()
        for i in range(self.K):
            x[ar, self.len_prompts + i, :] = text_prompt[i, :].repeat(n_cls, 1)
        x = x.permute(1, 0, 2)
        rows = []
        for blk in self.text_blocks:
            x = res_block(x, blk, self.heads_t, self.text_mask)
            if return_rows:
                xb = x.permute(1, 0, 2)
                rows.append(torch.stack([xb[ar, self.len_prompts + i] for i in range(self.K)], dim=1))
        x = x.permute(1, 0, 2)
        x = layer_norm(x, self.sd["ln_final.weight"], self.sd["ln_final.bias"])
        f = torch.stack([x[ar, self.len_prompts + i] for i in range(self.K)], dim=1)
        f = f @ self.sd["text_projection"]
        return (f, rows) if return_rows else f

    def image_tower(self, image: torch.Tensor, img_prompt: torch.Tensor, return_rows: bool = False):
        """trainers/rpo.py:195-210 -> img_f[B, K, e]."""
        B = image.shape[0]
        w = self.sd["visual.conv1.weight"]
        emb = F.conv2d(image, w, stride=self.patch)               # [B, d, g, g]
        emb = emb.reshape(B, emb.shape[1], -1).permute(0, 2, 1)   # [B, g*g, d]
        cls = self.sd["visual.class_embedding"].repeat(B, 1, 1)
        x = torch.cat([cls, emb], dim=1) + self.sd["visual.positional_embedding"]
        x = torch.cat([x, img_prompt.repeat(B, 1, 1)], dim=1)
        x = layer_norm(x, self.sd["visual.ln_pre.weight"], self.sd["visual.ln_pre.bias"])
        x = x.permute(1, 0, 2)
        rows = []
        for blk in self.img_blocks:
            x = res_block(x, blk, self.heads_v, self.visual_mask)
            if return_rows:
                rows.append(x.permute(1, 0, 2)[:, -self.K:, :])
        x = x.permute(1, 0, 2)
        f = layer_norm(x[:, -self.K:, :], self.sd["visual.ln_post.weight"], self.sd["visual.ln_post.bias"])
        f = f @ self.sd["visual.proj"]
        return (f, rows) if return_rows else f

    # -- head (trainers/rpo.py:215-230) -------------------------------------
    def head(self, img_f: torch.Tensor, text_f: torch.Tensor) -> torch.Tensor:
        text_f = text_f / text_f.norm(dim=-1, keepdim=True)
        img_f = img_f / img_f.norm(dim=-1, keepdim=True)
        scale = self.sd["logit_scale"].exp()
        logits = torch.zeros(img_f.shape[0], text_f.shape[0])
        for i in range(self.K):
            logits = logits + scale * img_f[:, i, :] @ text_f[:, i, :].t()
        return logits / self.K

    def forward(self, image, label=None) -> OracleOutput:
        image = _t(image).float()
        text_f = self.text_tower(self.text_prompt)
        img_f = self.image_tower(image, self.img_prompt)
        logits = self.head(img_f, text_f)
        loss = None
        if label is not None:
            loss = F.cross_entropy(logits, _t(label).long())
        return OracleOutput(logits=logits, loss=loss, text_f=text_f, img_f=img_f)

    def loss_and_grads(self, image, label):
        """forward -> zero_grad -> backward (trainers/rpo.py:306-308)."""
        for p in (self.text_prompt, self.img_prompt):
            p.grad = None
        out = self.forward(image, label)
        out.loss.backward()
        return out, self.text_prompt.grad.clone(), self.img_prompt.grad.clone()


# ---------------------------------------------------------------------------
# the unmasked towers: plain CLIP inference (clip/model.py:344-372), as trainers/zsclip.py:58-63 and the sibling
# trainers' CustomCLIP.forward (trainers/coop.py:196-208) use them
# ---------------------------------------------------------------------------
def coop_prompts(emb_t: torch.Tensor, tokens: torch.Tensor, ctx: torch.Tensor, class_token_position: str = "end") -> torch.Tensor:
    """PromptLearner.forward of trainers/coop.py:117-183: the token embeddings `emb_t` [n_cls, T, d] of the
    "X X .. name." prompts with the n_ctx placeholder slots replaced by `ctx` ([n_ctx, d] generic, expanded over the
    classes :119-121, or [n_cls, n_ctx, d] class-specific :84-86), the class-name tokens placed per
    CLASS_TOKEN_POSITION: "end" [SOS | ctx | name . EOT ..] (:126-134), "middle" [SOS | ctx[:n/2] | name | ctx[n/2:] | . EOT ..]
    (:136-159), "front" [SOS | name | ctx | . EOT ..] (:161-181).  name_len = len(_tokenizer.encode(name)) (:99) = the
    prompt's EOT index - n_ctx - 2."""
    n_cls = emb_t.shape[0]
    n_ctx = ctx.shape[-2]
    if ctx.dim() == 2:
        ctx = ctx.unsqueeze(0).expand(n_cls, -1, -1)
    prefix, suffix = emb_t[:, :1], emb_t[:, 1 + n_ctx:]
    if class_token_position == "end":
        return torch.cat([prefix, ctx, suffix], dim=1)
    name_lens = (tokens.argmax(dim=-1) - n_ctx - 2).tolist()
    half = n_ctx // 2
    out = []
    for i in range(n_cls):
        nl = int(name_lens[i])
        cls_i, suf_i = suffix[i:i + 1, :nl], suffix[i:i + 1, nl:]
        if class_token_position == "middle":
            parts = [prefix[i:i + 1], ctx[i:i + 1, :half], cls_i, ctx[i:i + 1, half:], suf_i]
        elif class_token_position == "front":
            parts = [prefix[i:i + 1], cls_i, ctx[i:i + 1], suf_i]
        else:
            raise ValueError(class_token_position)
        out.append(torch.cat(parts, dim=1))
    return torch.cat(out, dim=0)


def plain_clip_forward(state_dict, image, tokens, patch: int, ctx=None, class_token_position: str = "end"):
    """CLIP.forward(image, text) -> (logits_per_image [B, n_cls], image_features [B, e], text_features [n_cls, e]);
    features before normalisation.  Image tower: every token reads every token (clip/model.py:227-240); text tower:
    causal mask over the whole context (:287-292, :347-360), feature at the EOT position (= argmax of the ids).
    ctx [n_ctx, d_t]: CoOp's learned context (trainers/coop.py:117-134, class token at the end, generic context): the
    embeddings of positions 1 .. n_ctx of every class are replaced by it before the positional embedding is added
    (TextEncoder.forward, :47-58)."""
    sd = {k: _t(v).float() for k, v in state_dict.items()}
    image, tokens = _t(image).float(), _t(tokens).long()
    d_t, d_v = sd["ln_final.weight"].shape[0], sd["visual.class_embedding"].shape[0]
    B = image.shape[0]
    emb = F.conv2d(image, sd["visual.conv1.weight"], stride=patch)
    emb = emb.reshape(B, emb.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].repeat(B, 1, 1), emb], dim=1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]).permute(1, 0, 2)
    S = x.shape[0]
    for blk in _blocks(sd, "visual.transformer.resblocks."):
        x = res_block(x, blk, d_v // HEAD_DIM, torch.zeros(S, S))
    img_f = layer_norm(x.permute(1, 0, 2)[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
    emb_t = sd["token_embedding.weight"][tokens]
    if ctx is not None:
        ctx = ctx if isinstance(ctx, torch.Tensor) else _t(ctx).float()
        emb_t = coop_prompts(emb_t, tokens, ctx, class_token_position)
    t = (emb_t + sd["positional_embedding"]).permute(1, 0, 2)
    T = t.shape[0]
    causal = torch.full((T, T), float("-inf")).triu_(1)
    for blk in _blocks(sd, "transformer.resblocks."):
        t = res_block(t, blk, d_t // HEAD_DIM, causal)
    t = layer_norm(t.permute(1, 0, 2), sd["ln_final.weight"], sd["ln_final.bias"])
    txt_f = t[torch.arange(t.shape[0]), tokens.argmax(dim=-1)] @ sd["text_projection"]
    a = img_f / img_f.norm(dim=-1, keepdim=True)
    b = txt_f / txt_f.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * a @ b.t(), img_f, txt_f


def coop_loss_and_grad(state_dict, image, tokens, ctx, label, patch: int, class_token_position: str = "end"):
    """trainers/coop.py:266-270 (fp32 branch): logits, cross-entropy and d loss / d ctx (ctx [n_ctx, d] or, class-specific,
    [n_cls, n_ctx, d])."""
    c = _t(ctx).float().clone().requires_grad_(True)
    logits, _, _ = plain_clip_forward(state_dict, image, tokens, patch, ctx=c, class_token_position=class_token_position)
    loss = F.cross_entropy(logits, _t(label).long())
    loss.backward()
    return logits.detach(), loss.detach(), c.grad.clone()


def cocoop_loss_and_grads(state_dict, image, tokens, ctx, meta, label, patch: int):
    """trainers/cocoop.py:166-192 (CustomCLIP.forward) + :262-265 (fp32 branch): instance-conditioned context.
    image features (plain image tower, L2-normalised, :173-174) -> meta_net = linear1 [e/16, e] -> ReLU -> linear2
    [d_t, e/16] (:93-97) -> bias[b]; prompts of image b = [SOS | ctx + bias[b] | name . EOT] for every class (:137-153);
    logits[b] = exp(logit_scale) * imf_n[b] @ normalise(text_features(prompts_b))^T (:183-187); loss = mean CE.
    meta: dict with w1 [h, e], b1 [h], w2 [d_t, h], b2 [d_t].  Returns (logits [B, n_cls], loss, {"ctx", "w1", "b1", "w2",
    "b2"} gradients)."""
    sd = {k: _t(v).float() for k, v in state_dict.items()}
    image, tokens = _t(image).float(), _t(tokens).long()
    d_t, d_v = sd["ln_final.weight"].shape[0], sd["visual.class_embedding"].shape[0]
    B = image.shape[0]
    with torch.no_grad():                                            # frozen image tower (:220-222 freeze everything else)
        emb = F.conv2d(image, sd["visual.conv1.weight"], stride=patch)
        emb = emb.reshape(B, emb.shape[1], -1).permute(0, 2, 1)
        x = torch.cat([sd["visual.class_embedding"].repeat(B, 1, 1), emb], dim=1) + sd["visual.positional_embedding"]
        x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]).permute(1, 0, 2)
        S = x.shape[0]
        for blk in _blocks(sd, "visual.transformer.resblocks."):
            x = res_block(x, blk, d_v // HEAD_DIM, torch.zeros(S, S))
        img_f = layer_norm(x.permute(1, 0, 2)[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
        imf = img_f / img_f.norm(dim=-1, keepdim=True)
    P = {k: _t(v).float().clone().requires_grad_(True) for k, v in dict(meta, ctx=ctx).items()}
    bias = F.linear(F.relu(F.linear(imf, P["w1"], P["b1"])), P["w2"], P["b2"])            # [B, d_t]
    emb_t = sd["token_embedding.weight"][tokens]                                          # [n_cls, T, d_t]
    n_cls, T, n_ctx = emb_t.shape[0], emb_t.shape[1], P["ctx"].shape[0]
    causal = torch.full((T, T), float("-inf")).triu_(1)
    logits = []
    for b in range(B):
        c = (P["ctx"] + bias[b][None]).unsqueeze(0).expand(n_cls, -1, -1)
        t = torch.cat([emb_t[:, :1], c, emb_t[:, 1 + n_ctx:]], dim=1)
        t = (t + sd["positional_embedding"]).permute(1, 0, 2)
        for blk in _blocks(sd, "transformer.resblocks."):
            t = res_block(t, blk, d_t // HEAD_DIM, causal)
        t = layer_norm(t.permute(1, 0, 2), sd["ln_final.weight"], sd["ln_final.bias"])
        tf = t[torch.arange(n_cls), tokens.argmax(dim=-1)] @ sd["text_projection"]
        tf = tf / tf.norm(dim=-1, keepdim=True)
        logits.append(sd["logit_scale"].exp() * imf[b] @ tf.t())
    logits = torch.stack(logits)
    loss = F.cross_entropy(logits, _t(label).long())
    loss.backward()
    return logits.detach(), loss.detach(), {k: v.grad.clone() for k, v in P.items()}


# ---------------------------------------------------------------------------
# optimiser (torch.optim.SGD restated; Dassl's defaults are not in the tree, so
# momentum / weight decay / dampening are explicit -- SURVEY.md section 8c)
# ---------------------------------------------------------------------------

class OracleSGD:
    def __init__(self, lr: float, momentum: float = 0.9, weight_decay: float = 5e-4,
                 dampening: float = 0.0, nesterov: bool = False):
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.dampening, self.nesterov = dampening, nesterov
        self.buf: Dict[int, torch.Tensor] = {}

    @torch.no_grad()
    def step(self, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor]) -> None:
        for idx, (p, g) in enumerate(zip(params, grads)):
            g = g.clone()
            if self.wd != 0.0:
                g = g + self.wd * p
            if self.momentum != 0.0:
                if idx not in self.buf:
                    self.buf[idx] = g.clone()
                else:
                    self.buf[idx].mul_(self.momentum).add_(g, alpha=1.0 - self.dampening)
                g = g + self.momentum * self.buf[idx] if self.nesterov else self.buf[idx]
            p.add_(g, alpha=-self.lr)


def cosine_lr_with_constant_warmup(base_lr: float, epoch: int, max_epoch: int,
                                   warmup_epoch: int = 1, warmup_lr: float = 1e-5) -> float:
    """LR in force during ``epoch`` (0-based) for Dassl's ConstantWarmupScheduler
    wrapping CosineAnnealingLR(T_max=max_epoch)
    (configs/trainers/RPO/main_K24.yaml:15-22; the scheduler code is in the
    un-vendored Dassl -- semantics restated from its published source:
    epochs < warmup use the constant LR; the wrapper's step() forwards to the
    successor only once last_epoch >= warmup_epoch, so the cosine successor is
    evaluated at epoch - warmup_epoch, not at the global epoch index)."""
    if epoch < warmup_epoch:
        return warmup_lr
    return 0.5 * base_lr * (1.0 + math.cos(math.pi * (epoch - warmup_epoch) / max_epoch))


def train_steps(model: OracleRPO, opt: OracleSGD, batches) -> List[float]:
    """Run RPO.forward_backward (trainers/rpo.py:290-316) for each (image,label)."""
    losses = []
    for image, label in batches:
        out, gt, gi = model.loss_and_grads(image, label)
        opt.step([model.text_prompt, model.img_prompt], [gt, gi])
        losses.append(float(out.loss))
    return losses
