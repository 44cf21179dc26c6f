"""``PromptLearner`` / ``CustomCLIP`` with the reference's surface
(trainers/rpo.py:41-90, :93-232) on top of the HIP engine.

``model(image, label)`` returns the scalar loss while ``prompt_learner`` is in
training mode and ``logits[B, n_cls]`` otherwise, exactly like the reference
(:229-232).  The loss is connected to the two prompt parameters through a
``torch.autograd.Function`` whose backward hands back the gradients the HIP
kernels already produced, so the reference's
``optim.zero_grad(); loss.backward(); optim.step()`` (:306-309) works unchanged;
``rpo_amd.trainer.RPO`` uses the fused graph + ``rpo_sgd_step`` path instead.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import synth
from .config import RPOConfig
from .engine import Engine, make_engine


class PromptLearner(nn.Module):
    """K text prompts [K, d_t] and K image prompts [K, d_v] -- the only trainable
    state (trainers/rpo.py:69, :83).  Both alias one flat fp32 HBM buffer owned by
    the engine so a single all-reduce / SGD launch covers them."""

    def __init__(self, engine: Engine):
        super().__init__()
        self.K = engine.cfg.K
        self.text_prompt = nn.Parameter(engine.text_prompt)
        self.img_prompt = nn.Parameter(engine.img_prompt)
        assert self.text_prompt.data_ptr() == engine.text_prompt.data_ptr()

    def forward(self):
        return self.text_prompt, self.img_prompt


class _StepFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_prompt, img_prompt, engine: Engine, image, label):
        engine.forward_backward(image, label)
        ctx.save_for_backward(engine.g_text.clone(), engine.g_img.clone())
        return engine.loss.clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g_text, g_img = ctx.saved_tensors
        return grad_out * g_text, grad_out * g_img, None, None, None


def init_prompts(state_dict: Dict[str, np.ndarray], K: int, d_t: int, d_v: int):
    """PromptLearner.initialization_token (trainers/rpo.py:60-83): EOT-token embedding / class embedding repeated K
    times plus 0.1 x a unit-norm gaussian direction, drawn from torch's GLOBAL CPU generator in the reference's order
    (text noise first, then visual noise): from the same generator state the reference's constructor draws the same
    prompts (pinned by tests/golden/ref_init_seed3_d1_k4.npz, seed set right before construction; in a full reference
    run CLIP's own constructor consumes the generator first, which this repo does not replay)."""
    text = torch.from_numpy(np.asarray(state_dict["token_embedding.weight"][49407], dtype=np.float32)).repeat(K, 1)
    noise = torch.randn(K, d_t)
    text = text + 0.1 * (noise / noise.norm(dim=-1, keepdim=True))
    vis = torch.from_numpy(np.asarray(state_dict["visual.class_embedding"], dtype=np.float32)).repeat(K, 1)
    noise = torch.randn(K, d_v)
    vis = vis + 0.1 * (noise / noise.norm(dim=-1, keepdim=True))
    return text.numpy(), vis.numpy()


def config_from_state_dict(sd, K: int, n_cls: int, name: Optional[str] = None) -> RPOConfig:
    """Tower dimensions from a CLIP state dict, the way clip/model.py:403-432 `build_model` infers them."""
    d_v = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    layers_v = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed = sd["text_projection"].shape[1]
    context, d_t = sd["positional_embedding"].shape
    vocab = sd["token_embedding.weight"].shape[0]
    layers_t = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    if name is None:
        name = {(768, 16): "ViT-B/16", (768, 32): "ViT-B/32", (1024, 14): "ViT-L/14"}.get((d_v, patch), f"ViT-{d_v}/{patch}")
    return RPOConfig(name=name, image_size=grid * patch, patch=patch, d_v=d_v, layers_v=layers_v, d_t=d_t,
                     layers_t=layers_t, context=context, vocab=vocab, embed=embed, K=K, n_cls=n_cls)


class CustomCLIP(nn.Module):
    """Two ways in.  The native one takes what the engine needs (RPOConfig, a CLIP state dict of numpy arrays, the
    token ids).  The reference's own call shape, `CustomCLIP(cfg, classnames, prompt, clipmodel)`
    (trainers/rpo.py:99, called at :255), is accepted as well -- see `_from_reference_args`."""

    def __init__(self, cfg, state_dict=None, tokens=None, device: str | torch.device = "cuda:0",
                 act_dtype: Optional[torch.dtype] = None, max_batch: int = 32,
                 prompts: Optional[Sequence[np.ndarray]] = None, tokenize=None):
        super().__init__()
        if not isinstance(cfg, RPOConfig):
            cfg, state_dict, tokens, act_dtype = self._from_reference_args(cfg, state_dict, tokens, device, act_dtype,
                                                                           tokenize)
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if act_dtype is None:
            act_dtype = torch.bfloat16
        self.cfg = cfg
        device = torch.device(device)
        if tokens is None:
            if cfg.n_cls != 19:
                raise ValueError("tokens [n_cls, context] are required for any class set other than the bundled "
                                 "Oxford-Pets base split (synthetic ids are for tests and bench only)")
            tokens = synth.oxford_pets_base_tokens()
        self.engine = make_engine(cfg, state_dict, tokens, device, act_dtype, max_batch)
        self.prompt_learner = PromptLearner(self.engine)
        # trainers/rpo.py:63-67,77-81: drawn from torch's seeded global generator unless injected
        tp, ip = prompts if prompts is not None else init_prompts(state_dict, cfg.K, cfg.d_t, cfg.d_v)
        with torch.no_grad():
            self.prompt_learner.text_prompt.copy_(torch.from_numpy(np.asarray(tp, dtype=np.float32)))
            self.prompt_learner.img_prompt.copy_(torch.from_numpy(np.asarray(ip, dtype=np.float32)))
        self.len_prompts = self.engine.len_np

    @staticmethod
    def _from_reference_args(cfg, classnames, prompt, clipmodel, act_dtype, tokenize):
        """`CustomCLIP(cfg, classnames, prompt, clipmodel)` as trainers/rpo.py:255 calls it.

        cfg        the reference's yacs node (only TRAINER.RPO.K / .PREC and INPUT.SIZE are read, :47-56,247)
        classnames list of class names; prompt: the template with "_" as the slot (:133)
        clipmodel  anything with `.state_dict()` in CLIP's key layout (the reference's clip.model.CLIP), or the
                   state dict itself
        tokenize   the reference's `clip.tokenize` (any callable str -> [1, 77] ids).  The BPE tokenizer is out of
                   scope here; without it only the bundled Oxford-Pets base prompts can be looked up."""
        from .config import OXFORD_PETS_BASE_CLASSES, PROMPT_TEMPLATE, act_dtype_for_prec
        sd_t = clipmodel.state_dict() if hasattr(clipmodel, "state_dict") else clipmodel
        sd = {k: (v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32))
              for k, v in sd_t.items()}
        K = int(cfg.TRAINER.RPO.K)
        assert K >= 1, "K should be bigger than 0"                              # trainers/rpo.py:47
        classnames = list(classnames)
        rcfg = config_from_state_dict(sd, K, len(classnames))
        assert int(cfg.INPUT.SIZE[0]) == rcfg.image_size, \
            f"cfg_imsize ({cfg.INPUT.SIZE[0]}) must equal to clip_imsize ({rcfg.image_size})"   # :56
        texts = [prompt.replace("_", c) for c in classnames]                    # :133
        if tokenize is not None:
            toks = np.concatenate([np.asarray(tokenize(t)).reshape(1, -1) for t in texts]).astype(np.int64)
        elif prompt == PROMPT_TEMPLATE and tuple(classnames) == OXFORD_PETS_BASE_CLASSES:
            toks = synth.oxford_pets_base_tokens()
        else:
            raise ValueError("pass tokenize=clip.tokenize: only the Oxford-Pets base prompts are bundled as token ids")
        if act_dtype is None:
            prec = getattr(getattr(cfg.TRAINER, "RPO", None), "PREC", "fp16")
            act_dtype = act_dtype_for_prec(prec)
        return rcfg, sd, toks, act_dtype

    def forward(self, image: torch.Tensor, label: Optional[torch.Tensor] = None):
        with torch.cuda.device(self.engine.dev):
            return self._forward(image, label)

    def _forward(self, image: torch.Tensor, label: Optional[torch.Tensor] = None):
        image = image.to(device=self.engine.dev, dtype=torch.float32).contiguous()
        if self.prompt_learner.training:
            if label is None:
                raise ValueError("training forward needs labels (trainers/rpo.py:229-230)")
            if not label.is_cuda and label.numel() and (int(label.min()) < 0 or int(label.max()) >= self.cfg.n_cls):
                raise IndexError(f"Target out of bounds for n_cls = {self.cfg.n_cls}")   # as F.cross_entropy (:230)
            label = label.to(device=self.engine.dev, dtype=torch.int64)
            tp, ip = self.prompt_learner()
            return _StepFunction.apply(tp, ip, self.engine, image, label)
        with torch.no_grad():
            # prompts may have been edited through the nn.Parameter views: a cheap version key
            ver = (self.prompt_learner.text_prompt._version, self.prompt_learner.img_prompt._version)
            if ver != getattr(self, "_seen_version", None):
                self.engine.params_version += 1
                self._seen_version = ver
            return self.engine.forward_eval(image).clone()
