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
from .engine import Engine


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


class CustomCLIP(nn.Module):
    def __init__(self, cfg: RPOConfig, state_dict: Dict[str, np.ndarray], tokens: Optional[np.ndarray] = None,
                 device: str | torch.device = "cuda:0", act_dtype: torch.dtype = torch.bfloat16,
                 max_batch: int = 32, prompts: Optional[Sequence[np.ndarray]] = None, prompt_seed: int = 7):
        super().__init__()
        self.cfg = cfg
        device = torch.device(device)
        tokens = synth.default_tokens(cfg) if tokens is None else tokens
        self.engine = Engine(cfg, state_dict, tokens, device, act_dtype, max_batch)
        self.prompt_learner = PromptLearner(self.engine)
        tp, ip = prompts if prompts is not None else synth.prompts(cfg, state_dict, prompt_seed)
        with torch.no_grad():
            self.prompt_learner.text_prompt.copy_(torch.from_numpy(np.asarray(tp, dtype=np.float32)))
            self.prompt_learner.img_prompt.copy_(torch.from_numpy(np.asarray(ip, dtype=np.float32)))
        self.len_prompts = self.engine.len_np

    def forward(self, image: torch.Tensor, label: Optional[torch.Tensor] = None):
        image = image.to(device=self.engine.dev, dtype=torch.float32).contiguous()
        if self.prompt_learner.training:
            if label is None:
                raise ValueError("training forward needs labels (trainers/rpo.py:229-230)")
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
