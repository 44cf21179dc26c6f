"""Zero-shot CLIP inference on the HIP engine: the host-side mirror of `trainers/zsclip.py` (ZeroshotCLIP.build_model /
model_inference, :31-63) and of the unmasked towers the sibling trainers call (`trainers/coop.py:196-208`).

    logits = exp(logit_scale) * normalise(encode_image(image)) @ normalise(encode_text(prompts)).T

Nothing is trained here.  The class-name prompts are tokenised by the caller (the BPE tokenizer is out of scope; the
Oxford-Pets base prompts are bundled as ids); their text features are computed once, as the reference does (:48-53).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .config import RPOConfig
from .custom_clip import config_from_state_dict
from .engine import Engine, make_engine
from . import synth


class ZeroshotCLIP:
    def __init__(self, state_dict: Dict[str, np.ndarray], tokens: Optional[np.ndarray] = None,
                 device: str | torch.device = "cuda:0", act_dtype: torch.dtype = torch.float16, max_batch: int = 100,
                 cfg: Optional[RPOConfig] = None):
        """state_dict: CLIP weights under the reference's key names (numpy); tokens: int64 [n_cls, 77] prompt ids
        (default: the bundled Oxford-Pets base prompts); max_batch: the reference's test batch is 100
        (configs/trainers/RPO/main_K24.yaml:5)."""
        if tokens is None:
            tokens = synth.oxford_pets_base_tokens()
        tokens = np.asarray(tokens, dtype=np.int64)
        if cfg is None:
            cfg = config_from_state_dict(state_dict, 1, tokens.shape[0])     # one (unused) prompt row per image
        self.cfg = cfg
        self.engine = make_engine(cfg, state_dict, tokens, torch.device(device), act_dtype, max_batch)
        with torch.cuda.device(self.engine.dev):
            self.engine.cache_text_kv()                                       # text features: once (zsclip.py:48-53)

    def set_context(self, ctx) -> None:
        """Evaluate CoOp-style learned context vectors (trainers/coop.py:117-134: generic context [n_ctx, d_t], class
        token at the end): `tokens` must then be the ids of the reference's "X X .. name." prompts."""
        with torch.cuda.device(self.engine.dev):
            self.engine.set_context(ctx)
            self.engine.cache_text_kv()

    @torch.no_grad()
    def model_inference(self, image: torch.Tensor) -> torch.Tensor:
        """trainers/zsclip.py:58-63 -> logits [B, n_cls] (fp32, on the device)."""
        eng = self.engine
        with torch.cuda.device(eng.dev):
            image = image.to(device=eng.dev, dtype=torch.float32).contiguous()
            return eng.forward_plain(image).clone()

    __call__ = model_inference
