"""CoOp and CoCoOp on the HIP engine: the host-side mirror of `trainers/coop.py` (PromptLearner :60-134, CustomCLIP :185-208, the
`CoOp` trainer's `forward_backward` :258-281) -- SURVEY.md section 8f rank 4, the sibling trainer that shares RPO's towers
without the read-only mask.

    prompts  = [SOS | ctx (n_ctx learned vectors, shared by all classes) | class name . EOT]      (:117-134, "end")
    logits   = exp(logit_scale) * normalise(encode_image(image)) @ normalise(encode_text(prompts)).T
    loss     = F.cross_entropy(logits, label);   only `ctx` is trained (:228-230)

Unlike RPO's prompts, a context vector is read by every later token, so its gradient needs the DENSE text-tower backward
(all tokens of all classes, causal attention with dK / dV): `Engine.coop_forward_backward`.  The image tower is plain
frozen CLIP and is only run forward.  Generic context (CSC = False) and class token at the end -- the reference's own
defaults (configs/trainers/CoOp/vit_b16_ep50.yaml) -- are what is built; "middle" / "front" raise in the reference
config used here as well (its code path for them exists but is not exercised by the repo's scripts).

CoCoOp (`trainers/cocoop.py`: PromptLearner :60-153, CustomCLIP :156-192, trainer :255-275) adds a meta-net on the
normalised image feature whose output shifts the context PER IMAGE, so every image has its own text features for every
class: `Engine.cocoop_forward_backward` runs the class set once per image (B * n_cls virtual classes) through the same
dense text path and trains ctx + the two meta-net layers (`CoCoOpCustomCLIP`, `CoCoOp` below).

The caller provides the token ids of the "X X .. name." prompts (the BPE tokenizer is out of scope).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .config import RPOConfig
from .custom_clip import config_from_state_dict
from .engine import Engine
from .trainer import OptimConfig, lr_at_epoch


class CoOpPromptLearner:
    """State of `trainers/coop.py:PromptLearner`: `ctx` [n_ctx, d_t] lives in the engine (fp32 master copy on the device);
    `token_prefix` / `token_suffix` are the embeddings of SOS and of "name . EOT ..." that the reference registers as
    buffers (:100-101) and that its checkpoints therefore contain."""

    def __init__(self, engine: Engine, n_ctx: int, token_embedding: np.ndarray, tokens: np.ndarray):
        self.engine, self.n_ctx = engine, n_ctx
        emb = np.asarray(token_embedding)[np.asarray(tokens)]                       # [n_cls, 77, d_t]
        self.token_prefix = torch.from_numpy(np.ascontiguousarray(emb[:, :1]))
        self.token_suffix = torch.from_numpy(np.ascontiguousarray(emb[:, 1 + n_ctx:]))
        self.training = True

    @property
    def ctx(self) -> torch.Tensor:
        return self.engine.coop_ctx

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"ctx": self.ctx.detach().cpu().clone(), "token_prefix": self.token_prefix.clone(),
                "token_suffix": self.token_suffix.clone()}


class CoOpCustomCLIP:
    """`trainers/coop.py:CustomCLIP`: `model(image)` -> logits [B, n_cls] (fp32, on the device)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int,
                 device: str | torch.device = "cuda:0", act_dtype: torch.dtype = torch.float16, max_batch: int = 32,
                 ctx: Optional[np.ndarray] = None, cfg: Optional[RPOConfig] = None):
        tokens = np.asarray(tokenized_prompts, dtype=np.int64)
        if cfg is None:
            cfg = config_from_state_dict(state_dict, 1, tokens.shape[0])     # one (unused) RPO prompt row per image
        self.cfg = cfg
        self.engine = Engine(cfg, state_dict, tokens, torch.device(device), act_dtype, max_batch)
        with torch.cuda.device(self.engine.dev):
            self.engine.coop_setup(n_ctx)
            if ctx is None:
                # "Initializing a generic context": nn.init.normal_(ctx_vectors, std=0.02) from torch's global generator
                # (trainers/coop.py:87-88) -- the same draw, so a seeded run starts from the reference's vectors
                ctx = torch.empty(n_ctx, cfg.d_t).normal_(std=0.02).numpy()
            self.engine.coop_ctx.copy_(torch.as_tensor(np.asarray(ctx, dtype=np.float32)))
        self.prompt_learner = CoOpPromptLearner(self.engine, n_ctx, state_dict["token_embedding.weight"], tokens)
        self.tokenized_prompts = tokens

    def __call__(self, image: torch.Tensor) -> torch.Tensor:
        eng = self.engine
        with torch.cuda.device(eng.dev):
            image = image.to(device=eng.dev, dtype=torch.float32).contiguous()
            return eng.coop_forward_backward(image, None)


class CoOp:
    """The trainer's step (trainers/coop.py:258-281): forward -> cross-entropy -> zero_grad -> backward -> SGD step on
    `ctx`, returning {"loss", "acc"}; per-epoch LR update.  Optimiser hyper-parameters as for RPO (Dassl defaults are
    un-vendored, hence explicit: OptimConfig)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int = 16,
                 optim: Optional[OptimConfig] = None, device: str | torch.device = "cuda:0",
                 act_dtype: torch.dtype = torch.float16, batch_size: int = 32, num_batches: int = 1,
                 ctx: Optional[np.ndarray] = None, cfg: Optional[RPOConfig] = None, use_graph: bool = False):
        self.optim_cfg = optim or OptimConfig(lr=0.002, max_epoch=50)       # configs/trainers/CoOp/vit_b16_ep50.yaml
        self.model = CoOpCustomCLIP(state_dict, tokenized_prompts, n_ctx, device, act_dtype, batch_size, ctx, cfg)
        self.engine, self.cfg = self.model.engine, self.model.cfg
        self.device = self.engine.dev
        self.batch_size, self.num_batches = batch_size, num_batches
        self.epoch = self.batch_idx = self._steps = 0
        self.lr = lr_at_epoch(self.optim_cfg, 0)
        self.use_graph = use_graph
        self._graph = None                               # (HIP graph of one step, the learning rate it was captured with)

    def _enqueue(self, image: torch.Tensor, label: torch.Tensor) -> None:
        eng, oc = self.engine, self.optim_cfg
        eng.coop_forward_backward(image, label)
        ops.sgd_step(eng.coop_params, eng.coop_grads, eng.coop_moms, self.lr, oc.momentum, oc.weight_decay, 1.0,
                     first_step=(self._steps == 0))

    def step_async(self, image: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        """One optimisation step, nothing synchronised; returns the device loss scalar.  With use_graph the ~250 launches
        of a step (plain image tower, dense text tower forward + backward, head, SGD) are replayed from ONE HIP graph,
        captured after the first (eager) step and again whenever the learning rate changes (it is a kernel argument)."""
        if not self.use_graph or self._steps == 0 or image.shape[0] != self.batch_size:
            self._enqueue(image, label)
        else:
            if self._graph is None or self._graph[1] != self.lr:
                self._img = torch.empty_like(image)
                self._lab = torch.empty_like(label)
                self._img.copy_(image); self._lab.copy_(label)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._enqueue(self._img, self._lab)
                self._graph = (g, self.lr)
                # (capture does not execute: the replay below is this step)
            if image.data_ptr() != self._img.data_ptr():
                self._img.copy_(image, non_blocking=True)
            self._lab.copy_(label, non_blocking=True)
            self._graph[0].replay()
        self._steps += 1
        return self.engine.loss

    def parse_batch_train(self, batch):
        img = batch["img"].to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        label = torch.as_tensor(batch["label"])
        if not label.is_cuda:
            lo, hi = int(label.min()), int(label.max())
            if lo < 0 or hi >= self.cfg.n_cls:
                raise IndexError(f"Target {hi if hi >= self.cfg.n_cls else lo} is out of bounds (n_cls = {self.cfg.n_cls})")
        return img, label.to(self.device, dtype=torch.int64, non_blocking=True)

    def forward_backward(self, batch) -> Dict[str, float]:
        eng, oc = self.engine, self.optim_cfg
        with torch.cuda.device(self.device):
            image, label = self.parse_batch_train(batch)
            self.step_async(image, label)
            logits = eng.logits[:image.shape[0]]
            acc = float((logits.argmax(1) == label).float().mean().item()) * 100.0      # compute_accuracy()[0]
            summary = {"loss": float(eng.loss.item()), "acc": acc}
        if (self.batch_idx + 1) == self.num_batches:
            self.epoch += 1
            self.lr = lr_at_epoch(self.optim_cfg, self.epoch)
            self.batch_idx = 0
        else:
            self.batch_idx += 1
        return summary

    @torch.no_grad()
    def model_inference(self, image: torch.Tensor) -> torch.Tensor:
        return self.model(image)


class CoCoOpPromptLearner(CoOpPromptLearner):
    """`trainers/cocoop.py:PromptLearner`: ctx + meta_net (linear1 [e/16, e] -> ReLU -> linear2 [d_t, e/16], :93-97)."""

    def state_dict(self) -> Dict[str, torch.Tensor]:
        w1, b1, w2, b2 = (t.detach().cpu().clone() for t in self.engine.meta)
        return {"ctx": self.ctx.detach().cpu().clone(), "meta_net.linear1.weight": w1, "meta_net.linear1.bias": b1,
                "meta_net.linear2.weight": w2, "meta_net.linear2.bias": b2,
                "token_prefix": self.token_prefix.clone(), "token_suffix": self.token_suffix.clone()}


class CoCoOpCustomCLIP:
    """`trainers/cocoop.py:CustomCLIP`: `model(image)` -> logits [B, n_cls]; `model(image, label)` -> the cross-entropy
    (a device scalar) when the prompt learner is in training mode (:188-189)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int,
                 device: str | torch.device = "cuda:0", act_dtype: torch.dtype = torch.float16, max_batch: int = 1,
                 ctx: Optional[np.ndarray] = None, meta: Optional[Dict[str, np.ndarray]] = None,
                 cfg: Optional[RPOConfig] = None):
        tokens = np.asarray(tokenized_prompts, dtype=np.int64)
        if cfg is None:
            cfg = config_from_state_dict(state_dict, 1, tokens.shape[0])
        self.cfg = cfg
        self.engine = eng = Engine(cfg, state_dict, tokens, torch.device(device), act_dtype, max_batch)
        e, dt = cfg.embed, cfg.d_t
        h = e // 16                                                          # vis_dim // 16 (:94)
        with torch.cuda.device(eng.dev):
            eng.coop_setup(n_ctx, replicas=max_batch, meta_hidden=h)
            if ctx is None:                                                  # the reference's draws, in its order:
                ctx = torch.empty(n_ctx, dt).normal_(std=0.02).numpy()       # nn.init.normal_(ctx_vectors, std=0.02) (:83-84)
            if meta is None:                                                 # then nn.Linear's default initialisation
                l1, l2 = torch.nn.Linear(e, h), torch.nn.Linear(h, dt)
                meta = dict(w1=l1.weight.detach().numpy(), b1=l1.bias.detach().numpy(),
                            w2=l2.weight.detach().numpy(), b2=l2.bias.detach().numpy())
            eng.coop_ctx.copy_(torch.as_tensor(np.asarray(ctx, dtype=np.float32)))
            for t, k in zip(eng.meta, ("w1", "b1", "w2", "b2")):
                t.copy_(torch.as_tensor(np.asarray(meta[k], dtype=np.float32)))
        self.prompt_learner = CoCoOpPromptLearner(eng, n_ctx, state_dict["token_embedding.weight"], tokens)
        self.tokenized_prompts = tokens

    def __call__(self, image: torch.Tensor, label: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self.engine
        with torch.cuda.device(eng.dev):
            image = image.to(device=eng.dev, dtype=torch.float32).contiguous()
            if self.prompt_learner.training and label is not None:
                eng.cocoop_forward_backward(image, label.to(eng.dev, dtype=torch.int64))
                return eng.loss[0]
            return eng.cocoop_forward_backward(image, None)


class CoCoOp(CoOp):
    """The trainer's step (trainers/cocoop.py:255-275): loss = model(image, label) -> zero_grad -> backward -> SGD step
    on ctx and the meta-net; returns {"loss"} (no accuracy: the model returns the loss itself in training mode)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int = 4,
                 optim: Optional[OptimConfig] = None, device: str | torch.device = "cuda:0",
                 act_dtype: torch.dtype = torch.float16, batch_size: int = 1, num_batches: int = 1,
                 ctx: Optional[np.ndarray] = None, meta: Optional[Dict[str, np.ndarray]] = None,
                 cfg: Optional[RPOConfig] = None, use_graph: bool = False):
        self.optim_cfg = optim or OptimConfig(lr=0.002, max_epoch=10)    # configs/trainers/CoCoOp/vit_b16_c4_ep10_batch1.yaml
        self.model = CoCoOpCustomCLIP(state_dict, tokenized_prompts, n_ctx, device, act_dtype, batch_size, ctx, meta, cfg)
        self.engine, self.cfg = self.model.engine, self.model.cfg
        self.device = self.engine.dev
        self.batch_size, self.num_batches = batch_size, num_batches
        self.epoch = self.batch_idx = self._steps = 0
        self.lr = lr_at_epoch(self.optim_cfg, 0)
        self.use_graph = use_graph
        self._graph = None

    def _enqueue(self, image: torch.Tensor, label: torch.Tensor) -> None:
        eng, oc = self.engine, self.optim_cfg
        eng.cocoop_forward_backward(image, label)
        ops.sgd_step(eng.coop_params, eng.coop_grads, eng.coop_moms, self.lr, oc.momentum, oc.weight_decay, 1.0,
                     first_step=(self._steps == 0))

    def forward_backward(self, batch) -> Dict[str, float]:
        eng, oc = self.engine, self.optim_cfg
        with torch.cuda.device(self.device):
            image, label = self.parse_batch_train(batch)
            self.step_async(image, label)
            summary = {"loss": float(eng.loss.item())}
        if (self.batch_idx + 1) == self.num_batches:
            self.epoch += 1
            self.lr = lr_at_epoch(self.optim_cfg, self.epoch)
            self.batch_idx = 0
        else:
            self.batch_idx += 1
        return summary
