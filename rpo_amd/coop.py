"""CoOp and CoCoOp on the HIP engine: the host-side mirror of `trainers/coop.py` (PromptLearner :60-134, CustomCLIP :185-208, the
`CoOp` trainer's `forward_backward` :258-281) -- SURVEY.md section 8f rank 4, the sibling trainer that shares RPO's towers
without the read-only mask.

    prompts  = [SOS | ctx (n_ctx learned vectors, shared by all classes) | class name . EOT]      (:117-134, "end")
    logits   = exp(logit_scale) * normalise(encode_image(image)) @ normalise(encode_text(prompts)).T
    loss     = F.cross_entropy(logits, label);   only `ctx` is trained (:228-230)

Unlike RPO's prompts, a context vector is read by every later token, so its gradient needs the DENSE text-tower backward
(all tokens of all classes, causal attention with dK / dV): `Engine.coop_forward_backward`.  The image tower is plain
frozen CLIP and is only run forward.  The reference's defaults (configs/trainers/CoOp/vit_b16_ep50.yaml) are a generic
context (CSC = False) with the class token at the end; its other options are built too (round 4): class-specific contexts
(`csc=True`, :84-86), `class_token_position` "middle" / "front" (:136-183), `ctx_init` word embeddings (:72-80) and
the `amp` precision branch (:250, :263-270 -- what is left of GradScaler when gradients are fp32: a step whose gradient
holds Inf / NaN is skipped, as `RPO(amp=True)`).

CoCoOp (`trainers/cocoop.py`: PromptLearner :60-153, CustomCLIP :156-192, trainer :255-275) adds a meta-net on the
normalised image feature whose output shifts the context PER IMAGE, so every image has its own text features for every
class: `Engine.cocoop_forward_backward` runs the class set once per image (B * n_cls virtual classes) through the same
dense text path and trains ctx + the two meta-net layers (`CoCoOpCustomCLIP`, `CoCoOp` below).

The caller provides the token ids of the "X X .. name." prompts (the BPE tokenizer is out of scope).
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import numpy as np
import torch

from . import ops
from .config import RPOConfig
from .custom_clip import config_from_state_dict
from .engine import Engine, make_engine
from .trainer import OptimConfig, load_checkpoint_file, lr_at_epoch, write_checkpoint


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

    def named_parameters(self):
        """(name, device tensor) of what is trained, in the order of the engine's flat parameter buffer -- which is the
        order torch.optim.SGD numbers them in a checkpoint's optimizer state."""
        yield "ctx", self.engine.coop_ctx


def _init_ctx(cfg: RPOConfig, n_cls: int, n_ctx: int, csc: bool, ctx, ctx_init, token_embedding) -> np.ndarray:
    """trainers/coop.py:72-91: `ctx_init` = [n_words, d_t] embeddings of the initialisation words (the reference embeds
    `clip.tokenize(CTX_INIT)` and takes rows 1 .. n_ctx, :76-80; the tokenizer is out of scope here, so the caller passes
    either those rows or the words' token ids); else N(0, 0.02) from torch's global generator in the reference's shape
    -- [n_cls, n_ctx, d] for class-specific contexts (:84-86), [n_ctx, d] otherwise (:87-88)."""
    if ctx is not None:
        return np.asarray(ctx, dtype=np.float32)
    if ctx_init is not None:
        ci = np.asarray(ctx_init)
        if ci.ndim == 1:                                                     # token ids of the words
            ci = np.asarray(token_embedding)[ci.astype(np.int64)]
        assert ci.shape == (n_ctx, cfg.d_t), "ctx_init: n_ctx word embeddings (or their token ids)"
        # (with CSC the reference takes this branch too and silently trains ONE generic context, trainers/coop.py:72-80
        #  before :84; callers see it in coop_ctx's shape)
        return ci.astype(np.float32)
    shape = (n_cls, n_ctx, cfg.d_t) if csc else (n_ctx, cfg.d_t)
    return torch.empty(*shape).normal_(std=0.02).numpy()


class CoOpCustomCLIP:
    """`trainers/coop.py:CustomCLIP`: `model(image)` -> logits [B, n_cls] (fp32, on the device)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int,
                 device: str | torch.device = "cuda:0", act_dtype: torch.dtype = torch.float16, max_batch: int = 32,
                 ctx: Optional[np.ndarray] = None, cfg: Optional[RPOConfig] = None, csc: bool = False,
                 class_token_position: str = "end", ctx_init: Optional[np.ndarray] = None):
        tokens = np.asarray(tokenized_prompts, dtype=np.int64)
        if cfg is None:
            cfg = config_from_state_dict(state_dict, 1, tokens.shape[0])     # one (unused) RPO prompt row per image
        self.cfg = cfg
        self.engine = make_engine(cfg, state_dict, tokens, torch.device(device), act_dtype, max_batch)
        if ctx is None and ctx_init is not None:
            csc = False                      # CTX_INIT wins over CSC, as in the reference (trainers/coop.py:72-80 vs :84)
        with torch.cuda.device(self.engine.dev):
            self.engine.coop_setup(n_ctx, csc=csc, class_token_position=class_token_position)
            # (random: the reference's own draw from torch's global generator, so a seeded run starts from its vectors)
            ctx = _init_ctx(cfg, tokens.shape[0], n_ctx, csc, ctx, ctx_init, state_dict["token_embedding.weight"])
            self.engine.coop_ctx.copy_(torch.as_tensor(ctx).reshape(self.engine.coop_ctx.shape))
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
                 ctx: Optional[np.ndarray] = None, cfg: Optional[RPOConfig] = None, use_graph: bool = False,
                 csc: bool = False, class_token_position: str = "end", ctx_init: Optional[np.ndarray] = None,
                 amp: bool = False):
        self.optim_cfg = optim or OptimConfig(lr=0.002, max_epoch=50)       # configs/trainers/CoOp/vit_b16_ep50.yaml
        self.model = CoOpCustomCLIP(state_dict, tokenized_prompts, n_ctx, device, act_dtype, batch_size, ctx, cfg,
                                    csc=csc, class_token_position=class_token_position, ctx_init=ctx_init)
        self._init_common(batch_size, num_batches, use_graph, amp)

    def _init_common(self, batch_size: int, num_batches: int, use_graph: bool, amp: bool) -> None:
        self.engine, self.cfg = self.model.engine, self.model.cfg
        self.device = self.engine.dev
        self.batch_size, self.num_batches = batch_size, num_batches
        self.epoch = self.batch_idx = self._steps = 0
        self.lr = lr_at_epoch(self.optim_cfg, 0)
        self.use_graph = use_graph
        self._graph = None                               # (HIP graph of one step, the learning rate it was captured with)
        # PREC "amp" (trainers/coop.py:250, :263-270; trainers/cocoop.py:239, :256-263): logits, loss and gradients are
        # fp32 here, so GradScaler's scale / unscale has nothing to act on; what is left of it is skipping a step whose
        # gradient holds Inf / NaN (rpo_sgd_step_guarded: one workgroup scans the gradients, updates only if all finite)
        self.amp = amp
        self._found_inf = torch.zeros(2, dtype=torch.int32, device=self.device) if amp else None
        self.best_result = -float("inf")

    def _forward_backward(self, image: torch.Tensor, label: torch.Tensor) -> None:
        self.engine.coop_forward_backward(image, label)

    def _enqueue(self, image: torch.Tensor, label: torch.Tensor) -> None:
        eng, oc = self.engine, self.optim_cfg
        self._forward_backward(image, label)
        if self.amp:
            ops.sgd_step_guarded(eng.coop_params, eng.coop_grads, eng.coop_moms, self.lr, oc.momentum, oc.weight_decay,
                                 1.0, first_step=(self._steps == 0), found_inf=self._found_inf)
        else:
            ops.sgd_step(eng.coop_params, eng.coop_grads, eng.coop_moms, self.lr, oc.momentum, oc.weight_decay, 1.0,
                         first_step=(self._steps == 0))

    @property
    def skipped_steps(self) -> int:
        """amp: steps GradScaler would have skipped so far (a device read)."""
        return int(self._found_inf[1].item()) if self.amp else 0

    # -- checkpoints in Dassl's layout (the reader: trainers/coop.py:283-325 / trainers/cocoop.py:277-314) ---------------
    def save_model(self, directory: str, epoch: Optional[int] = None, is_best: bool = False,
                   val_result: Optional[float] = None) -> str:
        """`<directory>/prompt_learner/model.pth.tar-<epoch>` (+ `model-best.pth.tar`) with `state_dict` (ctx, the
        meta-net for CoCoOp, and the token_prefix / token_suffix buffers the reference's module registers, :100-101),
        `epoch`, `optimizer` (torch.optim.SGD's state-dict layout, parameters in `named_parameters` order)."""
        epoch = self.epoch if epoch is None else epoch
        pl = self.model.prompt_learner
        state, off, oc = {}, 0, self.optim_cfg
        names = [n for n, _ in pl.named_parameters()]
        if self._steps > 0:
            m = self.engine.coop_moms.detach().cpu()
            for i, (_, t) in enumerate(pl.named_parameters()):
                state[i] = {"momentum_buffer": m[off:off + t.numel()].reshape(t.shape).clone()}
                off += t.numel()
        group = {"lr": self.lr, "momentum": oc.momentum, "dampening": 0, "weight_decay": oc.weight_decay, "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None, "initial_lr": oc.lr,
                 "params": list(range(len(names)))}
        ck = {"state_dict": pl.state_dict(), "epoch": int(epoch), "optimizer": {"state": state, "param_groups": [group]},
              "scheduler": {"last_epoch": int(epoch)}, "val_result": val_result, "steps": int(self._steps)}
        return write_checkpoint(directory, ck, epoch, is_best)

    def load_model(self, directory: str, epoch: Optional[int] = None) -> Optional[dict]:
        """trainers/coop.py:283-325: `model-best.pth.tar` unless an epoch is named; token_prefix / token_suffix are
        dropped (:315-320: they belong to the class names the checkpoint was trained on); load_state_dict(strict=False).
        Returns the checkpoint dict it read (None where the reference skips: no directory) -- for `resume_model`; nothing
        of it is kept on the trainer."""
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return None
        model_file = "model-best.pth.tar" if epoch is None else f"model.pth.tar-{epoch}"
        model_path = os.path.join(directory, "prompt_learner", model_file)
        if not os.path.exists(model_path):
            raise FileNotFoundError(f'Model not found at "{model_path}"')
        ck = load_checkpoint_file(model_path)
        sd = dict(ck["state_dict"])
        for k in ("token_prefix", "token_suffix"):
            sd.pop(k, None)
        print(f'Loading weights to prompt_learner from "{model_path}" (epoch = {ck["epoch"]})')
        params = list(self.model.prompt_learner.named_parameters())
        with torch.no_grad():
            for name, p in params:
                if name in sd:
                    p.copy_(torch.as_tensor(sd[name]).to(p.dtype).reshape(p.shape))
        # weights only, as the reference's load_model (trainers/coop.py:283-325: load_state_dict(strict=False) and nothing
        # else): optimiser state, epoch and learning rate stay what they were -- loading a finished model-best file and
        # training on must not start at the cosine's last rate with stale momentum (advisor, round 4).  Resuming a run is
        # `resume_model` below (Dassl's resume_model_if_exist).
        self._graph = None
        return ck

    def resume_model(self, directory: str, epoch: Optional[int] = None) -> int:
        """Dassl's `resume_model_if_exist` for this trainer: `load_model` plus the optimiser's momentum buffers, the epoch
        and the learning rate of the checkpoint.  Returns the epoch to continue from.  A checkpoint whose momentum does
        not match the trained tensors (another n_ctx / CSC setting) is refused rather than half-applied."""
        ck = self.load_model(directory, epoch)
        if ck is None:
            raise ValueError("resume_model needs a checkpoint directory (load_model skipped: nothing was loaded)")
        params = list(self.model.prompt_learner.named_parameters())
        st = (ck.get("optimizer") or {}).get("state") or {}
        if st:
            bufs = [st[i]["momentum_buffer"] for i in range(len(params))]
            flat = torch.cat([torch.as_tensor(b).reshape(-1).float() for b in bufs])
            if flat.numel() != self.engine.coop_moms.numel():
                raise ValueError(f"checkpoint momentum has {flat.numel()} elements, the trainer {self.engine.coop_moms.numel()}: "
                                 "it was written with other context settings")
            self.engine.coop_moms.copy_(flat)
            self._steps = max(1, int(ck.get("steps", 1)))
        self.epoch = int(ck.get("epoch", 0))
        self.lr = lr_at_epoch(self.optim_cfg, self.epoch)
        self._graph = None
        return self.epoch

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

    def named_parameters(self):
        yield "ctx", self.engine.coop_ctx
        for name, t in zip(("meta_net.linear1.weight", "meta_net.linear1.bias", "meta_net.linear2.weight",
                            "meta_net.linear2.bias"), self.engine.meta):
            yield name, t

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
                 cfg: Optional[RPOConfig] = None, ctx_init: Optional[np.ndarray] = None):
        tokens = np.asarray(tokenized_prompts, dtype=np.int64)
        if cfg is None:
            cfg = config_from_state_dict(state_dict, 1, tokens.shape[0])
        self.cfg = cfg
        self.engine = eng = make_engine(cfg, state_dict, tokens, torch.device(device), act_dtype, max_batch)
        e, dt = cfg.embed, cfg.d_t
        h = e // 16                                                          # vis_dim // 16 (:94)
        with torch.cuda.device(eng.dev):
            eng.coop_setup(n_ctx, replicas=max_batch, meta_hidden=h)
            # the reference's draws, in its order: nn.init.normal_(ctx_vectors, std=0.02) (:83-84) or the CTX_INIT words
            # (:72-80, configs/trainers/CoCoOp/vit_b16_c4_ep10_batch1_ctxv1.yaml: "a photo of a") ...
            ctx = _init_ctx(cfg, tokens.shape[0], n_ctx, False, ctx, ctx_init, state_dict["token_embedding.weight"])
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
            # inference: the reference tests at batch 100 (configs/trainers/CoCoOp/*.yaml TEST.BATCH_SIZE) with a model
            # built for training at batch 1 -- every image has its own prompts, so a batch is walked in chunks of the
            # `replicas` the engine was set up with
            R, B = eng.coop_replicas, image.shape[0]
            if B <= R:
                return eng.cocoop_forward_backward(image, None)
            out = torch.empty(B, self.cfg.n_cls, dtype=torch.float32, device=eng.dev)
            for b0 in range(0, B, R):
                out[b0:b0 + R].copy_(eng.cocoop_forward_backward(image[b0:b0 + R], None))
            return out


class CoCoOp(CoOp):
    """The trainer's step (trainers/cocoop.py:255-275): loss = model(image, label) -> zero_grad -> backward -> SGD step
    on ctx and the meta-net; returns {"loss"} (no accuracy: the model returns the loss itself in training mode)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], tokenized_prompts: np.ndarray, n_ctx: int = 4,
                 optim: Optional[OptimConfig] = None, device: str | torch.device = "cuda:0",
                 act_dtype: torch.dtype = torch.float16, batch_size: int = 1, num_batches: int = 1,
                 ctx: Optional[np.ndarray] = None, meta: Optional[Dict[str, np.ndarray]] = None,
                 cfg: Optional[RPOConfig] = None, use_graph: bool = False, ctx_init: Optional[np.ndarray] = None,
                 amp: bool = False):
        self.optim_cfg = optim or OptimConfig(lr=0.002, max_epoch=10)    # configs/trainers/CoCoOp/vit_b16_c4_ep10_batch1.yaml
        self.model = CoCoOpCustomCLIP(state_dict, tokenized_prompts, n_ctx, device, act_dtype, batch_size, ctx, meta, cfg,
                                      ctx_init=ctx_init)
        self._init_common(batch_size, num_batches, use_graph, amp)

    def _forward_backward(self, image: torch.Tensor, label: torch.Tensor) -> None:
        self.engine.cocoop_forward_backward(image, label)

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
