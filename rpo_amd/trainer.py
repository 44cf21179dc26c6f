"""``RPO`` trainer step with the reference's surface (trainers/rpo.py:235-357).

Only the step semantics are reproduced -- Dassl's loop/data manager are out of
scope: ``forward_backward(batch)`` runs
forward -> zero-grad -> backward -> SGD step (:306-309), returns
``{"loss": float}`` (:311) and updates the learning rate after the last batch
of an epoch (:313-314).  The whole forward+backward is one captured HIP graph
replay; the optimiser is the library's fused SGD kernel; in data-parallel runs a
single RCCL all-reduce of the flat gradient buffer sits between the two.
"""
from __future__ import annotations

import collections
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .config import RPOConfig
from .custom_clip import CustomCLIP
from .dist import GradSync
from ._lib import xenv as _xenv


@dataclass
class OptimConfig:
    """configs/trainers/RPO/main_K24.yaml:15-22.  momentum / weight_decay are Dassl's
    defaults (un-vendored; 0.9 / 5e-4 upstream) and therefore explicit here."""
    lr: float = 0.01
    max_epoch: int = 15
    lr_scheduler: str = "cosine"
    warmup_epoch: int = 1
    warmup_type: str = "constant"
    warmup_cons_lr: float = 1e-5
    momentum: float = 0.9
    weight_decay: float = 5e-4


def lr_at_epoch(oc: OptimConfig, epoch: int) -> float:
    """LR in force during 0-based ``epoch`` under Dassl's ConstantWarmupScheduler wrapping
    CosineAnnealingLR(T_max=max_epoch) (`build_lr_scheduler`, trainers/rpo.py:275): the warm-up wrapper does NOT
    step its successor while it warms up, so the cosine starts at its own epoch 0 when the warm-up ends --
    epoch e >= warmup runs at cos(pi * (e - warmup) / T), i.e. epoch 1 of the yaml's schedule runs at the full
    0.01 (pinned against a re-creation driven by torch's CosineAnnealingLR in tests/test_host_logic.py)."""
    warm = oc.warmup_epoch if oc.warmup_type == "constant" else 0
    if epoch < warm:
        return oc.warmup_cons_lr
    if oc.lr_scheduler == "cosine":
        return 0.5 * oc.lr * (1.0 + math.cos(math.pi * (epoch - warm) / oc.max_epoch))
    return oc.lr


class RPO:
    def __init__(self, cfg: RPOConfig, state_dict: Dict[str, np.ndarray], tokens: Optional[np.ndarray] = None,
                 optim: Optional[OptimConfig] = None, device: str | torch.device = "cuda:0",
                 act_dtype: torch.dtype = torch.bfloat16, batch_size: int = 4, num_batches: int = 1,
                 use_graph: bool = True, sync: Optional[GradSync] = None, prompts=None, amp: bool = False):
        self.cfg = cfg
        # PREC "amp" (trainers/rpo.py:298-304): what is left of GradScaler when gradients are fp32 -- a step whose
        # gradient holds Inf / NaN is skipped on every rank (the flag is computed after the all-reduce)
        self.amp = amp
        self.optim_cfg = optim or OptimConfig()
        # default: joins (or creates) the process group when launched under torchrun; a lone process stays local.
        # Built BEFORE the device is resolved: under a launcher it makes this rank's GPU the current device, and an
        # index-less "cuda" then means that GPU on every rank (not cuda:0 eight times).
        self.sync = sync or GradSync()
        self.device = torch.device(device)
        if self.device.index is None:
            idx = self.sync.local_rank if (self.sync.enabled and torch.cuda.device_count() > self.sync.local_rank) \
                else torch.cuda.current_device()
            self.device = torch.device("cuda", idx)
        self.batch_size = batch_size
        self.num_batches = num_batches
        self.use_graph = use_graph
        self.epoch = 0
        self.batch_idx = 0
        self._steps = 0
        self._graph = None
        self._joint_bwd = False
        self._split_collective = False
        self._graph_collectives = False
        self._graph_collectives_note = None          # why the collectives are NOT in the graphs, when they are not
        self._text_ar_in_graph = False
        self._tail_graphs = {}
        self.best_result = -float("inf")
        self._found_inf = None                       # amp: int32[2] on the device (this step's flag, skipped steps)
        # trainers/rpo.py:287-288 turns on autograd anomaly detection ("nan detector"); there is no autograd graph
        # here, so the counterpart is a scan of loss + prompt gradients after the step (RPO_DETECT_ANOMALY=1 or
        # detect_anomaly=True; costs one D2H sync per step, so it is off in timed runs)
        self.detect_anomaly = os.environ.get("RPO_DETECT_ANOMALY") == "1"
        with torch.cuda.device(self.device):
            self.build_model(state_dict, tokens, act_dtype, prompts)

    # trainers/rpo.py:240-288
    def build_model(self, state_dict, tokens, act_dtype, prompts) -> None:
        self.model = CustomCLIP(self.cfg, state_dict, tokens, self.device, act_dtype,
                                max_batch=self.batch_size, prompts=prompts)
        self.engine = self.model.engine
        self.lr = lr_at_epoch(self.optim_cfg, 0)
        cfg = self.cfg
        self._image = torch.zeros(self.batch_size, 3, cfg.image_size, cfg.image_size, device=self.device)
        self._label = torch.zeros(self.batch_size, dtype=torch.int64, device=self.device)
        if self.sync.enabled:                       # identical prompts on every rank
            self.sync.broadcast(self.engine.params)

    def transform(self, is_train: bool):
        """Device-side counterpart of Dassl's `build_transform(cfg, is_train)` (rpo_amd/input_pipeline.py),
        built on first use."""
        from .input_pipeline import InputConfig, build_transform
        key = "_tf_train" if is_train else "_tf_test"
        if getattr(self, key, None) is None:
            icfg = InputConfig(SIZE=(self.cfg.image_size, self.cfg.image_size))
            setattr(self, key, build_transform(icfg, is_train, self.device, self.batch_size))
        return getattr(self, key)

    def parse_batch_train(self, batch):
        """trainers/rpo.py:318-323.  `batch["img"]` is either the float tensor the reference's DataLoader yields
        (transforms already applied) or a list of decoded uint8 [H, W, 3] images, in which case
        random_resized_crop + random_flip + normalize run on the device straight into the step's input buffer."""
        img = batch["img"]
        if isinstance(img, (list, tuple)):
            img = self.transform(True)(img, out=self._image if len(img) == self.batch_size else None)
        else:
            img = img.to(self.device, dtype=torch.float32, non_blocking=True)
        label = torch.as_tensor(batch["label"])
        if not label.is_cuda:
            # F.cross_entropy (trainers/rpo.py:230) raises on an out-of-range target; the head kernel would read out
            # of bounds instead, so validate where it is free (labels arrive on the host)
            lo, hi = int(label.min()), int(label.max())
            if lo < 0 or hi >= self.cfg.n_cls:
                raise IndexError(f"Target {hi if hi >= self.cfg.n_cls else lo} is out of bounds "
                                 f"(n_cls = {self.cfg.n_cls}; labels must be renumbered after the base/new split)")
        return img, label.to(self.device, dtype=torch.int64, non_blocking=True)

    def _capture(self) -> None:
        """Capture the step as FIVE HIP graphs on two streams instead of one graph with parallel
        branches: ROCm's graph executor started the image branch only after ~1.2 ms of the (latency-
        bound, tiny) text branch (rocprof timeline, profiles/), so the fork/join is done with stream
        events between graph launches -- text fwd | image fwd  ->  head  ->  text bwd | image bwd."""
        eng, B = self.engine, self.batch_size
        eng.forward_backward(self._image, self._label)          # eager warm-up: sets kernel attributes,
        torch.cuda.synchronize()                                # builds the text K/V cache
        # N > 1 without Python in the loop: the two all-reduces and the SGD launch are captured too (RCCL collectives are
        # capturable once the communicator exists -- one eager all-reduce here creates it), so a step is six graph replays
        # and a handful of event calls per rank.  Falls back to eager collectives if a capture fails;
        # RPO_NO_GRAPH_COLLECTIVES=1 keeps them eager.
        self._graph_collectives = (self.sync.enabled and self.sync.backend == "nccl"
                                   and os.environ.get("RPO_NO_GRAPH_COLLECTIVES") != "1")
        if self._graph_collectives:
            self.sync.all_reduce_sum(eng.grads)                 # (warm-up gradients: the first real step overwrites them)
            torch.cuda.synchronize()
        self._tail_graphs = {}

        def cap(fn):
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a data-parallel run may query events while we capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # replays on the stream current at replay()
                fn()
            return g

        self._g_text_fwd = cap(lambda: eng._text_forward(train=True))
        self._g_img_fwd = cap(lambda: eng._image_forward(self._image, train=True))
        self._g_head = cap(lambda: eng.head(B, self._label))
        # both backward chains as one chain of paired launches where the kernels allow it (Engine._joint_backward)
        self._joint_bwd = eng.joint_backward_ok(B)
        self._bwd_parts = 1
        # (asynchronous with RCCL only: gloo on device tensors blocks the host until the text backward and the collective
        #  are done, and the image backward would be launched behind it -- advisor, round 4)
        self._split_collective = (self.sync.enabled and os.environ.get("RPO_ONE_COLLECTIVE") != "1"
                                  and self.sync.backend == "nccl")
        if self._joint_bwd:
            self._g_bwd = cap(lambda: eng._joint_backward(B))
        else:
            self._text_ar_in_graph = False
            if self._split_collective and self._graph_collectives:
                try:                                # text backward + the all-reduce of its gradient as one graph
                    self._g_text_bwd = cap(lambda: (eng._text_backward(), self.sync.all_reduce_sum(eng.g_text_flat)))
                    self._text_ar_in_graph = True
                except Exception as ex:             # noqa: BLE001 -- any capture failure: eager collectives
                    # the fallback is for the PROCESS GROUP (this trainer's communicator), not for this one call: from here on
                    # every collective of every step is issued eagerly between the graphs, and the bench line says why
                    print(f"[rpo_amd] capturing the collective failed ({type(ex).__name__}: {ex}); collectives stay eager")
                    self._graph_collectives = False
                    self._graph_collectives_note = f"capture of the RCCL all-reduce failed ({type(ex).__name__}): eager for this process group"
                    torch.cuda.synchronize()
            if not self._text_ar_in_graph:
                self._g_text_bwd = cap(eng._text_backward)
            # RPO_BWD_PARTS = P > 1: the image tower's prompt-row chain as P independent chains over B / P images each, on
            # P streams (Engine._image_backward_rows: the rows of different images never meet before the batch sum)
            P = int(_xenv("RPO_BWD_PARTS", "1"))
            self._bwd_parts = P if (P > 1 and B % P == 0) else 1
            if self._bwd_parts > 1:
                per = B // self._bwd_parts
                self._g_img_bwd_parts = [cap(lambda i=i: eng._image_backward_rows(B, i * per, (i + 1) * per))
                                         for i in range(self._bwd_parts)]
                self._g_img_bwd_fin = cap(lambda: eng._image_backward_finish(B))
                self._part_streams = [torch.cuda.Stream(device=self.device) for _ in range(self._bwd_parts - 1)]
                self._ev_parts = [torch.cuda.Event() for _ in range(self._bwd_parts - 1)]
            else:
                self._g_img_bwd = cap(lambda: eng._image_backward(B))
        self._ev_fork = torch.cuda.Event()
        self._ev_text_fwd = torch.cuda.Event()
        self._ev_head = torch.cuda.Event()
        self._ev_text_bwd = torch.cuda.Event()
        # EARLY PATCH EMBED (round 6): im2col + the patch GEMM of the NEXT batch depend on nothing this step computes
        # (trainers/rpo.py:198-202: no prompt before :204), so when the caller names the next batch
        # (step_async(next_image=...)) they run on the side stream behind the text backward -- under the tail of this
        # step's image backward -- and the next step's image forward starts at img_embed_norm.  Captured on first use.
        self._g_patch = self._g_img_fwd_np = None
        self._ev_patch = torch.cuda.Event()
        self._patch_tag = None                       # (data_ptr, version) of the batch whose patch rows are in x_pre
        self._image_next = None
        # EARLY TEXT (round 5): the text tower's gradient is complete ~0.25 ms before the image tower's, and the next step's
        # text forward needs nothing but the updated text prompts -- so the text half of the SGD step runs on the side
        # stream right behind the text backward (and its all-reduce), and the NEXT step's text forward follows it there,
        # under the tail of this step's image backward instead of beside the next image forward.  Every step still runs
        # one text forward; `_text_fwd_for` says which prompt version the features on the device belong to, so anything
        # that changes the prompts from outside (load_model, set_prompts) simply makes the next step compute them again.
        # Not with amp (one found-inf verdict covers both halves), the paired-launch experiments or a whole-buffer
        # collective.  Measured same-box (profiles/r05_ab_early_text.txt): batch 4 -1.0 %, batch 8 -0.4 %, batch 32 +0.7 % --
        # at 32 images the text forward costs the image BACKWARD (the step's critical chain of small launches) more than it
        # cost the image forward -- so it is on below 2048 image token rows per step only.  RPO_EARLY_TEXT=1 / =0 force it.
        want = os.environ.get("RPO_EARLY_TEXT")
        small = self.batch_size * (self.cfg.n_frozen + self.cfg.K) < 2048
        self._early_text = ((want == "1" or (want is None and small)) and not self.amp and not self._joint_bwd
                            and self._bwd_parts == 1 and (not self.sync.enabled or self._split_collective))
        self._text_fwd_for = -1
        # ONE GRAPH PER STEP (round 6, RPO_ONE_GRAPH=1): text fwd || image fwd -> head -> text bwd (+ its all-reduce) ||
        # image bwd captured as ONE graph whose two branches fork and join through the side stream inside the capture,
        # instead of five replays with host-enqueued events between them.  Not with early text / early patch embed /
        # the paired or multi-part backward experiments (they reorder work ACROSS steps or streams).
        self._g_step = None
        if (os.environ.get("RPO_ONE_GRAPH") == "1" and not self._early_text and not self._joint_bwd and self._bwd_parts == 1
                and (not self._split_collective or self._text_ar_in_graph or not self.sync.enabled)):
            def whole():
                main, side = torch.cuda.current_stream(), eng.side
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    eng._text_forward(train=True)
                eng._image_forward(self._image, train=True)
                main.wait_stream(side)
                eng.head(B, self._label)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    eng._text_backward()
                    if self._text_ar_in_graph:
                        self.sync.all_reduce_sum(eng.g_text_flat)
                eng._image_backward(B)
                main.wait_stream(side)
            try:
                self._g_step = cap(whole)
            except Exception as ex:                 # noqa: BLE001 -- a capture failure keeps the five-graph path
                print(f"[rpo_amd] one-graph capture failed ({type(ex).__name__}: {ex}); five graphs per step")
                self._g_step = None
                torch.cuda.synchronize()
        self._graph = True
        torch.cuda.synchronize()

    def _patch_graph(self, next_image: torch.Tensor):
        """The early patch embed of `next_image` as a captured graph that reads the caller's buffer IN PLACE (one graph per
        distinct buffer address, up to eight: a loader's ring of device buffers); beyond that the batch is copied into a
        staging buffer with a graph of its own.  Also captures, once, the image forward that starts behind the patch rows."""
        eng = self.engine
        if self._g_img_fwd_np is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                eng._image_forward(self._image, train=True, patch_done=True)
            self._g_img_fwd_np = g
            self._g_patch = {}
        src = next_image
        if src.data_ptr() not in self._g_patch and len(self._g_patch) >= 8:
            if self._image_next is None:
                self._image_next = torch.zeros_like(self._image)
            # (the previous step's patch embed may still be READING the staging buffer on the side stream -- at small batches,
            #  with the early text forward in front of it, it outlives the main stream's backward: the copy goes behind it)
            torch.cuda.current_stream().wait_event(self._ev_patch)
            self._image_next.copy_(next_image, non_blocking=True)
            src = self._image_next
        g = self._g_patch.get(src.data_ptr())
        if g is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                eng._patch_embed_early(src)
            self._g_patch[src.data_ptr()] = (g, src)          # (keeps the buffer alive as long as its graph)
            torch.cuda.synchronize()
        else:
            g = g[0]
        return g

    def _replay(self, patch_ready: bool = False, g_patch=None) -> None:
        main, side = torch.cuda.current_stream(), self.engine.side
        if self._g_step is not None and not patch_ready and g_patch is None:
            self._g_step.replay()
            return
        self._ev_fork.record(main)                  # inputs / updated prompts are ready
        if not (self._early_text and self._text_fwd_for == self.engine.params_version):
            side.wait_event(self._ev_fork)          # (early text: the previous step already ran this forward, see below)
            with torch.cuda.stream(side):
                self._g_text_fwd.replay()
                self._ev_text_fwd.record(side)
        if patch_ready:                             # the previous step ran this batch's patch embed on the side stream
            main.wait_event(self._ev_patch)
            self._g_img_fwd_np.replay()
        else:
            self._g_img_fwd.replay()
        main.wait_event(self._ev_text_fwd)
        self._g_head.replay()
        if self._joint_bwd:
            self._g_bwd.replay()
            return
        self._ev_head.record(main)
        side.wait_event(self._ev_head)
        with torch.cuda.stream(side):
            self._g_text_bwd.replay()
            # N > 1: the text tower's gradient is complete here, ~0.15-0.3 ms before the image tower's: its all-reduce
            # goes out now, behind the text chain on the side stream, and travels under the image backward
            # (RPO_ONE_COLLECTIVE=1: one all-reduce of the whole buffer after the join, as before)
            if self._split_collective and not self._text_ar_in_graph:
                self.sync.all_reduce_sum(self.engine.g_text_flat)
            if self._early_text:
                self._run_tail("text")              # SGD on the text prompts, then the NEXT step's text forward
            self._ev_text_bwd.record(side)
            if g_patch is not None:
                # the NEXT batch's patch rows, behind the text backward and IN FRONT of the early text forward: the next image
                # forward waits for them, and behind a 0.45 ms text forward that wait serialised the two towers (first
                # version: batch 4 1.82 vs 1.47 ms, profiles/r06_bench_batchsweep.json of that tree)
                g_patch.replay()
                self._ev_patch.record(side)
            if self._early_text:
                self._g_text_fwd.replay()
                self._ev_text_fwd.record(side)
                self._text_fwd_for = self.engine.params_version + 1     # (step_async bumps the version after the tail)
        if self._bwd_parts > 1:
            for st, ev, g in zip(self._part_streams, self._ev_parts, self._g_img_bwd_parts[1:]):
                st.wait_event(self._ev_head)
                with torch.cuda.stream(st):
                    g.replay()
                    ev.record(st)
            self._g_img_bwd_parts[0].replay()
            for ev in self._ev_parts:
                main.wait_event(ev)
            self._g_img_bwd_fin.replay()
        else:
            self._g_img_bwd.replay()
        main.wait_event(self._ev_text_bwd)

    def forward_backward(self, batch) -> Dict[str, float]:
        """trainers/rpo.py:290-316."""
        with torch.cuda.device(self.device):
            image, label = self.parse_batch_train(batch)
            loss = self.step_async(image, label)
            summary = {"loss": float(loss.item())}              # D2H sync, as the reference (:311)
            if self.detect_anomaly:
                self.check_finite()
        if (self.batch_idx + 1) == self.num_batches:
            self.update_lr()
            self.batch_idx = 0
        else:
            self.batch_idx += 1
        return summary

    def step_async(self, image: torch.Tensor, label: torch.Tensor, next_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step, nothing synchronised; returns the device loss scalar.  The caller's current
        device must be this trainer's (kernels launch on the current device's streams).
        `next_image` (optional, graph path): the batch the NEXT call will pass as `image`, already on the device -- its
        patch embedding then runs under this step's backward (see _capture).  Contract: the next call's `image` is that
        same tensor, unmodified (checked by address + version counter; a mismatch just recomputes)."""
        eng, oc = self.engine, self.optim_cfg
        assert torch.cuda.current_device() == self.device.index, "set the trainer's device current (torch.cuda.set_device)"
        assert image.shape[0] == self.batch_size, "graph path needs the configured batch size"
        tag = (image.data_ptr(), image._version)
        pending = self.use_graph and self._graph is not None and self._patch_tag is not None
        patch_ready = pending and self._patch_tag == tag
        if pending and not patch_ready:             # a promise that was not kept: this step's own patch embed goes behind it
            torch.cuda.current_stream().wait_event(self._ev_patch)
        self._patch_tag = None
        if not patch_ready and image.data_ptr() != self._image.data_ptr():
            self._image.copy_(image, non_blocking=True)
        self._label.copy_(label, non_blocking=True)
        if self.use_graph:
            if self._graph is None:
                self._capture()
            patch_next = (next_image is not None and os.environ.get("RPO_EARLY_PATCH", "1") != "0"
                          and not self._joint_bwd and self._bwd_parts == 1)
            g_patch = None
            if patch_next:
                assert (next_image.shape == self._image.shape and next_image.dtype == torch.float32 and next_image.is_cuda
                        and next_image.is_contiguous())
                g_patch = self._patch_graph(next_image)
                self._patch_tag = (next_image.data_ptr(), next_image._version)
            self._replay(patch_ready, g_patch)
        else:
            eng.forward_backward(self._image, self._label)
        if self.amp and self._found_inf is None:
            self._found_inf = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._run_tail("img" if (self.use_graph and getattr(self, "_early_text", False)) else "all")
        self._steps += 1
        eng.params_version += 1
        eng.text_f_version = -1
        return eng.loss

    def _run_tail(self, which: str) -> None:
        """The step's tail -- `which` = "all", or its "text" / "img" half (early text) -- on the current stream: as a captured
        graph per learning rate when the collectives are captured (N > 1 on RCCL), else as plain launches."""
        tail = None
        if self.use_graph and getattr(self, "_graph_collectives", False) and self._steps > 0:
            key = (self.lr, which)                              # the rate is a kernel argument: one small graph per rate
            tail = self._tail_graphs.get(key)
            if tail is None:
                try:
                    tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(tail, capture_error_mode="thread_local"):
                        self._tail(which)
                    if len(self._tail_graphs) >= 64:   # a schedule that changes the rate every step: keep the cache bounded
                        self._tail_graphs.clear()
                    self._tail_graphs[key] = tail
                except Exception as ex:             # noqa: BLE001
                    print(f"[rpo_amd] capturing the step's tail failed ({type(ex).__name__}: {ex}); it stays eager")
                    self._graph_collectives_note = f"capture of the step's tail failed ({type(ex).__name__}): eager for this process group"
                    self._graph_collectives, tail = False, None
                    torch.cuda.synchronize()
        if tail is not None:
            tail.replay()
        else:
            self._tail(which)

    def _tail(self, which: str = "all") -> None:
        """What follows the backward chains: the all-reduce of what is still local, then the fused SGD step -- over the
        whole flat buffer [text | img], or over one half of it (early text: the text half on the side stream behind the
        text backward, whose all-reduce _replay has already issued; the image half on the main stream)."""
        eng, oc = self.engine, self.optim_cfg
        nt = eng.g_text_flat.numel()
        if which == "text":
            sl = slice(0, nt)
        elif which == "img":
            sl = slice(nt, None)
            self.sync.all_reduce_sum(eng.g_img_flat)
        else:
            sl = slice(None)
            if self.use_graph and self._split_collective and not self._joint_bwd:
                self.sync.all_reduce_sum(eng.g_img_flat)        # (g_text went out behind the text backward, _replay)
            else:
                self.sync.all_reduce_sum(eng.grads)
        if self.amp:
            ops.sgd_step_guarded(eng.params, eng.grads, eng.mom, self.lr, oc.momentum, oc.weight_decay,
                                 self.sync.grad_scale, first_step=(self._steps == 0), found_inf=self._found_inf)
        else:
            ops.sgd_step(eng.params[sl], eng.grads[sl], eng.mom[sl], self.lr, oc.momentum, oc.weight_decay,
                         self.sync.grad_scale, first_step=(self._steps == 0))

    def update_lr(self) -> None:
        self.epoch += 1
        self.lr = lr_at_epoch(self.optim_cfg, self.epoch)

    def check_finite(self) -> None:
        """NaN / Inf scan of the step's outputs (the reference's `set_detect_anomaly(True)`, trainers/rpo.py:288)."""
        eng = self.engine
        self._join_side()
        bad = [n for n, t in (("loss", eng.loss), ("logits", eng.logits[:self.batch_size]), ("grad text_prompt", eng.g_text),
                              ("grad img_prompt", eng.g_img), ("prompts", eng.params)) if not bool(torch.isfinite(t).all())]
        if bad:
            raise FloatingPointError(f"non-finite values after step {self._steps}: {', '.join(bad)}")
        # experiments only (RPO_EXPERIMENTAL=1): a bounded spin of a persistent launch that gave up leaves results
        # undefined, not non-finite -- its give-up words are part of the scan (advisor, round 4)
        for name, t, idx in (("rpo_chain_bwd (image)", getattr(eng, "chain_state_v", None), 0),
                             ("rpo_chain_bwd (text)", getattr(eng, "chain_state_t", None), 0),
                             ("rpo_mlp_fused", getattr(eng, "mlp_counters", None) if _xenv("RPO_MLP_FUSED") != "0" else None, -1)):
            if t is not None and int(t[idx]) != 0:
                raise RuntimeError(f"{name}: a bounded spin gave up in or before step {self._steps}: the results are undefined")

    # -- evaluation (trainers/rpo.py:229-232 eval branch) -----------------------------------
    def _join_side(self) -> None:
        """Early text / early patch embed: the next step's text forward or patch embed may still be running on the side
        stream; whatever reads or writes the prompts, the text tower's buffers or the image tower's input buffers from
        the current stream goes behind it (and a pending patch embed is forgotten: the next step computes its own)."""
        if not self._graph:
            return
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_early_text", False):
            cur.wait_event(self._ev_text_fwd)
        if getattr(self, "_patch_tag", None) is not None:
            cur.wait_event(self._ev_patch)
            self._patch_tag = None

    @torch.no_grad()
    def model_inference(self, image) -> torch.Tensor:
        self._join_side()
        with torch.cuda.device(self.device):
            if isinstance(image, (list, tuple)):                # decoded uint8 images: resize + center crop + normalize
                image = self.transform(False)(image)
            self.model.prompt_learner.eval()
            try:
                return self.model(image)
            finally:
                self.model.prompt_learner.train()

    # -- checkpoints: Dassl layout <dir>/prompt_learner/{model.pth.tar-<epoch>, model-best.pth.tar} ----------
    def save_model(self, directory: str, epoch: Optional[int] = None, is_best: bool = False,
                   val_result: Optional[float] = None) -> str:
        """What Dassl's `TrainerBase.save_model` -> `save_checkpoint` leaves on disk and the reference's reader
        (trainers/rpo.py:325-357) consumes: `state_dict` (the two prompt tensors), `epoch`, `optimizer`,
        `scheduler`, `val_result`; `is_best` also writes `model-best.pth.tar`, the file `load_model` opens by
        default (:333).  Interoperable with the reference: `state_dict`, `epoch` and `optimizer` (torch.optim.SGD's own
        state-dict layout).  NOT interoperable: `scheduler` -- Dassl's `resume_from_checkpoint` would
        `load_state_dict` it into its warm-up wrapper, which only restores the wrapper's `last_epoch` (the successor
        cosine restarts), so only {"last_epoch"} is written and a reference run should rebuild its scheduler from
        `epoch`.  Dassl writes `model-best.pth.tar` alone when a validation result improves; `after_epoch_eval` here
        also keeps the numbered file it is a copy of."""
        epoch = self.epoch if epoch is None else epoch
        self._join_side()
        ck = checkpoint_dict(self.model.prompt_learner.state_dict(), epoch, self.engine.mom, self.optim_cfg,
                             self.lr, self._steps, self.cfg.K * self.cfg.d_t, val_result)
        return write_checkpoint(directory, ck, epoch, is_best)

    def after_epoch_eval(self, directory: str, val_result: float) -> bool:
        """Dassl's `after_epoch` bookkeeping for `model-best`: keep the checkpoint with the best validation result."""
        is_best = val_result > self.best_result
        if is_best:
            self.best_result = val_result
            self.save_model(directory, is_best=True, val_result=val_result)
        return is_best

    def load_model(self, directory: str, epoch: Optional[int] = None) -> None:
        """trainers/rpo.py:325-357."""
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        model_file = "model-best.pth.tar" if epoch is None else f"model.pth.tar-{epoch}"
        model_path = os.path.join(directory, "prompt_learner", model_file)
        if not os.path.exists(model_path):
            raise FileNotFoundError(f'Model not found at "{model_path}"')
        ck = load_checkpoint_file(model_path)
        sd = dict(ck["state_dict"])
        for k in ("token_prefix", "token_suffix"):              # :348-352
            sd.pop(k, None)
        print(f'Loading weights to prompt_learner from "{model_path}" (epoch = {ck["epoch"]})')
        self._join_side()
        with torch.no_grad():                                   # load_state_dict(strict=False), :357
            for name, p in self.model.prompt_learner.named_parameters():
                if name in sd:
                    p.copy_(sd[name].to(p.dtype))
        mom = _momentum_from_optimizer_state(ck.get("optimizer"), self.engine.mom.numel())
        if mom is not None:
            self.engine.mom.copy_(mom)
            self._steps = max(1, int(ck.get("steps", 1)))
        self.engine.params_version += 1
        self.epoch = int(ck.get("epoch", 0))
        self.lr = lr_at_epoch(self.optim_cfg, self.epoch)


# Names a checkpoint written by the reference's own run may reference (module, qualified name).  Dassl's
# `save_checkpoint` pickles `scheduler.state_dict()`; with the yaml's WARMUP_EPOCH = 1 that is
# ConstantWarmupScheduler's dict, whose `successor` entry IS a CosineAnnealingLR object holding the SGD optimiser
# (defaultdict state, Parameters), so `weights_only=True` rejects every real reference checkpoint.
_CKPT_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "Counter"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
}


class _Inert:
    """Stand-in for the optimiser / scheduler OBJECTS met while unpickling a reference checkpoint (the un-vendored
    dassl.optim.lr_scheduler.* wrappers, torch.optim.lr_scheduler.* successors and the torch.optim.* optimiser they
    hold): takes any constructor arguments and any state, does nothing.  Only `state_dict`, `epoch` and `optimizer`
    (a plain dict) of a checkpoint are read, so nothing needs these objects alive."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


import pickle as _pickle


class _RestrictedUnpickler(_pickle.Unpickler):
    """Resolves ONLY: the exact (module, name) pairs of _CKPT_ALLOWED; torch dtypes and typed-storage classes; and maps
    everything under `dassl` and `torch.optim` (optimisers, schedulers, and whatever those modules re-export, e.g.
    functools.partial) to the inert stand-in.  Dotted names are refused outright: pickle protocol 4 resolves
    `a.b` by a getattr chain, which would reach `types.FunctionType` through any allowed module that imports `types`
    (advisor finding, round 3)."""

    def find_class(self, module, name):
        if "." in name or not name:
            raise _pickle.UnpicklingError(f"{module}.{name} is not allowed in a checkpoint (dotted name)")
        top = module.split(".")[0]
        if top == "dassl" or module == "torch.optim" or module.startswith("torch.optim."):
            return _Inert
        key_mod = "builtins" if module == "__builtin__" else module        # protocol-2 spelling
        ok = (key_mod, name) in _CKPT_ALLOWED
        if not ok and module == "torch":
            obj = getattr(torch, name, None)
            # dtypes and the legacy typed-storage classes torch.save names in persistent ids
            ok = isinstance(obj, torch.dtype) or (isinstance(obj, type) and name.endswith("Storage"))
        if not ok:
            raise _pickle.UnpicklingError(f"{module}.{name} is not allowed in a checkpoint")
        return super().find_class(module, name)


class _RestrictedPickle:                           # the `pickle_module` protocol torch.load expects
    __name__ = "rpo_amd_restricted_pickle"
    Unpickler = _RestrictedUnpickler
    load = staticmethod(lambda f, **k: _RestrictedUnpickler(f, **k).load())


def load_checkpoint_file(path: str) -> dict:
    """torch.load restricted to what a checkpoint needs.  First `weights_only=True` (tensors, numbers, plain containers:
    every file this package writes).  A file written by the reference's own Dassl run holds the cosine scheduler object
    and through it the optimiser (see _CKPT_ALLOWED), which torch's weights-only unpickler cannot rebuild even when
    allow-listed (it refuses SETITEMS on the optimiser's defaultdict state); such files go through
    _RestrictedUnpickler: an exact allow-list of data types, optimiser / scheduler classes replaced by an inert
    stand-in, everything else -- and every dotted name -- refused.  Only `state_dict`, `epoch`, `optimizer` are read
    afterwards.  RPO_TRUST_CHECKPOINT=1: full unpickle for files holding anything else."""
    pickle = _pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as err:
        if os.environ.get("RPO_TRUST_CHECKPOINT") == "1":
            return torch.load(path, map_location="cpu", weights_only=False)
        first = err
    try:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_RestrictedPickle)
    except pickle.UnpicklingError as err:
        raise pickle.UnpicklingError(
            f"{path}: {err} (weights-only attempt: {str(first).splitlines()[-3] if str(first).count(chr(10)) > 2 else first}); "
            "set RPO_TRUST_CHECKPOINT=1 to unpickle it fully if you trust the file") from err


def checkpoint_dict(prompt_state, epoch: int, momentum: Optional[torch.Tensor], oc: OptimConfig, lr: float,
                    steps: int, n_text: int, val_result: Optional[float] = None) -> dict:
    """The dict Dassl's `save_checkpoint` pickles (keys `state_dict`, `epoch`, `optimizer`, `scheduler`,
    `val_result`); `optimizer` in torch.optim.SGD.state_dict() layout (param 0 = text_prompt, 1 = img_prompt)."""
    sd = {k: v.detach().cpu().clone() for k, v in prompt_state.items()}
    state = {}
    if momentum is not None and steps > 0:
        m = momentum.detach().cpu()
        state = {0: {"momentum_buffer": m[:n_text].reshape(sd["text_prompt"].shape).clone()},
                 1: {"momentum_buffer": m[n_text:].reshape(sd["img_prompt"].shape).clone()}}
    group = {"lr": lr, "momentum": oc.momentum, "dampening": 0, "weight_decay": oc.weight_decay, "nesterov": False,
             "maximize": False, "foreach": None, "differentiable": False, "fused": None, "initial_lr": oc.lr,
             "params": [0, 1]}
    return {"state_dict": sd, "epoch": int(epoch), "optimizer": {"state": state, "param_groups": [group]},
            "scheduler": {"last_epoch": int(epoch)}, "val_result": val_result, "steps": int(steps)}


def write_checkpoint(directory: str, ck: dict, epoch: int, is_best: bool = False) -> str:
    path = os.path.join(directory, "prompt_learner")
    os.makedirs(path, exist_ok=True)
    fn = os.path.join(path, f"model.pth.tar-{epoch}")
    torch.save(ck, fn)
    with open(os.path.join(path, "checkpoint"), "w") as f:      # Dassl's pointer file to the newest checkpoint
        f.write(os.path.basename(fn) + "\n")
    if is_best:
        import shutil
        shutil.copyfile(fn, os.path.join(path, "model-best.pth.tar"))
    return fn


def _momentum_from_optimizer_state(opt_state, numel: int) -> Optional[torch.Tensor]:
    if not opt_state or not opt_state.get("state"):
        return None
    st = opt_state["state"]
    try:
        bufs = [st[i]["momentum_buffer"] for i in (0, 1)]
    except (KeyError, TypeError):
        return None
    if any(b is None for b in bufs):
        return None
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    return flat if flat.numel() == numel else None
