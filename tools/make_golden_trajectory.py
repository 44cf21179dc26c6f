#!/usr/bin/env python3
"""A whole training run of the REAL reference as a fixture: BASELINE.json configs[0] -- ViT-B/16 (12 + 12 layers),
K = 24 prompts, batch 4, 15 epochs x 4 iterations, SGD (lr 0.01, momentum 0.9, weight decay 5e-4 passed explicitly:
Dassl's defaults are un-vendored), one constant warm-up epoch at 1e-5, then cosine decay, `update_lr` after the last
batch of every epoch (configs/trainers/RPO/main_K24.yaml:1-22, trainers/rpo.py:306-314).

What it pins that the 2- and 4-step fixtures do not: schedule x fused SGD x kernels END TO END -- "learned prompt
embeddings match the reference" over the run the north star names.  Stored: both prompt tensors after epochs 1, 5 and
15, the 60 losses, the learning rate of every epoch.  Inputs are regenerated from seeds (rpo_amd.synth): iteration i of
EVERY epoch sees the same 4 images / labels (seeds 1234 + 10 i / 4321 + 10 i) -- a 16-image few-shot set walked in a
fixed order, like Oxford-Pets 1-shot at batch 4 with drop_last.

Runs only in the build container (imports /root/reference with the six missing modules stubbed exactly as
tools/make_golden.py does; nothing of the reference is copied).  ~1.5 minutes of CPU.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

from make_golden import _reference, build_reference_model, set_prompts  # noqa: E402
from rpo_amd import synth  # noqa: E402
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, vit_b16  # noqa: E402

LR, MOM, WD = 0.01, 0.9, 5e-4
MAX_EPOCH, ITERS, B, K = 15, 4, 4, 24
WARMUP_EPOCH, CONS_LR = 1, 1e-5
KEEP = (1, 5, 15)


class ConstantWarmupScheduler(torch.optim.lr_scheduler.LRScheduler):
    """Dassl's warm-up wrapper, re-created from its published semantics on torch's scheduler base class (as
    tools/make_golden.py G8 and tests/test_host_logic.py do): constant `cons_lr` for `warmup_epoch` epochs, during
    which the successor is NOT stepped; afterwards every step() steps the successor and reports its rate."""

    def __init__(self, optimizer, successor, warmup_epoch, cons_lr):
        self.successor, self.warmup_epoch, self.cons_lr = successor, warmup_epoch, cons_lr
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch >= self.warmup_epoch:
            return self.successor.get_last_lr()
        return [self.cons_lr for _ in self.base_lrs]

    def step(self, epoch=None):
        if self.last_epoch >= self.warmup_epoch:
            self.successor.step(epoch)
            self._last_lr = self.successor.get_last_lr()
        else:
            super().step(epoch)


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    _, CLIP, ref_rpo = _reference()
    cfg = vit_b16(K=K)
    ls = float(np.log(100.0))
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=ls)
    model = build_reference_model(CLIP, ref_rpo, cfg, sd, OXFORD_PETS_BASE_CLASSES)
    tp, ip = synth.prompts(cfg, sd, seed=7)
    set_prompts(model, tp, ip)
    pl = model.prompt_learner
    # build_optimizer(self.model.prompt_learner, cfg.OPTIM) / build_lr_scheduler(self.optim, cfg.OPTIM), trainers/rpo.py:274-275
    opt = torch.optim.SGD(pl.parameters(), lr=LR, momentum=MOM, weight_decay=WD)
    cosine = torch.optim.lr_scheduler.CosineAnnealingLR(opt, float(MAX_EPOCH))
    sched = ConstantWarmupScheduler(opt, cosine, WARMUP_EPOCH, CONS_LR)
    batches = [(torch.from_numpy(synth.images(cfg, B, seed=1234 + 10 * i)), torch.from_numpy(synth.labels(cfg, B, seed=4321 + 10 * i)))
               for i in range(ITERS)]
    rec, losses, lrs = {}, [], []
    pl.train()
    t0 = time.time()
    for epoch in range(MAX_EPOCH):
        lrs.append(opt.param_groups[0]["lr"])
        for it in range(ITERS):
            image, label = batches[it]
            model.text_x = model.text_x.detach()          # SURVEY.md finding 6 (CPU-only aliasing of the cached text_x)
            loss = model(image, label)                    # trainers/rpo.py:306
            opt.zero_grad()                               # :307
            loss.backward()                               # :308
            opt.step()                                    # :309
            losses.append(loss.item())                    # :311
            if it + 1 == ITERS:
                sched.step()                              # :313-314 update_lr
        if epoch + 1 in KEEP:
            rec[f"text_prompt_e{epoch + 1}"] = pl.text_prompt.detach().numpy().copy()
            rec[f"img_prompt_e{epoch + 1}"] = pl.img_prompt.detach().numpy().copy()
        print(f"epoch {epoch + 1:2d} lr {lrs[-1]:.6g} loss {np.mean(losses[-ITERS:]):.6f}  ({time.time() - t0:.0f} s)", flush=True)
    pl.eval()
    model.text_x = model.text_x.detach()
    with torch.no_grad():
        logits = model(batches[0][0])
    rec.update(losses=np.asarray(losses, dtype=np.float32), lrs=np.asarray(lrs, dtype=np.float64),
               final_logits=logits.numpy(), hparams=np.asarray([LR, MOM, WD, MAX_EPOCH, ITERS, B, WARMUP_EPOCH, CONS_LR], dtype=np.float64),
               weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
    out = os.path.join(REPO, "tests", "golden", "ref_traj_d12_k24_b4_e15.npz")
    np.savez_compressed(out, **rec)
    man_path = os.path.join(REPO, "tests", "golden", "manifest_fullsize.json")
    man = json.load(open(man_path))
    man["cases"]["traj_d12_k24_b4_e15"] = dict(
        source="reference", generator="tools/make_golden_trajectory.py", model="ViT-B/16", K=K, B=B, epochs=MAX_EPOCH,
        iters_per_epoch=ITERS, lr=LR, momentum=MOM, weight_decay=WD, warmup_epoch=WARMUP_EPOCH, warmup_cons_lr=CONS_LR,
        first_loss=float(losses[0]), last_loss=float(losses[-1]), bytes=os.path.getsize(out))
    json.dump(man, open(man_path, "w"), indent=1)
    print("wrote", out, os.path.getsize(out), "bytes; loss", losses[0], "->", losses[-1])


if __name__ == "__main__":
    main()
