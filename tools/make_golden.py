#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on this repo's
synthetic workload.  Runs only in the build container (needs /root/reference);
the fixtures it writes are data (inputs are regenerated from seeds, outputs are
stored) and travel to the GPU box, the reference does not.

The reference imports six modules that are not installed here (dassl.engine,
dassl.metrics, dassl.utils, dassl.optim, torchvision.transforms, ftfy;
trainers/rpo.py:13-19, clip/clip.py:9, clip/simple_tokenizer.py:6).  They are
stubbed in sys.modules: the trainer registry decorator becomes the identity,
``ftfy.fix_text`` the identity (exact for ASCII class names).  No reference code
is modified or copied.

What is captured:
  G1  token ids + len_prompts of the Oxford-Pets base prompts (reference BPE)
  G3  eval logits, train loss, text_prompt.grad, img_prompt.grad for several
      (depth, K, B) including the full 12-layer ViT-B/16 at K=24, B=4
  G4  prompt rows after every block (depth 2) for kernel-level bisecting
  G5  prompts after 1 and 4 SGD steps with explicit (lr, momentum, wd)
  G7  the reference's OWN prompt initialisation (trainers/rpo.py:63-67, :77-81 draw from the global
      RNG) with torch.manual_seed set right before CustomCLIP is constructed, + the logits it gives:
      pins the reference-shaped constructor of rpo_amd.custom_clip
  G8  a checkpoint directory in the layout the reference's reader consumes
      (trainers/rpo.py:325-357: <dir>/prompt_learner/model-best.pth.tar and model.pth.tar-<epoch>
      with "state_dict" / "epoch", plus the token_prefix / token_suffix keys it deletes) holding
      the reference's prompts after one SGD step, and the eval logits those prompts give
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RPO_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from rpo_amd import synth  # noqa: E402
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, PROMPT_TEMPLATE, vit_b16  # noqa: E402


def _install_stubs() -> None:
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Registry:
        def register(self):
            return lambda cls: cls

    class TrainerX:  # noqa: D401 - placeholder base class
        pass

    mod("dassl")
    mod("dassl.engine", TRAINER_REGISTRY=_Registry(), TrainerX=TrainerX)
    mod("dassl.metrics", compute_accuracy=None)
    mod("dassl.utils", load_pretrained_weights=None, load_checkpoint=None)
    mod("dassl.optim", build_optimizer=None, build_lr_scheduler=None)
    mod("torchvision")
    mod("torchvision.transforms", Compose=None, Resize=None, CenterCrop=None, ToTensor=None,
        Normalize=None)
    mod("ftfy", fix_text=lambda s: s)


def _reference():
    _install_stubs()
    sys.path.insert(0, REF)
    from clip import clip as ref_clip            # noqa: E402
    from clip.model import CLIP                  # noqa: E402
    import trainers.rpo as ref_rpo               # noqa: E402
    return ref_clip, CLIP, ref_rpo


def build_reference_model(CLIP, ref_rpo, cfg, sd_np, classnames, seed=None):
    clip_model = CLIP(cfg.embed, cfg.image_size, cfg.layers_v, cfg.d_v, cfg.patch,
                      cfg.context, cfg.vocab, cfg.d_t, cfg.heads_t, cfg.layers_t).float()
    missing = clip_model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()},
                                         strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ns = types.SimpleNamespace
    rcfg = ns(TRAINER=ns(RPO=ns(K=cfg.K)), INPUT=ns(SIZE=(cfg.image_size, cfg.image_size)))
    if seed is not None:          # the generator state PromptLearner.initialization_token starts from (CLIP's own
        torch.manual_seed(seed)   # constructor draws too, so the seed is set between the two)
    model = ref_rpo.CustomCLIP(rcfg, list(classnames), PROMPT_TEMPLATE, clip_model)
    for name, p in model.named_parameters():          # trainers/rpo.py:258-260
        if "prompt_learner" not in name:
            p.requires_grad_(False)
    return model


def set_prompts(model, tp, ip):
    model.prompt_learner.text_prompt.data = torch.from_numpy(tp.copy())
    model.prompt_learner.img_prompt.data = torch.from_numpy(ip.copy())


def ref_train_eval(model, image, label):
    """(eval logits, train loss, grads) from the reference forward/backward."""
    pl = model.prompt_learner
    pl.eval()
    with torch.no_grad():
        logits = model(image).clone()
    pl.train()
    model.text_x = model.text_x.detach()               # SURVEY.md finding 6
    for p in pl.parameters():
        p.grad = None
    loss = model(image, label)
    loss.backward()
    model.text_x = model.text_x.detach()
    return logits, loss.detach(), pl.text_prompt.grad.clone(), pl.img_prompt.grad.clone()


def hook_prompt_rows(model, K, len_prompts):
    """Forward hooks on every residual block returning the prompt rows."""
    img_rows, text_rows, handles = [], [], []
    for blk in model.img_transformer.resblocks:
        handles.append(blk.register_forward_hook(
            lambda m, i, o: img_rows.append(o.detach().permute(1, 0, 2)[:, -K:, :].clone())))
    n_cls = len(len_prompts)
    ar = torch.arange(n_cls)
    for blk in model.text_transformers.resblocks:
        def hk(m, i, o):
            xb = o.detach().permute(1, 0, 2)
            text_rows.append(torch.stack([xb[ar, len_prompts + j] for j in range(K)], dim=1).clone())
        handles.append(blk.register_forward_hook(hk))
    return img_rows, text_rows, handles


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    ref_clip, CLIP, ref_rpo = _reference()
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    data_dir = os.path.join(REPO, "rpo_amd", "data")
    os.makedirs(data_dir, exist_ok=True)

    # ---- G1 tokens -----------------------------------------------------------
    texts = [PROMPT_TEMPLATE.replace("_", c) for c in OXFORD_PETS_BASE_CLASSES]
    toks = torch.cat([ref_clip.tokenize(p) for p in texts]).numpy().astype(np.int64)
    lens = (toks.argmax(-1) + 1).tolist()
    with open(os.path.join(data_dir, "tokens_oxford_pets_base.json"), "w") as f:
        json.dump({"classes": list(OXFORD_PETS_BASE_CLASSES), "template": PROMPT_TEMPLATE,
                   "len_prompts": lens, "tokens": toks.tolist(),
                   "source": "reference clip.tokenize (clip/clip.py:185-221) via tools/make_golden.py"}, f)
    print("G1 len_prompts", lens)

    only = set(sys.argv[1:])                 # e.g. `make_golden.py ckpt` regenerates G7 / G8 alone
    manifest = {}
    # (tag, depth, K, B, logit_scale, extras)
    cases = [] if only and "cases" not in only else [
        ("d1_k4_b2", 1, 4, 2, np.log(100.0), dict(rows=False, sgd=False)),
        ("d2_k8_b3", 2, 8, 3, np.log(100.0), dict(rows=True, sgd=True)),
        ("d2_k24_b2_init", 2, 24, 2, np.log(1 / 0.07), dict(rows=False, sgd=False)),
        ("d2_k16_b2", 2, 16, 2, np.log(100.0), dict(rows=False, sgd=False)),
        ("d2_k48_b2", 2, 48, 2, np.log(100.0), dict(rows=False, sgd=False)),
        ("d12_k24_b4", 12, 24, 4, np.log(100.0), dict(rows=False, sgd=True)),
    ]
    for tag, depth, K, B, ls, extra in cases:
        cfg = vit_b16(layers_v=depth, layers_t=depth, K=K)
        sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(ls))
        model = build_reference_model(CLIP, ref_rpo, cfg, sd, OXFORD_PETS_BASE_CLASSES)
        assert np.array_equal(model.text_tokenized.numpy(), toks)
        tp, ip = synth.prompts(cfg, sd, seed=7)
        set_prompts(model, tp, ip)
        image = torch.from_numpy(synth.images(cfg, B))
        label = torch.from_numpy(synth.labels(cfg, B))
        rec = {}
        if extra["rows"]:
            img_rows, text_rows, handles = hook_prompt_rows(model, K, model.len_prompts)
        logits, loss, gt, gi = ref_train_eval(model, image, label)
        if extra["rows"]:
            for h in handles:
                h.remove()
            # hooks fired twice (eval + train): keep the train pass
            rec["img_rows"] = torch.stack(img_rows[-depth:]).numpy()
            rec["text_rows"] = torch.stack(text_rows[-depth:]).numpy()
        rec.update(logits=logits.numpy(), loss=np.float32(loss.item()),
                   g_text=gt.numpy(), g_img=gi.numpy(),
                   label=label.numpy(), weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
        if extra["sgd"]:
            lr, mom, wd = 0.01, 0.9, 5e-4
            pl = model.prompt_learner
            opt = torch.optim.SGD(pl.parameters(), lr=lr, momentum=mom, weight_decay=wd)
            pl.train()
            losses = []
            for step in range(4):
                im = torch.from_numpy(synth.images(cfg, B, seed=1234 + 10 * step))
                lb = torch.from_numpy(synth.labels(cfg, B, seed=4321 + 10 * step))
                model.text_x = model.text_x.detach()
                l_ = model(im, lb)
                opt.zero_grad()
                l_.backward()
                opt.step()
                losses.append(l_.item())
                if step in (0, 3):
                    rec[f"text_prompt_step{step + 1}"] = pl.text_prompt.detach().numpy().copy()
                    rec[f"img_prompt_step{step + 1}"] = pl.img_prompt.detach().numpy().copy()
            rec["sgd_losses"] = np.asarray(losses, dtype=np.float32)
            rec["sgd_hparams"] = np.asarray([lr, mom, wd], dtype=np.float64)
        path = os.path.join(out_dir, f"ref_{tag}.npz")
        np.savez_compressed(path, **rec)
        manifest[tag] = dict(depth=depth, K=K, B=B, logit_scale=float(ls),
                             loss=float(loss), bytes=os.path.getsize(path))
        print(tag, "loss", float(loss), "|logits|max", float(logits.abs().max()),
              "|g_text|max", float(gt.abs().max()), "|g_img|max", float(gi.abs().max()))
    # ---- G7: seeded reference initialisation ------------------------------------
    cfg = vit_b16(layers_v=1, layers_t=1, K=4)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    model = build_reference_model(CLIP, ref_rpo, cfg, sd, OXFORD_PETS_BASE_CLASSES, seed=3)
    image = torch.from_numpy(synth.images(cfg, 2))
    model.prompt_learner.eval()
    with torch.no_grad():
        logits = model(image)
    np.savez_compressed(os.path.join(out_dir, "ref_init_seed3_d1_k4.npz"),
                        text_prompt=model.prompt_learner.text_prompt.detach().numpy(),
                        img_prompt=model.prompt_learner.img_prompt.detach().numpy(), logits=logits.numpy(),
                        seed=np.int64(3))
    print("G7 init |text_prompt|max", float(model.prompt_learner.text_prompt.abs().max()))

    # ---- G8: checkpoint in the reference reader's layout ------------------------
    tp, ip = synth.prompts(cfg, sd, seed=7)
    set_prompts(model, tp, ip)
    pl = model.prompt_learner
    pl.train()
    opt = torch.optim.SGD(pl.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    model.text_x = model.text_x.detach()
    l_ = model(image, torch.from_numpy(synth.labels(cfg, 2)))
    opt.zero_grad(); l_.backward(); opt.step()
    pl.eval()
    model.text_x = model.text_x.detach()
    with torch.no_grad():
        logits = model(image)
    ck_dir = os.path.join(out_dir, "ckpt_d1_k4", "prompt_learner")
    os.makedirs(ck_dir, exist_ok=True)
    state = {k: v.detach().clone() for k, v in pl.state_dict().items()}
    # CoOp-era checkpoints carry these two; the reader deletes them (trainers/rpo.py:348-352)
    state["token_prefix"] = torch.zeros(2, 1, cfg.d_t)
    state["token_suffix"] = torch.zeros(2, 3, cfg.d_t)
    # Dassl's save_checkpoint (un-vendored) stores these five keys; the reader uses the first two.  `scheduler` is
    # scheduler.state_dict() of what build_lr_scheduler returns for the yaml (WARMUP_EPOCH 1, constant): Dassl's
    # ConstantWarmupScheduler wrapping CosineAnnealingLR.  LRScheduler.state_dict() is every attribute but the
    # optimiser, so it HOLDS the successor CosineAnnealingLR object (and through it the SGD optimiser): a real
    # reference checkpoint is not a tensors-only pickle.  The wrapper is re-created from its published semantics
    # (tests/test_host_logic.py does the same); only its state dict is stored, not the class.
    class ConstantWarmupScheduler(torch.optim.lr_scheduler.LRScheduler):
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

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        succ = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=15)
        sched = ConstantWarmupScheduler(opt, succ, 1, 1e-5)
        sched.step(); sched.step()                                        # two epochs done
    ck = {"state_dict": state, "epoch": 2, "optimizer": opt.state_dict(), "scheduler": sched.state_dict(),
          "val_result": 12.5}
    torch.save(ck, os.path.join(ck_dir, "model.pth.tar-2"))
    torch.save(ck, os.path.join(ck_dir, "model-best.pth.tar"))
    np.savez_compressed(os.path.join(out_dir, "ref_ckpt_d1_k4.npz"),
                        text_prompt=pl.text_prompt.detach().numpy(), img_prompt=pl.img_prompt.detach().numpy(),
                        logits=logits.numpy(), momentum_text=opt.state_dict()["state"][0]["momentum_buffer"].numpy(),
                        momentum_img=opt.state_dict()["state"][1]["momentum_buffer"].numpy())
    print("G8 checkpoint written", ck_dir)

    if only and "cases" not in only:
        return
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(dict(generator="tools/make_golden.py", torch=torch.__version__,
                       numpy=np.__version__, cases=manifest), f, indent=1)


if __name__ == "__main__":
    main()
