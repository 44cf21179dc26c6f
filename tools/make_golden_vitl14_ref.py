#!/usr/bin/env python3
"""ViT-L/14 (BASELINE.json configs[3]) pinned by the reference's OWN code.

trainers/rpo.py cannot run ViT-L/14 as shipped: four dimension literals are those of ViT-B/16 -- `self.d_v = 768`
(:52), `attn_head = 8` (:142), `1 + 14 * 14 + K` (:154) and the `512` of an empty accumulator (:185) -- and
clip/clip.py has no ViT-L/14 download.  Everything else on the path is dimension-generic: clip/model.py's `CLIP`,
`VisionTransformer`, `Transformer`, `ResidualAttentionBlock`, `LayerNorm`, `QuickGELU` take their sizes as arguments,
and `CustomCLIP.forward` (:161-232) reads every size but that one `512` from its tensors.  Until round 5 this config was
pinned only by the repo's own restatement (oracle/rpo_oracle.py); here the reference's real `CustomCLIP` -- its forward,
its autograd, torch's `nn.MultiheadAttention` under it -- runs at ViT-L/14 widths, with exactly those four literals
supplied from OUTSIDE the imported module (no reference source is modified or copied):

  * `PromptLearner.initialization_token` (draws `randn(K, 768)` against a 1024-wide class embedding) is replaced, for
    the construction only, by one that allocates the two parameters at the model's widths; their VALUES are then set
    from this repo's generator, as every fixture does (tools/make_golden.py: set_prompts);
  * `define_mask`'s result is rebuilt after construction by the same statements with `attn_head = width // 64` and the
    model's patch grid (the masks are data on the module: `model.text_mask`, `model.visual_mask`);
  * the module-level name `torch` of trainers.rpo is wrapped so that `torch.empty(n, 0, 512, ...)` returns the empty
    accumulator at the embedding width (every other attribute is torch's own).

Writes tests/golden/ref_full_vitl14_k24_b16.npz (24 + 12 layers, batch 16: eval logits, loss, both prompt gradients)
and ref_vitl14_d2_k24_b2.npz (2 + 2 layers, batch 2, plus the prompt rows after every block) -- the small one is what
the CPU suite holds the oracle to.  ~2 minutes and ~20 GB of host memory for the full one.
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import make_golden as mg  # noqa: E402
from rpo_amd import synth  # noqa: E402
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, PROMPT_TEMPLATE, vit_l14  # noqa: E402


class _TorchWithWideAccumulator:
    """`torch` as trainers.rpo sees it, except for the one literal: torch.empty(n, 0, 512, ...) -> (n, 0, embed)."""

    def __init__(self, embed: int):
        self._embed = embed

    def __getattr__(self, name):
        return getattr(torch, name)

    def empty(self, *size, **kw):
        if len(size) == 3 and size[1] == 0 and size[2] == 512:
            size = (size[0], 0, self._embed)
        return torch.empty(*size, **kw)


def build(CLIP, ref_rpo, cfg, sd_np):
    clip_model = CLIP(cfg.embed, cfg.image_size, cfg.layers_v, cfg.d_v, cfg.patch,
                      cfg.context, cfg.vocab, cfg.d_t, cfg.heads_t, cfg.layers_t).float()
    res = clip_model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ns = types.SimpleNamespace
    rcfg = ns(TRAINER=ns(RPO=ns(K=cfg.K)), INPUT=ns(SIZE=(cfg.image_size, cfg.image_size)))

    def init_at_model_widths(self, clip_model_):
        self.text_prompt = torch.nn.Parameter(torch.zeros(self.K, self.d_t, dtype=self.dtype))
        self.img_prompt = torch.nn.Parameter(torch.zeros(self.K, clip_model_.visual.class_embedding.shape[0], dtype=self.dtype))

    keep = ref_rpo.PromptLearner.initialization_token
    ref_rpo.PromptLearner.initialization_token = init_at_model_widths
    ref_rpo.torch = _TorchWithWideAccumulator(cfg.embed)
    try:
        # (define_mask inside __init__ builds ViT-B/16-sized masks; they are replaced below)
        model = ref_rpo.CustomCLIP(rcfg, list(OXFORD_PETS_BASE_CLASSES), PROMPT_TEMPLATE, clip_model)
    finally:
        ref_rpo.PromptLearner.initialization_token = keep
    # ---- define_mask (trainers/rpo.py:140-159) with the model's head count and patch grid
    len_max, attn_head = cfg.context, cfg.heads_t
    text_mask = torch.empty(0, len_max, len_max)
    for idx in model.len_prompts:
        mask = torch.empty(len_max, len_max)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        mask[:, idx:].fill_(float("-inf"))
        text_mask = torch.cat([text_mask, mask.repeat(attn_head, 1, 1)])
    model.text_mask = text_mask
    att_size = 1 + cfg.n_patches + cfg.K
    visual_mask = torch.zeros((att_size, att_size), dtype=model.dtype, requires_grad=False)
    visual_mask[:, -1 * cfg.K:] = float("-inf")
    model.visual_mask = visual_mask
    for name, p in model.named_parameters():          # trainers/rpo.py:258-260
        if "prompt_learner" not in name:
            p.requires_grad_(False)
    return model


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    out_dir = os.path.join(REPO, "tests", "golden")
    _, CLIP, ref_rpo = mg._reference()
    toks = synth.oxford_pets_base_tokens()
    manifest = {}
    only = set(sys.argv[1:])
    for tag, kw, B, rows in (("vitl14_d2_k24_b2", dict(layers_v=2, layers_t=2, K=24), 2, True),
                             ("full_vitl14_k24_b16", dict(K=24), 16, False)):
        if only and tag not in only:
            continue
        t0 = time.time()
        cfg = vit_l14(**kw)
        sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
        model = build(CLIP, ref_rpo, cfg, sd)
        assert np.array_equal(model.text_tokenized.numpy(), toks)
        tp, ip = synth.prompts(cfg, sd, seed=7)
        mg.set_prompts(model, tp, ip)
        image = torch.from_numpy(synth.images(cfg, B))
        label = torch.from_numpy(synth.labels(cfg, B))
        rec = {}
        if rows:
            img_rows, text_rows, handles = mg.hook_prompt_rows(model, cfg.K, model.len_prompts)
        logits, loss, gt, gi = mg.ref_train_eval(model, image, label)
        if rows:
            for h in handles:
                h.remove()
            rec["img_rows"] = torch.stack(img_rows[-cfg.layers_v:]).numpy()       # the train pass (hooks fired twice)
            rec["text_rows"] = torch.stack(text_rows[-cfg.layers_t:]).numpy()[:, :4]     # (4 classes: 0.6 MB instead of 2.8)
        rec.update(logits=logits.numpy(), loss=np.float32(loss.item()), g_text=gt.numpy(), g_img=gi.numpy(),
                   label=label.numpy(), weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
        path = os.path.join(out_dir, f"ref_{tag}.npz")
        np.savez_compressed(path, **rec)
        manifest[tag] = dict(source="reference", how="trainers/rpo.py CustomCLIP at ViT-L/14 widths, four dimension "
                             "literals supplied by tools/make_golden_vitl14_ref.py", model="ViT-L/14",
                             depth=cfg.layers_v, K=cfg.K, B=B, loss=float(loss), bytes=os.path.getsize(path),
                             seconds=round(time.time() - t0, 1))
        print(tag, manifest[tag], "|logits|max", float(logits.abs().max()), "|g_text|max", float(gt.abs().max()),
              "|g_img|max", float(gi.abs().max()), flush=True)
        del model
    ref_rpo.torch = torch
    mp = os.path.join(out_dir, "manifest_fullsize.json")
    man = json.load(open(mp))
    man["cases"].update(manifest)
    json.dump(man, open(mp, "w"), indent=1)


if __name__ == "__main__":
    main()
