#!/usr/bin/env python3
"""Full-size golden vectors: the shapes bench.py itself runs, from the REAL reference.

tools/make_golden.py pins the oracle and the HIP path at small batches (B <= 4), where the GEMM heuristics pick the
generic tiles.  This script runs the reference (imported from /root/reference with the same six stubs, nothing copied)
on the bench's own shapes so that the model-level comparison goes through the one-round kernels
(gemm_w4 / gemm_w4g / gemm_w4k at M = 32 x 221) and the K = 48 generic path:

  ref_full_k24_b32    12 layers, K = 24, B = 32 (BASELINE.json configs[1]): eval logits, loss, both gradients, the
                      prompts after 1 and 2 SGD steps (explicit lr / momentum / weight decay) and the two losses
  ref_full_k{4,8,16,48}_b32   configs[4], the K sweep: eval logits, loss, both gradients
  (configs[3], ViT-L/14, 24 + 12 layers, B = 16: ref_full_vitl14_k24_b16 from tools/make_golden_vitl14_ref.py)

Inputs are regenerated from seeds on the GPU box (rpo_amd.synth); only outputs are stored (~0.3 MB).
Runs in the build container only: ~10 s per reference step on 8 cores.
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

from rpo_amd import synth  # noqa: E402
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, vit_b16  # noqa: E402

import make_golden as mg  # noqa: E402  (the reference harness: stubs, model builder, one train/eval pass)

SGD = (0.01, 0.9, 5e-4)


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    out_dir = os.path.join(REPO, "tests", "golden")
    manifest_path = os.path.join(out_dir, "manifest_fullsize.json")
    manifest = {}
    only = set(sys.argv[1:])
    ref_clip, CLIP, ref_rpo = mg._reference()
    toks = synth.oxford_pets_base_tokens()
    B = 32
    for K in (24, 4, 8, 16, 48):
        tag = f"full_k{K}_b{B}"
        if only and tag not in only:
            continue
        t0 = time.time()
        cfg = vit_b16(K=K)
        sd = synth.clip_state_dict(cfg, seed=0)
        model = mg.build_reference_model(CLIP, ref_rpo, cfg, sd, OXFORD_PETS_BASE_CLASSES)
        assert np.array_equal(model.text_tokenized.numpy(), toks)
        tp, ip = synth.prompts(cfg, sd, seed=7)
        mg.set_prompts(model, tp, ip)
        image = torch.from_numpy(synth.images(cfg, B))
        label = torch.from_numpy(synth.labels(cfg, B))
        logits, loss, gt, gi = mg.ref_train_eval(model, image, label)
        rec = dict(logits=logits.numpy(), loss=np.float32(loss.item()), g_text=gt.numpy(), g_img=gi.numpy(),
                   label=label.numpy(), weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
        if K == 24:
            lr, mom, wd = SGD
            pl = model.prompt_learner
            opt = torch.optim.SGD(pl.parameters(), lr=lr, momentum=mom, weight_decay=wd)
            pl.train()
            losses = []
            for step in range(2):
                im = torch.from_numpy(synth.images(cfg, B, seed=1234 + 10 * step))
                lb = torch.from_numpy(synth.labels(cfg, B, seed=4321 + 10 * step))
                model.text_x = model.text_x.detach()            # SURVEY.md finding 6
                l_ = model(im, lb)
                opt.zero_grad()
                l_.backward()
                opt.step()
                losses.append(l_.item())
                rec[f"text_prompt_step{step + 1}"] = pl.text_prompt.detach().numpy().copy()
                rec[f"img_prompt_step{step + 1}"] = pl.img_prompt.detach().numpy().copy()
            rec["sgd_losses"] = np.asarray(losses, dtype=np.float32)
            rec["sgd_hparams"] = np.asarray(SGD, dtype=np.float64)
        path = os.path.join(out_dir, f"ref_{tag}.npz")
        np.savez_compressed(path, **rec)
        manifest[tag] = dict(source="reference", model="ViT-B/16", depth=12, K=K, B=B, loss=float(loss),
                             bytes=os.path.getsize(path), seconds=round(time.time() - t0, 1))
        print(tag, manifest[tag], "|logits|max", float(logits.abs().max()), flush=True)
        del model

    # (configs[3], ViT-L/14: tools/make_golden_vitl14_ref.py -- the reference's own CustomCLIP at ViT-L/14 widths;
    #  until round 5 that fixture came from the oracle)

    old = {}
    if os.path.exists(manifest_path):
        with open(manifest_path) as f:
            old = json.load(f).get("cases", {})
    old.update(manifest)
    with open(manifest_path, "w") as f:
        json.dump(dict(generator="tools/make_golden_fullsize.py", torch=torch.__version__, numpy=np.__version__,
                       cases=old), f, indent=1)


if __name__ == "__main__":
    main()
