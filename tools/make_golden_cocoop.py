#!/usr/bin/env python3
"""Golden vectors of the sibling trainer CoCoOp from the REAL reference (`trainers/cocoop.py`: CustomCLIP.forward
:166-192 + F.cross_entropy + backward), for a GIVEN context and meta-net: logits, loss and the gradients of every trained
tensor (ctx, meta_net.linear1 / linear2 weight + bias).  Same synthetic CLIP weights / images as tools/make_golden.py;
the token ids of its "X X .. name." prompts are stored with the vectors (data; the tokenizer is out of scope on the other
side).  Writes tests/golden/ref_cocoop_*.npz.  Runs in the build container only (needs /root/reference)."""
import os, sys, types
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from make_golden import _reference, REPO          # noqa: E402  (stubs the absent dassl / yacs imports, nothing copied)
from rpo_amd import synth                         # noqa: E402
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES, vit_b16  # noqa: E402

ref_clip, CLIP, _ = _reference()
import trainers.cocoop as ref_cocoop              # noqa: E402

ns = types.SimpleNamespace
for tag, depth, B, n_ctx in (("d2_b1_ctx4", 2, 1, 4), ("d2_b3_ctx4", 2, 3, 4)):
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    clip_model = CLIP(cfg.embed, cfg.image_size, cfg.layers_v, cfg.d_v, cfg.patch, cfg.context, cfg.vocab, cfg.d_t,
                      cfg.heads_t, cfg.layers_t).float()
    clip_model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    rcfg = ns(TRAINER=ns(COCOOP=ns(N_CTX=n_ctx, CTX_INIT="", PREC="fp32")), INPUT=ns(SIZE=(cfg.image_size, cfg.image_size)))
    model = ref_cocoop.CustomCLIP(rcfg, list(OXFORD_PETS_BASE_CLASSES), clip_model)
    for name, p in model.named_parameters():
        p.requires_grad_("prompt_learner" in name)             # trainers/cocoop.py:220-222
    rng = np.random.default_rng(13)
    h = cfg.embed // 16
    vals = dict(ctx=(rng.standard_normal((n_ctx, cfg.d_t)) * 0.02).astype(np.float32),
                w1=(rng.standard_normal((h, cfg.embed)) * cfg.embed ** -0.5).astype(np.float32),
                b1=(rng.standard_normal((h,)) * 0.1).astype(np.float32),
                w2=(rng.standard_normal((cfg.d_t, h)) * 0.05).astype(np.float32),
                b2=(rng.standard_normal((cfg.d_t,)) * 0.01).astype(np.float32))
    pl = model.prompt_learner
    pl.ctx.data = torch.from_numpy(vals["ctx"].copy())
    pl.meta_net.linear1.weight.data = torch.from_numpy(vals["w1"].copy())
    pl.meta_net.linear1.bias.data = torch.from_numpy(vals["b1"].copy())
    pl.meta_net.linear2.weight.data = torch.from_numpy(vals["w2"].copy())
    pl.meta_net.linear2.bias.data = torch.from_numpy(vals["b2"].copy())
    image = torch.from_numpy(synth.images(cfg, B))
    label = torch.from_numpy(synth.labels(cfg, B))
    pl.eval()
    with torch.no_grad():
        logits = model(image)
    pl.train()
    loss = model(image, label)                                 # :188-189 returns the cross-entropy in training mode
    loss.backward()
    path = os.path.join(REPO, "tests", "golden", f"ref_cocoop_{tag}.npz")
    np.savez_compressed(path, logits=logits.numpy(), loss=np.float32(loss.item()), label=label.numpy(),
                        tokenized_prompts=model.tokenized_prompts.numpy().astype(np.int64),
                        g_ctx=pl.ctx.grad.numpy(), g_w1=pl.meta_net.linear1.weight.grad.numpy(),
                        g_b1=pl.meta_net.linear1.bias.grad.numpy(), g_w2=pl.meta_net.linear2.weight.grad.numpy(),
                        g_b2=pl.meta_net.linear2.bias.grad.numpy(),
                        weights_crc=np.bytes_(synth.state_dict_checksum(sd)), **vals)
    print("cocoop", tag, "loss", float(loss), "|g_ctx|max", float(pl.ctx.grad.abs().max()), "|g_w1|max",
          float(pl.meta_net.linear1.weight.grad.abs().max()), os.path.getsize(path), "bytes")
