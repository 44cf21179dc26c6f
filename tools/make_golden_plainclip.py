#!/usr/bin/env python3
"""Golden vectors of the UNMASKED towers (plain CLIP inference) from the real reference: `clip.model.CLIP.forward`
(clip/model.py:344-372), which is what trainers/zsclip.py:58-63 and the sibling trainers' CustomCLIP.forward
(trainers/coop.py:196-208) run.  Same synthetic weights / images / token ids as tools/make_golden.py; writes
tests/golden/ref_plainclip_*.npz.  Runs in the build container only (needs /root/reference)."""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from make_golden import _reference, REPO          # noqa: E402  (stubs the absent dassl / yacs imports, nothing copied)
from rpo_amd import synth                         # noqa: E402
from rpo_amd.config import vit_b16                # noqa: E402

ref_clip, CLIP, _ = _reference()
toks = synth.oxford_pets_base_tokens()
only = [a for a in sys.argv[1:] if not a.startswith("-")]       # fixture tags to (re)generate; none: all of them
for tag, depth, B in (("d2_b3", 2, 3), ("d12_b2", 12, 2)):
    if only and tag not in only:
        continue
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    model = CLIP(cfg.embed, cfg.image_size, cfg.layers_v, cfg.d_v, cfg.patch, cfg.context, cfg.vocab, cfg.d_t,
                 cfg.heads_t, cfg.layers_t).float().eval()
    res = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    image = torch.from_numpy(synth.images(cfg, B))
    text = torch.from_numpy(toks)
    with torch.no_grad():
        logits, _ = model(image, text)
        img_f, txt_f = model.encode_image(image), model.encode_text(text)
    path = os.path.join(REPO, "tests", "golden", f"ref_plainclip_{tag}.npz")
    np.savez_compressed(path, logits=logits.numpy(), image_features=img_f.numpy(), text_features=txt_f.numpy(),
                        weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
    print(tag, "|logits|max", float(logits.abs().max()), os.path.getsize(path), "bytes")

# ---- CoOp (trainers/coop.py): learned context vectors in front of the class name, otherwise the unmasked towers --------
# logits, loss and d loss / d ctx of the reference's own CustomCLIP + F.cross_entropy for a GIVEN ctx (generic context,
# class token at the end: configs/trainers/CoOp/vit_b16_ep50.yaml defaults).  The token ids of its "X X .. name." prompts
# are saved with the vectors: they are data, and the BPE tokenizer is out of scope on the other side.
import types
import torch.nn.functional as F
import trainers.coop as ref_coop                  # noqa: E402  (same stubs as trainers/rpo.py)
from rpo_amd.config import OXFORD_PETS_BASE_CLASSES  # noqa: E402

ns = types.SimpleNamespace
# (tag, depth, batch, n_ctx, CSC, CLASS_TOKEN_POSITION): the defaults, then the reference's other options (round 4)
CASES = (("d2_b3_ctx4", 2, 3, 4, False, "end"), ("d2_b2_ctx16", 2, 2, 16, False, "end"),
         ("d2_b3_ctx4_csc", 2, 3, 4, True, "end"), ("d2_b2_ctx4_middle", 2, 2, 4, False, "middle"),
         ("d2_b2_ctx5_middle_csc", 2, 2, 5, True, "middle"), ("d2_b2_ctx4_front", 2, 2, 4, False, "front"))
for tag, depth, B, n_ctx, csc, position in CASES:
    if only and tag not in only:
        continue
    cfg = vit_b16(layers_v=depth, layers_t=depth, K=1)
    sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)))
    clip_model = CLIP(cfg.embed, cfg.image_size, cfg.layers_v, cfg.d_v, cfg.patch, cfg.context, cfg.vocab, cfg.d_t,
                      cfg.heads_t, cfg.layers_t).float()
    clip_model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    rcfg = ns(TRAINER=ns(COOP=ns(N_CTX=n_ctx, CTX_INIT="", CSC=csc, CLASS_TOKEN_POSITION=position, PREC="fp32")),
              INPUT=ns(SIZE=(cfg.image_size, cfg.image_size)))
    model = ref_coop.CustomCLIP(rcfg, list(OXFORD_PETS_BASE_CLASSES), clip_model)
    for name, p in model.named_parameters():
        p.requires_grad_("prompt_learner" in name)             # trainers/coop.py:228-230
    shape = (len(OXFORD_PETS_BASE_CLASSES), n_ctx, cfg.d_t) if csc else (n_ctx, cfg.d_t)
    ctx = (np.random.default_rng(11).standard_normal(shape) * 0.02).astype(np.float32)
    assert tuple(model.prompt_learner.ctx.shape) == shape
    model.prompt_learner.ctx.data = torch.from_numpy(ctx.copy())
    image = torch.from_numpy(synth.images(cfg, B))
    label = torch.from_numpy(synth.labels(cfg, B))
    logits = model(image)
    loss = F.cross_entropy(logits, label)                     # trainers/coop.py:268-269
    loss.backward()
    path = os.path.join(REPO, "tests", "golden", f"ref_coop_{tag}.npz")
    np.savez_compressed(path, logits=logits.detach().numpy(), loss=np.float32(loss.item()), ctx=ctx,
                        ctx_grad=model.prompt_learner.ctx.grad.numpy(), label=label.numpy(),
                        tokenized_prompts=model.tokenized_prompts.numpy().astype(np.int64),
                        name_lens=np.asarray(model.prompt_learner.name_lens, dtype=np.int64),
                        csc=np.bool_(csc), class_token_position=np.str_(position),
                        weights_crc=np.bytes_(synth.state_dict_checksum(sd)))
    print("coop", tag, "loss", float(loss), "|ctx_grad|max", float(model.prompt_learner.ctx.grad.abs().max()),
          os.path.getsize(path), "bytes")
