#!/usr/bin/env python3
"""Race screen of the whole graph-replayed train step: two runs of 300 steps (B=32, ViT-B/16, K=24, bf16) over the same
four batches must end in bit-identical prompts (all kernels are deterministic: no atomics, fixed split-K order)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO
cfg = vit_b16(K=24)
toks = synth.oxford_pets_base_tokens()
sd = synth.clip_state_dict(cfg, seed=0, logit_scale=float(np.log(100.0)), token_rows=np.unique(toks).tolist() + [49407])
pr = synth.prompts(cfg, sd, seed=7)
B = 32
imgs = [torch.from_numpy(synth.images(cfg, B, seed=100 + i)).cuda() for i in range(4)]
labs = [torch.from_numpy(synth.labels(cfg, B, seed=200 + i)).cuda() for i in range(4)]
outs = []
for run in range(2):
    tr = RPO(cfg, sd, toks, device="cuda:0", act_dtype=torch.bfloat16, batch_size=B, prompts=pr)
    for s in range(300):
        tr.step_async(imgs[s % 4], labs[s % 4])
    torch.cuda.synchronize()
    outs.append((tr.engine.params.clone(), tr.engine.loss.clone()))
    del tr
same, finite = torch.equal(outs[0][0], outs[1][0]), torch.isfinite(outs[0][0]).all().item()
print("two runs of 300 graph-replayed steps (B = 32, bf16): bitwise equal params:", same, "loss:", outs[0][1].item(),
      outs[1][1].item(), "finite:", finite)
sys.exit(0 if same and finite else 1)
