#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace`: replays ONLY the image-backward graph (then only the image-forward graph), each
preceded by a marker launch (sgd_kernel), so that tools/prof_chain.py can print the kernel-by-kernel timeline of one replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import ops, synth
from rpo_amd.config import vit_b16, vit_l14
from rpo_amd.trainer import RPO
LARGE = os.environ.get("MODEL") == "ViT-L/14"
cfg = vit_l14() if LARGE else vit_b16()
BATCH = 16 if LARGE else 32
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=BATCH, num_batches=10**9)
img = torch.randn(BATCH, 3, cfg.image_size, cfg.image_size, device="cuda"); lab = torch.zeros(BATCH, dtype=torch.int64, device="cuda")
for _ in range(3): tr.step_async(img, lab)
torch.cuda.synchronize()
p, g, b = (torch.zeros(256, device="cuda") for _ in range(3))
which = sys.argv[1] if len(sys.argv) > 1 else "img_bwd"
graph = {"img_bwd": tr._g_img_bwd, "img_fwd": tr._g_img_fwd, "text_fwd": tr._g_text_fwd, "text_bwd": tr._g_text_bwd}[which]
for _ in range(4):
    torch.cuda.synchronize()
    ops.sgd_step(p, g, b, 0.0, 0.0, 0.0, 1.0, first_step=False)
    graph.replay()
torch.cuda.synchronize()
