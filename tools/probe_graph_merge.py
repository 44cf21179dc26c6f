#!/usr/bin/env python3
"""What a graph boundary on ONE stream costs: head ; image backward as two replays against one captured graph, and
image forward ; head ; image backward as three against one (B = 32, bf16; wall time over 50 replays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO
cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=32, num_batches=10**9)
img = torch.randn(32, 3, 224, 224, device="cuda"); lab = torch.zeros(32, dtype=torch.int64, device="cuda")
for _ in range(3): tr.step_async(img, lab)
torch.cuda.synchronize()
eng = tr.engine
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.head(32, tr._label); eng._image_backward(32)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    eng._image_forward(tr._image, train=True); eng.head(32, tr._label); eng._image_backward(32)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
for rep in range(2):
    print("head ; img_bwd as two graphs   %.1f us" % t(lambda: (tr._g_head.replay(), tr._g_img_bwd.replay())))
    print("head + img_bwd as one graph    %.1f us" % t(lambda: g.replay()))
    print("img_fwd ; head ; img_bwd (3)   %.1f us" % t(lambda: (tr._g_img_fwd.replay(), tr._g_head.replay(), tr._g_img_bwd.replay())))
    print("img_fwd + head + img_bwd (1)   %.1f us" % t(lambda: g2.replay()))
