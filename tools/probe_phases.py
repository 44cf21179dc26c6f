#!/usr/bin/env python3
"""Wall time of the five step graphs alone and of the overlapped pairs for any (batch, K, model): the short form of
probe_graph_launch.py.  Usage: python tools/probe_phases.py [batch] [K] [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16, vit_l14
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = (vit_l14 if (len(sys.argv) > 3 and "L" in sys.argv[3]) else vit_b16)(K=K)
from rpo_amd.trainer import RPO
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10**9)
img = torch.randn(B, 3, 224, 224, device="cuda"); lab = torch.zeros(B, dtype=torch.int64, device="cuda")
for _ in range(3): tr.step_async(img, lab)
torch.cuda.synchronize()
side = tr.engine.side
def timed(fn, reps=20):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6
def pair(a, b):
    def f():
        with torch.cuda.stream(side): a.replay()
        b.replay()
    return f
print(f"B={B} K={K}")
for n in ("text_fwd", "img_fwd", "head") + (() if tr._joint_bwd else ("text_bwd", "img_bwd")):
    print(f"{n:10s} {timed(getattr(tr, '_g_' + n).replay):8.1f} us")
print(f"fwd pair   {timed(pair(tr._g_text_fwd, tr._g_img_fwd)):8.1f} us")
if tr._joint_bwd:
    print(f"joint bwd  {timed(tr._g_bwd.replay):8.1f} us")
else:
    print(f"bwd pair   {timed(pair(tr._g_text_bwd, tr._g_img_bwd)):8.1f} us")
def step():
    tr.step_async(img, lab)
t = timed(lambda: [step() for _ in range(20)], reps=5) / 20
print(f"step       {t:8.1f} us")
