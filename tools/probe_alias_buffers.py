#!/usr/bin/env python3
"""Experiment: does the image forward get faster when the fp32 residual-stream buffers are reused across layers
(two ping-pong x buffers, one xm buffer) instead of one buffer per layer?  (Results are wrong for the backward -- the
per-layer buffers exist because the backward reads the prompt rows of every layer -- this only times the forward.)"""
import os, sys, time
os.environ.setdefault("RPO_NO_HILO", "1")   # this probe reads frozen rows of Engine.x / xm (see the invariant there)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO

cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
img = torch.randn(32, 3, 224, 224, device="cuda"); lab = torch.zeros(32, dtype=torch.int64, device="cuda")

def fwd_time(alias):
    tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=32, num_batches=10**9)
    eng = tr.engine
    if alias in ("x", "x+qkv"):
        eng.x = [eng.x[i % 2] for i in range(len(eng.x))]
        eng.xm = [eng.xm[0]] * len(eng.xm)
    if alias == "x+qkv":
        eng.qkv = [eng.qkv[0]] * len(eng.qkv)
        eng.u = [eng.u[0]] * len(eng.u)
    for _ in range(3): eng._image_forward(img, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng._image_forward(img, True)
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        ts.append(1e6 * (time.perf_counter() - t0))
    del tr
    return sorted(ts)[len(ts) // 2]

for rep in range(2):
    for alias in ("none", "x", "x+qkv"):
        print(f"image forward graph, aliasing {alias:6s}: {fwd_time(alias):8.1f} us")
