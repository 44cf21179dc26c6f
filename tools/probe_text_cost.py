#!/usr/bin/env python3
"""Marginal in-step cost of the text tower's two chains: the step's own graph sequence (RPO._replay + SGD) replayed
back to back with the text forward and / or the text backward left out (their buffers keep the previous values, so the
numbers are timing only).  Usage: python tools/probe_text_cost.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import ops, synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10**9)
img = torch.randn(B, 3, 224, 224, device="cuda"); lab = torch.zeros(B, dtype=torch.int64, device="cuda")
for _ in range(3): tr.step_async(img, lab)
torch.cuda.synchronize()
eng, side, main = tr.engine, tr.engine.side, torch.cuda.current_stream()
ev = [torch.cuda.Event() for _ in range(4)]
def step(text_fwd=True, text_bwd=True):
    tr._image.copy_(img, non_blocking=True); tr._label.copy_(lab, non_blocking=True)
    ev[0].record(main)
    if text_fwd:
        side.wait_event(ev[0])
        with torch.cuda.stream(side):
            tr._g_text_fwd.replay(); ev[1].record(side)
    tr._g_img_fwd.replay()
    if text_fwd: main.wait_event(ev[1])
    tr._g_head.replay()
    ev[2].record(main)
    if text_bwd:
        side.wait_event(ev[2])
        with torch.cuda.stream(side):
            tr._g_text_bwd.replay(); ev[3].record(side)
    tr._g_img_bwd.replay()
    if text_bwd: main.wait_event(ev[3])
    ops.sgd_step(eng.params, eng.grads, eng.mom, 1e-9, 0.9, 0.0, 1.0, first_step=False)
def timed(**kw):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): step(**kw)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
    return best * 1e6
full = timed()
print(f"B={B}: full step {full:8.1f} us")
for name, kw in (("no text fwd", dict(text_fwd=False)), ("no text bwd", dict(text_bwd=False)), ("no text at all", dict(text_fwd=False, text_bwd=False))):
    t = timed(**kw)
    print(f"  {name:15s} {t:8.1f} us  ({full - t:+7.1f} us = what that chain costs the step)")
