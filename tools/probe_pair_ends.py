#!/usr/bin/env python3
"""Which chain ends last when the text and image graphs of a phase run side by side?  Events at the end of each stream."""
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
main, side = torch.cuda.current_stream(), tr.engine.side
for name, gi, gt in (("forward", tr._g_img_fwd, tr._g_text_fwd), ("backward", tr._g_img_bwd, tr._g_text_bwd)):
    res = []
    for _ in range(7):
        e0, ei, et = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(main); side.wait_event(e0)
        with torch.cuda.stream(side):
            gt.replay(); et.record(side)
        gi.replay(); ei.record(main)
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(ei) * 1e3, e0.elapsed_time(et) * 1e3))
    res.sort()
    print(f"{name}: image chain ends at {res[3][0]:7.1f} us, text chain at {res[3][1]:7.1f} us (median of 7)")
