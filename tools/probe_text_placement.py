#!/usr/bin/env python3
"""Where should the text tower's two chains run?  Wall time of the step's phases under three placements (B = 32, bf16; the
graphs of a captured step replayed out of order -- values are garbage, timing is what is wanted):
  now     : [img_fwd | text_fwd] -> head -> [img_bwd | text_bwd]
  early   : [img_fwd] -> head -> [img_bwd | text_bwd ; text_fwd]                 (RPO_EARLY_TEXT=1)
  deferred: [img_fwd | text_bwd ; text_fwd] -> head -> [img_bwd]                 (text backward of step i under img_fwd of i + 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
os.environ["RPO_EARLY_TEXT"] = "0"
cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=B, num_batches=10 ** 9)
img = torch.randn(B, 3, 224, 224, device="cuda"); lab = torch.zeros(B, dtype=torch.int64, device="cuda")
for _ in range(3): tr.step_async(img, lab)
torch.cuda.synchronize()
side, main = tr.engine.side, torch.cuda.current_stream()
e0, e1, e2, e3 = (torch.cuda.Event() for _ in range(4))

def step(mode):
    e0.record(main); side.wait_event(e0)
    with torch.cuda.stream(side):
        if mode == "deferred": tr._g_text_bwd.replay()
        if mode in ("now", "deferred"): tr._g_text_fwd.replay()
        e1.record(side)
    tr._g_img_fwd.replay(); main.wait_event(e1); tr._g_head.replay(); e2.record(main); side.wait_event(e2)
    with torch.cuda.stream(side):
        if mode in ("now", "early"): tr._g_text_bwd.replay()
        e3.record(side)                      # the main stream joins behind the text BACKWARD ...
        if mode == "early": tr._g_text_fwd.replay()     # ... the next step's text forward only has to beat its head (e1)
    tr._g_img_bwd.replay(); main.wait_event(e3)

def t(mode, n=100):
    for _ in range(10): step(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(mode)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

for r in range(3):
    print(f"B={B} round {r}: " + "  ".join(f"{m} {t(m):.4f} ms" for m in ("now", "early", "deferred")))
