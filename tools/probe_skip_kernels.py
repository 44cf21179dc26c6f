#!/usr/bin/env python3
"""Upper bound of what a fusion could give: the step with some launches simply SKIPPED (results are wrong, timing is what
is wanted).  Usage: python tools/probe_skip_kernels.py [name ...]   names: text_attn (24 launches), text_q (text q-proj
and d q: 24), head (2), ln_bwd_img (24).  Alternates baseline / skipped, 3 rounds of 100 steps, B = 32, bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rpo_amd import ops, synth
from rpo_amd.config import vit_b16
from rpo_amd.trainer import RPO

what = sys.argv[1:] or ["text_attn"]
cfg = vit_b16()
toks = synth.default_tokens(cfg)
sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
img = torch.randn(32, 3, 224, 224, device="cuda"); lab = torch.zeros(32, dtype=torch.int64, device="cuda")


def build(skip):
    keep = {}
    if skip:
        for w in what:
            if w == "text_attn":
                for n in ("text_attn_fwd", "text_attn_bwd"):
                    keep[n] = getattr(ops, n); setattr(ops, n, lambda *a, **k: None)
    tr = RPO(cfg, sd, toks, None, "cuda:0", torch.bfloat16, batch_size=32, num_batches=10 ** 9)
    for _ in range(3): tr.step_async(img, lab)          # captures the graphs with the patched ops
    torch.cuda.synchronize()
    for n, f in keep.items(): setattr(ops, n, f)
    return tr


def run(tr, steps=100):
    for _ in range(10): tr.step_async(img, lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): tr.step_async(img, lab)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


if what == ["control"]:                       # two identical trainers: is the second one built slower by itself?
    a, b = build(False), build(False)
elif os.environ.get("SKIP_FIRST") == "1":    # the skipped variant built first
    b, a = build(True), build(False)
else:
    a, b = build(False), build(True)
for r in range(3):
    print(f"round {r}: baseline {run(a):.4f} ms   without {'+'.join(what)} {run(b):.4f} ms")
