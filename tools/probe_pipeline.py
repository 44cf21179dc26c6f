#!/usr/bin/env python3
"""Emulates the cross-step pipeline with the existing graphs to see what contention does:
stream A (low priority): image forward of batch i+1; streams B, C (high priority): the small-kernel
chains of batch i (text fwd | [text fwd again ~ prompt-row image fwd]) -> head -> (text bwd | image bwd)."""
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
for prio in (False, True):
    lo, hi = (0, -1) if prio else (0, 0)
    sa = torch.cuda.Stream(priority=lo); sb = torch.cuda.Stream(priority=hi); sc = torch.cuda.Stream(priority=hi)
    evF = [torch.cuda.Event() for _ in range(64)]; evB = torch.cuda.Event(); evC = torch.cuda.Event(); evH = torch.cuda.Event(); evDone = torch.cuda.Event()
    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(sa):
            tr._g_img_fwd.replay(); evF[0].record(sa)            # F(0)
        for i in range(n):
            with torch.cuda.stream(sa):
                tr._g_img_fwd.replay(); evF[i + 1].record(sa)    # F(i+1), overlaps chains of step i
            with torch.cuda.stream(sb):
                if i: sb.wait_event(evDone)
                tr._g_text_fwd.replay(); evB.record(sb)
            with torch.cuda.stream(sc):
                if i: sc.wait_event(evDone)
                sc.wait_event(evF[i])                              # needs F(i)
                tr._g_text_fwd.replay()                            # ~ prompt-row image forward chain
                sc.wait_event(evB)
                tr._g_head.replay(); evH.record(sc)
                tr._g_img_bwd.replay(); evC.record(sc)
            with torch.cuda.stream(sb):
                sb.wait_event(evH)
                tr._g_text_bwd.replay()
                sb.wait_event(evC)
                evDone.record(sb)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    run(3)
    print(f"priorities={prio}: pipelined emulation {1e3 * run(20):.3f} ms per step  (sequential now ~3.9 ms)")
