#!/usr/bin/env python3
"""How long does the HOST spend in each graph replay, and do two graphs on two streams overlap?"""
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
side = tr.engine.side
def timed(name, g, stream=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if stream is None: g.replay()
    else:
        with torch.cuda.stream(stream): g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:10s} host {1e6*(t1-t0):8.1f} us   total {1e6*(t2-t0):8.1f} us")
for _ in range(2):
    timed("text_fwd", tr._g_text_fwd); timed("img_fwd", tr._g_img_fwd); timed("head", tr._g_head)
    timed("text_bwd", tr._g_text_bwd); timed("img_bwd", tr._g_img_bwd)
# overlap test: image first on main, then text on side
for order in ("text-first", "img-first"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if order == "text-first":
        with torch.cuda.stream(side): tr._g_text_fwd.replay()
        tr._g_img_fwd.replay()
    else:
        tr._g_img_fwd.replay()
        with torch.cuda.stream(side): tr._g_text_fwd.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{order}: host {1e6*(t1-t0):8.1f} us total {1e6*(t2-t0):8.1f} us")
for order in ("text-first", "img-first"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if order == "text-first":
        with torch.cuda.stream(side): tr._g_text_bwd.replay()
        tr._g_img_bwd.replay()
    else:
        tr._g_img_bwd.replay()
        with torch.cuda.stream(side): tr._g_text_bwd.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"bwd {order}: host {1e6*(t1-t0):8.1f} us total {1e6*(t2-t0):8.1f} us")

# --- emulate the cross-step pipeline: frozen forward of batch i+1 (stream A) concurrently with the whole
# small-kernel chain of batch i (text fwd + [text fwd again ~ prompt-row image fwd] + head + both backward chains)
main = torch.cuda.current_stream()
sb, sc = torch.cuda.Stream(), torch.cuda.Stream()
ev1, ev2, ev3 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr._g_img_fwd.replay()                               # A: F(i+1)
    with torch.cuda.stream(sb):
        tr._g_text_fwd.replay()
        tr._g_text_fwd.replay()                          # stands in for the prompt-row image forward chain
        tr._g_head.replay()
        ev1.record(sb)
    sc.wait_event(ev1)
    with torch.cuda.stream(sc):
        tr._g_img_bwd.replay(); ev2.record(sc)
    with torch.cuda.stream(sb):
        tr._g_text_bwd.replay(); ev3.record(sb)
    main.wait_event(ev2); main.wait_event(ev3)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pipeline emulation: {1e6*(t2-t0):8.1f} us per step")

# --- does a high-priority stream for the image forward remove the text tower's interference? ---
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
for name, s_img, s_txt in (("img hi / text lo", hi, lo), ("img lo / text hi", lo, hi), ("both default", torch.cuda.current_stream(), side)):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s_txt): tr._g_text_fwd.replay()
        with torch.cuda.stream(s_img): tr._g_img_fwd.replay()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fwd pair, {name}: {1e6*(t2-t0):8.1f} us")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s_txt): tr._g_text_bwd.replay()
        with torch.cuda.stream(s_img): tr._g_img_bwd.replay()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"bwd pair, {name}: {1e6*(t2-t0):8.1f} us")
