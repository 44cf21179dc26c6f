#!/usr/bin/env python3
"""rpo_head_fwd_bwd alone: the cosine-logit head + cross-entropy, forward and backward, per launch (graph replay of 20 calls,
median of 7).  RPO_HIP_LIB=<variant .so> selects a build.  Usage: python tools/bench_head.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import ops as o
dev = torch.device("cuda:0")
for B, C, K, e in ((32, 19, 24, 512), (32, 100, 24, 512), (32, 397, 24, 512), (32, 1000, 24, 512), (16, 1000, 24, 768)):
    g = torch.Generator().manual_seed(1)
    i_f, t_f = torch.randn(B, K, e, generator=g).to(dev), torch.randn(C, K, e, generator=g).to(dev)
    lab = torch.tensor([(7 * b + 1) % C for b in range(B)], device=dev)
    logits, loss = torch.empty(B, C, device=dev), torch.empty(1, device=dev)
    d_i, d_t = torch.empty(B, K, e, device=dev), torch.empty(C, K, e, device=dev)
    ia, ta = torch.empty(B, K, e, dtype=torch.bfloat16, device=dev), torch.empty(C, K, e, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(o.head_workspace_floats(B, C, K, e), device=dev)
    run = lambda: o.head_fwd_bwd(i_f, t_f, lab, 100.0, logits, loss, d_i, d_t, ws, d_img_f_act=ia, d_text_f_act=ta)
    for _ in range(3):
        run()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            run()
    gr.replay()
    ts = []
    for _ in range(7):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); t.record(); t.synchronize()
        ts.append(1e3 * s.elapsed_time(t) / 20)
    print(f"B={B:3d} C={C:5d} K={K} e={e}: {sorted(ts)[3]:8.1f} us per forward + backward   loss {loss.item():.6f}")
