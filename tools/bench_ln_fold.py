#!/usr/bin/env python3
"""Isolated cost of the LayerNorm-fold epilogues (include/rpo_amd.h RPO_EPI_LN_*) at the B=32 ViT-B/16 shapes:
producer (BIAS_RESID with / without out2 + ln_stats), consumer (LN_BIAS* vs BIAS*), and the stand-alone LayerNorm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_LN_BIAS, EPI_LN_BIAS_QGELU

dev = torch.device("cuda:0")
M, d = 7072, 768
dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    ts = []
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(7):
        s.record(); g.replay(); e.record(); e.synchronize()
        ts.append(1e3 * s.elapsed_time(e) / n)
    return sorted(ts)[3]


x = torch.randn(M, d, device=dev)
xb = x.to(dt)
stats = torch.zeros(M, d // 64, 2, device=dev)
stats[..., 1] = 64.0
att = torch.randn(M, d, device=dev).to(dt)
for name, N, K, epi_plain, epi_ln in (("in_proj", 2304, 768, EPI_BIAS, EPI_LN_BIAS), ("c_fc", 3072, 768, EPI_BIAS_QGELU, EPI_LN_BIAS_QGELU)):
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    bias, s = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=dt, device=dev)
    for cfg in (0, 8, 2, 10):
        a = timeit(lambda: ops.gemm_nt(xb, w, out, epi_plain, bias=bias, tile_config=cfg))
        b = timeit(lambda: ops.gemm_nt(xb, w, out, epi_ln, bias=bias, ln_stats=stats, ln_colsum=s, tile_config=cfg))
        print(f"{name:8s} cfg{cfg}: plain {a:6.2f} us   LN-fold {b:6.2f} us   ({b - a:+.2f})")
for name, K in (("out_proj", 768), ("c_proj", 3072)):
    a_ = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(d, K, device=dev) * K ** -0.5).to(dt)
    bias = torch.randn(d, device=dev)
    xo = torch.empty(M, d, device=dev)
    h = torch.empty(M, d, dtype=dt, device=dev)
    a = timeit(lambda: ops.gemm_nt(a_, w, xo, EPI_BIAS_RESID, bias=bias, resid=x))
    b = timeit(lambda: ops.gemm_nt(a_, w, xo, EPI_BIAS_RESID, bias=bias, resid=x, out2=h, ln_stats=stats))
    c = timeit(lambda: ops.gemm_nt(a_, w, xo, EPI_BIAS_RESID, bias=bias, resid=x, out2=h))
    print(f"{name:8s}: plain {a:6.2f} us   + out2 + stats {b:6.2f} us ({b - a:+.2f})   + out2 only {c:6.2f} us")
g_, b_ = torch.ones(d, device=dev), torch.zeros(d, device=dev)
h = torch.empty(M, d, dtype=dt, device=dev)
print(f"layernorm_fwd {M}x{d}: {timeit(lambda: ops.layernorm_fwd(x, g_, b_, h)):6.2f} us")
