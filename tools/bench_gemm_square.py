#!/usr/bin/env python3
"""rpo_gemm_nt on large square bf16 problems (where does the kernel stand against the guide's templates?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import ops
from rpo_amd._lib import EPI_BIAS, EPI_NONE

dev = torch.device("cuda:0")
for n in (2048, 4096, 8192):
    for kdim in (768, n):
        a = torch.randn(n, kdim, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, kdim, device=dev) * kdim ** -0.5).to(torch.bfloat16)
        out = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
        bias = torch.randn(n, device=dev)
        for cfg in (2, 3):
            for _ in range(3):
                ops.gemm_nt(a, w, out, EPI_BIAS, bias=bias, tile_config=cfg)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gemm_nt(a, w, out, EPI_BIAS, bias=bias, tile_config=cfg)
            e.record(); e.synchronize()
            us = 1e3 * s.elapsed_time(e) / 10
            print(f"M=N={n} K={kdim} cfg{cfg}: {us:8.1f} us {2.0*n*n*kdim/us/1e6:7.1f} TF", flush=True)
