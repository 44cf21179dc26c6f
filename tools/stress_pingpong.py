#!/usr/bin/env python3
"""Race screen of gemm_pp_kernel: many launches per shape, interleaved with other kernels that disturb timing,
every result compared bitwise with the first one and with the lock-step 256x256 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_QGELU
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
noise_a = torch.randn(4096, 4096, device=dev)
for (M, N, K) in [(7072, 2304, 768), (7072, 3072, 768), (4496, 3072, 1024), (2048, 1536, 64), (3000, 2312, 128), (14144, 2304, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    for epi in (EPI_BIAS, EPI_BIAS_QGELU):
        ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(a, w, ref, epi, bias=bias, tile_config=3)
        outs = [torch.empty_like(ref) for _ in range(4)]
        for it in range(300):
            o = outs[it % 4]
            ops.gemm_nt(a, w, o, epi, bias=bias, tile_config=7)
            if it % 3 == 0:
                noise_a.mul_(1.0001)                   # an unrelated memory-bound kernel in between
            if it % 4 == 3:
                for oo in outs:
                    if not torch.equal(oo, ref):
                        bad += 1
        print(M, N, K, epi, "mismatches so far:", bad, flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean")
