#!/usr/bin/env python3
"""What does the memory-side cache (256 MB Infinity Cache) give a streaming kernel?  Device-to-device copies of a buffer
pair of growing size, repeated: a pair that fits the cache is served by it from the second pass on.  Uses the library's
own copy kernel (rpo_probe_peak_copy) and torch's for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    a, b = torch.randn(n, device=dev), torch.empty(n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("lib", lambda: lib.rpo_probe_peak_copy(a.data_ptr(), b.data_ptr(), n * 4, st)), ("torch", lambda: b.copy_(a))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(20, 4096 // mb)
        s.record()
        for _ in range(reps): fn()
        e.record(); e.synchronize()
        res[name] = 2 * n * 4 * reps / (s.elapsed_time(e) * 1e-3) / 1e12
    print(f"copy of {mb:5d} MB (pair {2 * mb:5d} MB): lib {res['lib']:5.2f} TB/s   torch {res['torch']:5.2f} TB/s (read + write)", flush=True)
