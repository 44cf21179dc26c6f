#!/usr/bin/env python3
"""Micro-benchmark of rpo_gemm_nt at the shapes of the B=32 ViT-B/16 step (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rpo_amd import ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_NONE, EPI_QGELU_BWD

dev = torch.device("cuda:0")
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
SHAPES = [  # name, M, N, K, epi, out dtype, split
    ("qkv", 7072, 2304, 768, EPI_BIAS, torch.bfloat16, 1),
    ("out_proj", 7072, 768, 768, EPI_BIAS_RESID, torch.float32, 1),
    ("c_fc", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("c_proj", 7072, 768, 3072, EPI_BIAS_RESID, torch.float32, 1),
    ("last_kv", 6304, 1536, 768, EPI_BIAS, torch.bfloat16, 1),           # K / V of the frozen rows in the last image block
    ("c_fc_a", 5376, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),      # one full round of 256x256 tiles
    ("c_fc_b", 1696, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),      # the rest of c_fc's rows
    ("bwd_du", 768, 3072, 768, EPI_QGELU_BWD, torch.bfloat16, 1),
    ("bwd_dh2", 768, 768, 3072, EPI_NONE, torch.float32, 8),
    ("bwd_dh2", 768, 768, 3072, EPI_NONE, torch.float32, 2),
    ("bwd_dh2", 768, 768, 3072, EPI_NONE, torch.float32, 3),
    ("bwd_dh2", 768, 768, 3072, EPI_NONE, torch.float32, 4),
    ("bwd_dh2", 768, 768, 3072, EPI_NONE, torch.float32, 6),
    ("bwd_dh1", 768, 768, 768, EPI_NONE, torch.float32, 1),
    ("bwd_dh1", 768, 768, 768, EPI_NONE, torch.float32, 2),
    ("txt_dh2", 456, 512, 2048, EPI_NONE, torch.float32, 2),
    ("txt_dh2", 456, 512, 2048, EPI_NONE, torch.float32, 4),
    ("txt_dh1", 456, 512, 512, EPI_NONE, torch.float32, 1),
    ("txt_dh1", 456, 512, 512, EPI_NONE, torch.float32, 2),
    ("bwd_da", 768, 768, 768, EPI_NONE, torch.bfloat16, 1),
    ("bwd_dh1", 768, 768, 768, EPI_NONE, torch.float32, 4),
    ("txt_q", 456, 512, 512, EPI_BIAS, torch.bfloat16, 1),
    ("txt_fc", 456, 2048, 512, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("txt_proj", 456, 512, 2048, EPI_BIAS_RESID, torch.float32, 1),
    # the image forward at small batches (M = B * 221): every GEMM on the small-M tiles
    ("b4_qkv", 884, 2304, 768, EPI_BIAS, torch.bfloat16, 1),
    ("b4_out", 884, 768, 768, EPI_BIAS_RESID, torch.float32, 1),
    ("b4_fc", 884, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("b4_proj", 884, 768, 3072, EPI_BIAS_RESID, torch.float32, 1),
    ("b8_fc", 1768, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("b8_proj", 1768, 768, 3072, EPI_BIAS_RESID, torch.float32, 1),
    ("b8_qkv", 1768, 2304, 768, EPI_BIAS, torch.bfloat16, 1),
    ("b8_out", 1768, 768, 768, EPI_BIAS_RESID, torch.float32, 1),
    ("b16_qkv", 3536, 2304, 768, EPI_BIAS, torch.bfloat16, 1),
    ("b16_out", 3536, 768, 768, EPI_BIAS_RESID, torch.float32, 1),
    ("b16_fc", 3536, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("b16_proj", 3536, 768, 3072, EPI_BIAS_RESID, torch.float32, 1),
    # the text tower at 1000 classes x K = 24 (configs/trainers/RPO/imagenet_k24_ep15.yaml): 24 000 prompt rows
    ("t1k_q", 24000, 512, 512, EPI_BIAS, torch.bfloat16, 1),
    ("t1k_out", 24000, 512, 512, EPI_BIAS_RESID, torch.float32, 1),
    ("t1k_fc", 24000, 2048, 512, EPI_BIAS_QGELU, torch.bfloat16, 1),
    ("t1k_proj", 24000, 512, 2048, EPI_BIAS_RESID, torch.float32, 1),
    ("t1k_du", 24000, 2048, 512, EPI_QGELU_BWD, torch.bfloat16, 1),
    ("t1k_dh2", 24000, 512, 2048, EPI_NONE, torch.float32, 1),
    ("t1k_dh2", 24000, 512, 2048, EPI_NONE, torch.float32, 3),
    ("t1k_dh1", 24000, 512, 512, EPI_NONE, torch.float32, 1),
    ("t1k_dh1", 24000, 512, 512, EPI_NONE, torch.float32, 2),
    ("t1k_da", 24000, 512, 512, EPI_NONE, torch.bfloat16, 1),
]
for name, M, N, K, epi, odt, split in SHAPES:
    if ONLY is not None and name != ONLY and not (ONLY.endswith("*") and name.startswith(ONLY[:-1])):
        continue
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.empty((split, M, N) if split > 1 else (M, N), dtype=odt, device=dev)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev)
    kw = dict(bias=bias if epi in (EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID) else None,
              resid=resid if epi == EPI_BIAS_RESID else None,
              aux=aux if epi in (EPI_QGELU_BWD,) else None, split_k=split)
    if M % 221 == 0 and M >= 1768 and os.environ.get("BENCH_NO_UNITS") != "1":
        kw["row_units"] = (197, 24, 197 * (M // 221))              # the image tower's layout: one image = 197 frozen + 24 prompt rows
    res = []
    cfgs = [int(c) for c in os.environ.get('BENCH_CFGS', '').split(',') if c] if (ONLY is not None or 'BENCH_CFGS' in os.environ) else [2, 5, 6]
    for cfg in [0] + cfgs:
        for _ in range(3):
            ops.gemm_nt(a, w, out, epi, tile_config=cfg, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.gemm_nt(a, w, out, epi, tile_config=cfg, **kw)
        g.replay()
        ts = []
        for _ in range(7):
            s.record(); g.replay(); e.record(); e.synchronize()
            ts.append(1e3 * s.elapsed_time(e) / 20)
        us = sorted(ts)[len(ts) // 2]                   # median of 7 graph replays of 20 launches
        cur = out.float().clone()
        if cfg == 0:
            ref0 = cur
        err = (cur - ref0).abs().max().item()
        res.append(f"cfg{cfg}: {us:7.2f} us {2.0*M*N*K/us/1e6:7.1f} TF" + (f" ERR {err:.2e}" if err > 0 else ""))
    print(f"{name:9s} M={M:5d} N={N:5d} K={K:5d} split={split}: " + " | ".join(res))
