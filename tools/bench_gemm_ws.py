#!/usr/bin/env python3
"""rpo_gemm_ws against rpo_gemm_nt at the prompt-row shapes of the step (HIP events around graph replays of 20 launches).
Two regimes per shape: the same weight every launch (L2-hot) and a rotation over 12 layers' weights (what a chain sees:
each launch's weight was last touched 12 launches ago)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rpo_amd import ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_LN_BIAS, EPI_NONE, EPI_QGELU_BWD

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [  # name, M, N, K, epi, out dtype, split candidates
    ("img d c_proj", 768, 3072, 768, EPI_QGELU_BWD, bf, (1,)),
    ("img d c_fc", 768, 768, 3072, EPI_NONE, torch.float32, (3, 4, 2, 6)),
    ("img d q", 768, 768, 768, EPI_NONE, torch.float32, (2, 1, 4)),
    ("img d proj", 768, 768, 512, EPI_NONE, torch.float32, (1, 2)),
    ("img last q", 768, 768, 768, EPI_BIAS, bf, (1,)),
    ("img last out", 768, 768, 768, EPI_BIAS_RESID, torch.float32, (1,)),
    ("img last fc", 768, 3072, 768, EPI_BIAS_QGELU, bf, (1,)),
    ("img last proj", 768, 768, 3072, EPI_BIAS_RESID, torch.float32, (1,)),
    ("txt q", 456, 512, 512, EPI_BIAS, bf, (1,)),
    ("txt out", 456, 512, 512, EPI_BIAS_RESID, torch.float32, (1,)),
    ("txt fc", 456, 2048, 512, EPI_BIAS_QGELU, bf, (1,)),
    ("txt proj", 456, 512, 2048, EPI_BIAS_RESID, torch.float32, (1,)),
    ("txt d c_proj", 456, 2048, 512, EPI_QGELU_BWD, bf, (1,)),
    ("txt d c_fc", 456, 512, 2048, EPI_NONE, torch.float32, (3, 4, 2)),
    ("txt d q", 456, 512, 512, EPI_NONE, torch.float32, (2, 1)),
    ("txt d out", 456, 512, 512, EPI_NONE, bf, (1,)),
    ("b4 d c_proj", 96, 3072, 768, EPI_QGELU_BWD, bf, (1,)),
    ("b4 d c_fc", 96, 768, 3072, EPI_NONE, torch.float32, (3, 4, 8)),
    ("L14 d c_proj", 384, 4096, 1024, EPI_QGELU_BWD, bf, (1,)),
    ("L14 d c_fc", 384, 1024, 4096, EPI_NONE, torch.float32, (3, 4, 8)),
]
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
LAYERS = 12


def timed(fn):
    for _ in range(2):
        fn(0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for it in range(24):
            fn(it)
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        s.record(); g.replay(); e.record(); e.synchronize()
        ts.append(1e3 * s.elapsed_time(e) / 24)
    return sorted(ts)[len(ts) // 2]


for name, M, N, K, epi, odt, splits in SHAPES:
    if ONLY is not None and ONLY not in name:
        continue
    a = torch.randn(M, K, device=dev).to(bf)
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(LAYERS)]
    wps = [ops.gemm_ws_pack(w) for w in ws]
    bias, resid = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev).to(bf)
    base_kw = dict(bias=bias if epi in (EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID) else None,
                   resid=resid if epi == EPI_BIAS_RESID else None, aux=aux if epi == EPI_QGELU_BWD else None)
    line = [f"{name:14s} M={M:4d} N={N:4d} K={K:4d}"]
    for split in splits:
        out = torch.empty((split, M, N) if split > 1 else (M, N), dtype=odt, device=dev)
        kw = dict(base_kw, split_k=split)
        for rot, tag in ((False, "hot"), (True, "rot")):
            nt = timed(lambda it: ops.gemm_nt(a, ws[it % LAYERS if rot else 0], out, epi, **kw))
            res = [f"nt {nt:5.2f}"]
            for cfg in (0, 330, 220, 120, 110):
                try:
                    t = timed(lambda it: ops.gemm_ws(a, wps[it % LAYERS if rot else 0], out, epi, tile_config=cfg, **kw))
                    res.append(f"ws{cfg or 'auto'} {t:5.2f}")
                except Exception as ex:                                       # geometry not admitted for this epilogue
                    res.append(f"ws{cfg} n/a")
            line.append(f"| split {split} {tag}: " + " ".join(res))
    print(" ".join(line), flush=True)
