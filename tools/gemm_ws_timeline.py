#!/usr/bin/env python3
"""In-kernel phase timeline of rpo_gemm_ws (debug build with -DRPO_TIMELINE: tools/build_debug.sh): s_memtime stamps of
thread 0 of 8 workgroups -- start, prologue requests issued (1), first chunk parked (2), end of every chunk (3 + i), loop
end / reduction barrier (58 / 59), end (61) -- and the shader clock over the span (s_memrealtime, 100 MHz)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rpo_amd import _lib, ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_RESID, EPI_NONE, EPI_QGELU_BWD

dbg = os.path.join(ROOT, "rpo_amd", "build", "librpo_hip_dbg.so")
if not os.path.exists(dbg) or os.environ.get("RPO_REBUILD_DBG"):
    # one recipe for the -DRPO_TIMELINE library (tools/build_debug.sh: EVERY translation unit the ABI needs); compiler output
    # goes to a log next to the library, never into the timeline this script prints
    os.makedirs(os.path.dirname(dbg), exist_ok=True)
    with open(dbg + ".log", "w") as log:
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_debug.sh")], stdout=log, stderr=log)
lib = _lib.load(dbg)
_lib._lib = lib
lib.rpo_debug_set_timeline.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
assert lib.rpo_debug_set_timeline(buf.data_ptr()) == 0
FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
bf = torch.bfloat16


def show(name):
    t = buf.view(8, 64).cpu()
    print(f"== {name}")
    for b in range(8):
        r = t[b]
        if r[0] == 0:
            continue
        ch = [int(r[3 + i]) for i in range(41) if r[3 + i] != 0]
        prev, per = int(r[2]), []
        for c in ch:
            per.append(c - prev); prev = c
        rt = int(r[63] - r[62])
        clk = f" | {int(r[61] - r[0]) / rt * 0.1:.2f} GHz over {rt / 100:.2f} us" if rt > 0 else ""
        print(f" wg {b * 97:4d}: setup {int(r[1] - r[0]):5d} | prologue (requests -> chunk 0 parked) {int(r[2] - r[1]):5d} | chunks {per} | "
              f"to barrier {int(r[59] - r[58]):5d} | epilogue {int(r[61] - r[59]):5d} | total {int(r[61] - r[0])}{clk}")


for name, M, N, K, epi, odt, split, cfg in [("img d c_proj 330", 768, 3072, 768, EPI_QGELU_BWD, bf, 1, 330),
                                            ("img d c_fc 330 split 4", 768, 768, 3072, EPI_NONE, torch.float32, 4, 330),
                                            ("img d q 220", 768, 768, 768, EPI_NONE, torch.float32, 1, 220),
                                            ("txt q 110", 456, 512, 512, EPI_BIAS, bf, 1, 110),
                                            ("txt proj 110", 456, 512, 2048, EPI_BIAS_RESID, torch.float32, 1, 110)]:
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    wp = ops.gemm_ws_pack(w)
    out = torch.empty((split, M, N) if split > 1 else (M, N), dtype=odt, device=dev)
    kw = dict(bias=torch.randn(N, device=dev) if epi in (EPI_BIAS, EPI_BIAS_RESID) else None,
              resid=torch.randn(M, N, device=dev) if epi == EPI_BIAS_RESID else None,
              aux=torch.randn(M, N, device=dev).to(bf) if epi == EPI_QGELU_BWD else None, split_k=split, tile_config=cfg)
    for cold in (False, True):
        for _ in range(3):
            if cold:
                FLUSH.fill_(1)
            buf.zero_()
            ops.gemm_ws(a, wp, out, epi, **kw)
        torch.cuda.synchronize()
        show(name + (" -- operands COLD" if cold else " -- operands warm"))
