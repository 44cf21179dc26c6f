#!/usr/bin/env python3
"""In-kernel phase timeline of rpo_gemm_nt (debug build with -DRPO_GEMM_TIMELINE)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rpo_amd import _lib, ops
from rpo_amd._lib import EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_NONE
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


def show(name, M, N, K, cfg):
    t = buf.view(8, 64).cpu()
    print(f"== {name} M={M} N={N} K={K}  (s_memtime ticks = 100 MHz? shown raw deltas)")
    for b in range(8):
        r = t[b]
        if r[0] == 0: continue
        nk = int((r[2:52] != 0).sum())
        deltas = [int(r[2 + i] - r[2 + i - 1]) for i in range(1, min(nk, 14))]
        if cfg in (8, 10, 11):
            nk32 = K // 32
            rt = int(r[63] - r[62])                    # 100 MHz ticks over the same span
            clk = f" | {int(r[61]-r[0]) / rt * 0.1:.2f} GHz over {rt / 100:.1f} us" if rt > 0 else ""
            print(f" wg {b*97:5d}: start->loop {int(r[2]-r[0]):6d} | loop {int(r[60]-r[2]):6d} = {nk32} tiles x {int(r[60]-r[2])//nk32} | epilogue {int(r[61]-r[60]):6d} | total {int(r[61]-r[0])}{clk}")
            continue
        if cfg == 7:
            nk32 = K // 32
            g = lambda i: int(r[40 + i]) // nk32
            print(f" wg {b*97:5d}: start->loop {int(r[2]-r[0]):6d} | loop {int(r[60]-r[2]):6d} = {nk32} tiles x {int(r[60]-r[2])//nk32} | epilogue {int(r[61]-r[60]):6d} | total {int(r[61]-r[0])}"
                  f" || per tile, group 0: load {g(0)} bar {g(1)} compute {g(2)} bar {g(3)}; group 1: load {g(4)} bar {g(5)} compute {g(6)} bar {g(7)}")
            continue
        print(f" wg {b*97:5d}: start->tile0 {int(r[2]-r[0]):6d} | per-k-tile {deltas} | last-tile->epi {int(r[60]-r[2+nk-1]):6d} | epilogue {int(r[61]-r[60]):6d} | total {int(r[61]-r[0])} || wave0 per-iter: vmcnt-wait {int(r[53])//max(nk,1)} barrier {int(r[54])//max(nk,1)} body(issue) {int(r[55])//max(nk,1)}")


for name, M, N, K, epi, odt, cfg in [("c_proj_mid2", 7072, 768, 3072, EPI_NONE, torch.float32, 2), 
                                     ("c_proj_tall", 7072, 768, 3072, EPI_NONE, torch.float32, 6), ("one_wg_mid", 128, 128, 3072, EPI_NONE, torch.float32, 2),
                                     ("c_fc", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 0), ("qkv_big", 7072, 2304, 768, EPI_BIAS, torch.bfloat16, 3),
                                     ("qkv_pingpong (stamps per 32-deep k-tile)", 7072, 2304, 768, EPI_BIAS, torch.bfloat16, 7),
                                     ("qkv_w4 (one wave per SIMD)", 7072, 2304, 768, EPI_BIAS, torch.bfloat16, 8),
                                     ("c_fc_w4 (one wave per SIMD, 2 rounds)", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 8),
                                     ("c_fc_w4g (224x384, one round)", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 10),
                                     ("c_fc_w4g bias-only epilogue (store cost alone)", 7072, 3072, 768, EPI_BIAS, torch.bfloat16, 10),
                                     ("c_fc_w4g + saved QuickGELU operand of the prompt rows (act dtype)", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 10),
                                     ("c_fc_w4g + saved QuickGELU operand of the prompt rows (fp32)", 7072, 3072, 768, EPI_BIAS_QGELU, torch.bfloat16, 10),
                                     ("c_proj_w4k (224x96, waves split k)", 7072, 768, 3072, EPI_BIAS_RESID, torch.float32, 11),
                                     ("out_proj_w4k (224x96, waves split k)", 7072, 768, 768, EPI_BIAS_RESID, torch.float32, 11)]:
    if os.environ.get("TIMELINE_ONLY") and os.environ["TIMELINE_ONLY"] not in name:
        continue
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=odt, device=dev); bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == EPI_BIAS_RESID else None
    units = (197, 24, 6304) if M == 7072 else None
    for cold in ((False, True) if cfg in (8, 10, 11) else (False,)):
      for _ in range(3):
        if cold: FLUSH.fill_(1)                      # 512 MB written in between: operands out of L2 / MALL
        buf.zero_()
        aux = None
        if "saved" in name:
            aux = torch.empty(768, N, dtype=torch.bfloat16 if "act dtype" in name else torch.float32, device=dev)
        ops.gemm_nt(a, w, out, epi, bias=bias if epi != EPI_NONE else None, resid=resid, tile_config=cfg, row_units=units,
                    aux=aux, aux_row0=6304 if aux is not None else 0)
      torch.cuda.synchronize()
      show(name + (" -- operands COLD" if cold else ""), M, N, K, cfg)
    continue
